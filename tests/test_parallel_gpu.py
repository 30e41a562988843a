"""GPU, world_size 2: the product's data-parallel train step (HIP path + GradSync bucket hooks + FusedClampAdam with
grad_scale) on two ranks equals the single-process step on the concatenated batch.

A one-GPU box cannot host two RCCL ranks, so the two processes share cuda:0 and exchange the flat gradient buckets
over gloo (which stages CUDA tensors through the host); everything except the transport -- bucket order, in-place
reduction of the flat buffers that ARE the .grad views, 1/world applied before the clamp -- is what bench.py runs
under RCCL."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

INFO = [dict(att_num=9, att_feat_size=24, fc_feat_size=16), dict(att_num=6, att_feat_size=40, fc_feat_size=40)]
GAIN = 40.0     # scales the loss so that the element-wise clamp bites


def _setup():
    sys.path.insert(0, ROOT)
    from oracle import rfn_oracle as O
    cfg = O.make_cfg(INFO, vocab_size=50, rnn_size=32, input_encoding_size=32, att_hid_size=32, num_review_steps_0=3,
                     num_review_steps=3, top_words_count=20, seq_length=5)
    return O, cfg, O.seeded_params(cfg, 7), O.synthetic_batch(cfg, 8, seed=3)


def _step(cfg, P, batch, sync_world, dev, shard=None):
    import recurrent_fusion_network_amd as R
    from recurrent_fusion_network_amd import parallel as DP
    model = R.RecurrentFusionModel(cfg)
    model.load_state_dict(P)
    model = model.to(dev).train()
    crit = R.ReviewNetEnsembleCriterion(cfg)
    opt = R.FusedClampAdam(model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=1.0, shard=shard)
    sync = DP.GradSync(model, sync_world, shard_optimizer=opt if shard else None)
    fc, att, labels, masks, top = [[x.to(dev) for x in t] if isinstance(t, list) else t.to(dev) for t in batch]
    opt.zero_grad()
    lp, reason = model(fc, att, labels)
    (crit(lp, labels[:, 1:], masks[:, 1:], reason, top, 1.0) * GAIN).backward()
    order = list(sync.buckets)
    scale = sync.finish()
    grads = {k: (p.grad * scale).cpu() for k, p in model.named_parameters()}      # p.grad holds the SUM over ranks
    opt.step(grad_scale=scale)
    torch.cuda.synchronize()
    return order, grads, {k: p.detach().cpu() for k, p in model.named_parameters()}


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    O, cfg, P, batch = _setup()
    from recurrent_fusion_network_amd import parallel as DP
    r, w, _ = DP.init_from_env('gloo')
    assert (r, w) == (rank, world)
    lo, hi = DP.shard_rows(8, rank, world)
    shard = [[x[lo:hi] for x in t] if isinstance(t, list) else t[lo:hi] for t in batch]
    order, grads, params = _step(cfg, P, shard, world, torch.device('cuda:0'))
    if rank == 0:
        q.put((order, {k: v.numpy() for k, v in grads.items()}, {k: v.numpy() for k, v in params.items()}))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_gpu_ranks_equal_the_single_process_step(dev):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    order, avg, stepped = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    O, cfg, P, batch = _setup()
    order1, full, ref = _step(cfg, P, batch, 1, dev)
    # buckets are announced in the order backward finishes them: decoder, fusion core, every encoder's big a-bucket, then the
    # b-buckets of the long att_2_att_h products
    assert order == order1 == ['decoder', 'core', 'enc0a', 'enc1a', 'enc0b', 'enc1b']
    clipped = False
    for k, g in full.items():
        a = torch.from_numpy(avg[k])
        assert float((a - g).abs().max()) <= 1e-5 + 1e-4 * float(g.abs().max()), k   # mean of shard grads == full grad
        clipped |= bool((g.abs() > 1.0).any())
        sel = g.abs() > 1e-4
        if bool(sel.any()):
            assert float((torch.from_numpy(stepped[k])[sel] - ref[k][sel]).abs().max()) < 5e-6, k
    assert clipped


# ---- optimizer-state sharding: the real FusedClampAdam(shard=(rank, world)) against the all-reduce path -----------------------
def _two_steps(cfg, P, batch, world, rank, dev, sharded):
    """Two train steps on this rank's rows -> (parameters, full Adam moments gathered from the shards)."""
    import recurrent_fusion_network_amd as R
    from recurrent_fusion_network_amd import parallel as DP
    model = R.RecurrentFusionModel(cfg)
    model.load_state_dict(P)
    model = model.to(dev).train()
    crit = R.ReviewNetEnsembleCriterion(cfg)
    opt = R.FusedClampAdam(model, lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=1.0,
                           shard=(rank, world) if sharded else None)
    sync = DP.GradSync(model, world, shard_optimizer=opt if sharded else None)
    fc, att, labels, masks, top = [[x.to(dev) for x in t] if isinstance(t, list) else t.to(dev) for t in batch]
    for _ in range(2):
        opt.zero_grad()
        lp, reason = model(fc, att, labels)         # the sharded optimizer's all-gathers are waited for in here
        (crit(lp, labels[:, 1:], masks[:, 1:], reason, top, 1.0) * GAIN).backward()
        opt.step(grad_scale=sync.finish())
    opt.wait_params()
    torch.cuda.synchronize()
    sd = opt.state_dict()               # collective under sharding: every rank calls it
    moments = {n: (b['m'].cpu(), b['v'].cpu()) for n, b in sd['buckets'].items()}
    # parameters read through the model (views of the gathered flat buffers); the shard padding is no parameter
    return {k: p.detach().cpu() for k, p in model.named_parameters()}, moments, {n: model.bucket_layout(n) for n in model.bucket_names()}


def _shard_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    O, cfg, P, batch = _setup()
    from recurrent_fusion_network_amd import parallel as DP
    DP.init_from_env('gloo')
    lo, hi = DP.shard_rows(8, rank, world)
    rows = [[x[lo:hi] for x in t] if isinstance(t, list) else t[lo:hi] for t in batch]
    dev = torch.device('cuda:0')
    p_ar, m_ar, lay_ar = _two_steps(cfg, P, rows, world, rank, dev, sharded=False)
    p_sh, m_sh, lay_sh = _two_steps(cfg, P, rows, world, rank, dev, sharded=True)
    ok = all(torch.equal(p_ar[k], p_sh[k]) for k in p_ar)
    # moments: compare parameter by parameter (the two layouts differ only in the padding behind the last parameter)
    for name in m_ar:
        params, offs, _ = lay_ar[name]
        params2, offs2, _ = lay_sh[name]
        assert offs == offs2
        for p_, o in zip(params, offs):
            n = p_.numel()
            ok = ok and torch.equal(m_ar[name][0][o:o + n], m_sh[name][0][o:o + n]) and torch.equal(m_ar[name][1][o:o + n],
                                                                                                    m_sh[name][1][o:o + n])
    q.put((rank, bool(ok)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_sharded_optimizer_equals_the_all_reduce_path_bit_for_bit(dev):
    """Two ranks sharing the GPU, two steps each way: all-reduce + full update on every rank against reduce + 1/2 of every
    bucket updated per rank + all-gather of the parameters (waited for inside the next forward): parameters and both Adam
    moments of every parameter identical on both ranks."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31000 + os.getpid() % 2000
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=900) for _ in range(2))
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert got == {0: True, 1: True}

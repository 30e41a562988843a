"""CPU, world_size 2, 3 (uneven shards) and 4 over gloo: the data-parallel recipe of recurrent_fusion_network_amd/parallel.py
(shard rows -> local gradients -> SUM all-reduce of flat buffers -> scale 1/world BEFORE the element-wise
clamp -> Adam) reproduces the single-process step on the concatenated batch.
The per-rank compute is the CPU oracle here (the HIP path needs a GPU); the distributed plumbing under test
is the product's."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from oracle import rfn_oracle as O
    from recurrent_fusion_network_amd import parallel as DP
    torch.set_num_threads(2)
    r, w, _ = DP.init_from_env('gloo')
    assert (r, w) == (rank, world)
    info = [dict(att_num=5, att_feat_size=24, fc_feat_size=24), dict(att_num=7, att_feat_size=40, fc_feat_size=32)]
    cfg = O.make_cfg(info, vocab_size=50, rnn_size=16, input_encoding_size=16, att_hid_size=16, num_review_steps_0=3,
                     num_review_steps=3, top_words_count=20, seq_length=5)
    P = O.seeded_params(cfg, 7)
    fc, att, labels, masks, top = O.synthetic_batch(cfg, 8, seed=3)
    lo, hi = DP.shard_rows(8, rank, world)
    sl = lambda t: t[lo:hi]  # noqa: E731
    loss, grads = O.train_step_loss_and_grads(cfg, P, [sl(f) for f in fc], [sl(a) for a in att], sl(labels), sl(masks),
                                              sl(top), 1.0)
    keys = sorted(grads)
    # shards of different sizes (world 3: 3 + 3 + 2 rows) weigh their local mean by rows / all rows (1.0 for equal shards)
    w_loss = DP.shard_loss_scale(hi - lo, 8, world)
    assert world == 3 or w_loss == 1.0
    flat = torch.cat([grads[k].reshape(-1) * (40.0 * w_loss) for k in keys])     # x40: make the clamp bite
    # the overlapped path of bench.py: buckets are handed to GradSync as backward finishes them
    class _M:                       # the two attributes GradSync touches
        grad_ready_hook = None
    holder = _M()
    sync = DP.GradSync(holder, world)
    n3 = flat.numel() // 3
    chunks = [flat[:n3], flat[n3:2 * n3], flat[2 * n3:]]          # views: reduced in place
    sync.record = True              # bench.py's exposed_ms: how long the step waited for the exchange, bucket by bucket
    for i, c in enumerate(chunks):
        holder.grad_ready_hook('bucket%d' % i, c)
    assert sync.buckets == ['bucket0', 'bucket1', 'bucket2']
    scale = sync.finish()
    assert scale == 1.0 / world and not sync.works
    total, per, how = sync.exposed_ms()
    assert total >= 0.0 and sorted(per) == ['bucket0', 'bucket1', 'bucket2'] and abs(sum(per.values()) - total) < 1e-9
    assert how.startswith('host clock') and sync.exposed_ms() == (None, {}, None)       # drained by the read
    t = DP.max_over_ranks(float(rank + 1), world, torch.device('cpu'))
    assert t == float(world)
    # clamp AFTER averaging, then Adam -- as rfn_adam_step does with grad_scale
    off, avg = 0, {}
    for k in keys:
        n = grads[k].numel()
        avg[k] = (flat[off:off + n] * scale).view_as(grads[k])
        off += n
    P_new = {k: v.clone() for k, v in P.items()}
    O.clip_and_adam(P_new, avg, {}, lr=5e-4, weight_decay=1e-5, grad_clip=1.0)
    if rank == 0:
        q.put({k: v.numpy() for k, v in P_new.items()})
        q.put({k: v.numpy() for k, v in avg.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 4])
def test_n_rank_step_equals_single_process_step(world):
    sys.path.insert(0, ROOT)
    from oracle import rfn_oracle as O
    from recurrent_fusion_network_amd import parallel as DP
    assert [DP.shard_rows(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    stepped = q.get(timeout=300)
    avg = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    info = [dict(att_num=5, att_feat_size=24, fc_feat_size=24), dict(att_num=7, att_feat_size=40, fc_feat_size=32)]
    cfg = O.make_cfg(info, vocab_size=50, rnn_size=16, input_encoding_size=16, att_hid_size=16, num_review_steps_0=3,
                     num_review_steps=3, top_words_count=20, seq_length=5)
    P = O.seeded_params(cfg, 7)
    batch = O.synthetic_batch(cfg, 8, seed=3)
    loss, grads = O.train_step_loss_and_grads(cfg, P, *batch, 1.0)
    full = {k: g * 40.0 for k, g in grads.items()}
    clipped_some = False
    for k in grads:
        # (row-weighted) mean of the shard gradients == gradient of the concatenated batch
        assert float((torch.from_numpy(avg[k]) - full[k]).abs().max()) < 1e-5 + 1e-4 * float(full[k].abs().max()), k
        clipped_some |= bool((full[k].abs() > 1.0).any())
    assert clipped_some
    P_ref = {k: v.clone() for k, v in P.items()}
    O.clip_and_adam(P_ref, full, {}, lr=5e-4, weight_decay=1e-5, grad_clip=1.0)
    for k in P_ref:
        sel = full[k].abs() > 1e-4
        if bool(sel.any()):
            assert float((torch.from_numpy(stepped[k])[sel] - P_ref[k][sel]).abs().max()) < 5e-6, k


# ---- optimizer-state sharding: reduce the bucket, update 1/world of it, all-gather the parameters -------
def _shard_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from oracle import rfn_oracle as O
    from recurrent_fusion_network_amd import parallel as DP
    torch.set_num_threads(2)
    DP.init_from_env('gloo')
    info = [dict(att_num=5, att_feat_size=24, fc_feat_size=24), dict(att_num=7, att_feat_size=40, fc_feat_size=32)]
    cfg = O.make_cfg(info, vocab_size=50, rnn_size=16, input_encoding_size=16, att_hid_size=16, num_review_steps_0=3,
                     num_review_steps=3, top_words_count=20, seq_length=5)
    P = O.seeded_params(cfg, 7)
    fc, att, labels, masks, top = O.synthetic_batch(cfg, 8, seed=3)
    lo, hi = DP.shard_rows(8, rank, world)
    sl = lambda t: t[lo:hi]  # noqa: E731
    keys = sorted(P)
    offs, n = {}, 0
    for k in keys:                      # the product's bucket layout: parameters back to back, 16-B aligned starts
        offs[k] = n
        n += (P[k].numel() + 3) & ~3
    pad = DP.shard_pad(world)
    total = -(-n // pad) * pad
    assert total % (4 * world) == 0 and DP.shard_bounds(total, world - 1, world)[1] == total

    def flat_of(tensors):
        f = torch.zeros(total)
        for k in keys:
            f[offs[k]:offs[k] + tensors[k].numel()] = tensors[k].reshape(-1)
        return f

    p_flat = flat_of(P)
    a, b = DP.shard_bounds(total, rank, world)
    p_shd = p_flat.clone()
    st_ref, st_shd = {}, {}             # oracle Adam state: whole bucket (reference) / this rank's shard only
    for step in (1, 2):                 # two steps: the second one starts from gathered parameters and carried moments
        cur = {k: p_shd[offs[k]:offs[k] + P[k].numel()].view_as(P[k]).clone() for k in keys}
        _, grads = O.train_step_loss_and_grads(cfg, cur, [sl(f) for f in fc], [sl(x) for x in att], sl(labels), sl(masks),
                                               sl(top), 1.0)
        g = flat_of({k: grads[k] * (40.0 * DP.shard_loss_scale(hi - lo, 8, world)) for k in keys})
        dist.all_reduce(g)              # gloo has no reduce-scatter: GradSync all-reduces there, the update reads its slice
        g = g / world
        assert bool((g.abs() > 1.0).any())                       # the clamp bites
        # reference: the unsharded update of the whole bucket
        ref_p = {'w': p_flat}
        O.clip_and_adam(ref_p, {'w': g}, st_ref, lr=5e-4, weight_decay=1e-5, grad_clip=1.0)
        # sharded: update [a, b) only, with this rank's moment shards, then gather in place
        mine = {'w': p_shd[a:b].clone()}
        O.clip_and_adam(mine, {'w': g[a:b]}, st_shd, lr=5e-4, weight_decay=1e-5, grad_clip=1.0)
        p_shd[a:b] = mine['w']
        w = DP.gather_shards(p_shd, rank, world, async_op=True)
        w.wait()
        assert torch.equal(p_shd, p_flat), 'step %d: gathered parameters differ from the unsharded update' % step
        assert torch.equal(st_shd['m']['w'], st_ref['m']['w'][a:b]) and torch.equal(st_shd['v']['w'], st_ref['v']['w'][a:b])
    if rank == 0:
        q.put(float(p_shd.double().sum()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 4])
def test_sharded_update_equals_the_unsharded_update_bit_for_bit(world):
    """Every rank updates 1/world of each flat bucket (its own Adam moments only) and the parameters are all-gathered in
    place: element for element the arithmetic of the all-reduce path, so parameters and moments must be IDENTICAL after one
    and after two steps, for world sizes that do and do not divide the bucket (padding to 4 * world elements)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0

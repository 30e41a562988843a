"""GPU, world_size 2: ensemble members living on DIFFERENT ranks.

Reference: eval_utils.model_ensemble_feat_array_one_step_multi_gpu (eval_utils.py:293-317) keeps one member per GPU and
moves every member's (B, V+1) logits to one device with `.cuda()` copies before it averages them.  Here every rank runs its
own member's decoder step and the logit sum crosses the ranks through ONE sum all-reduce per step
(`EnsembleDecoder(process_group=...)`).  Two processes share the test GPU (a one-GPU box cannot host two RCCL ranks, so
the all-reduce goes over gloo, as in tests/test_parallel_gpu.py), one member each; greedy and beam decodes on both ranks must
equal the single-process two-member ensemble BIT FOR BIT (a two-term sum does not depend on who adds)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
SEEDS = (None, 41)          # member 0: the golden tier's weights; member 1: a second seeded set


def _members_and_batch():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from conftest import load_case
    from oracle import rfn_oracle as O
    cfg, spec, P, batch, gold = load_case('mid')
    return cfg, [P if s is None else O.seeded_params(cfg, s) for s in SEEDS], batch


def _decode(ens, fc, att):
    seq, seq_lp, lp_all = ens.sample(fc, att)
    bseq, bseq_lp, top_seq, top_prob = ens.sample_beam(fc, att, {'beam_size': 3})
    return dict(seq=seq.cpu(), seq_lp=seq_lp.cpu(), lp_all=lp_all.cpu(), bseq=bseq.cpu(), bseq_lp=bseq_lp.cpu(),
                top_seq=[t.cpu() for t in top_seq], top_prob=[list(p) for p in top_prob])


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    cfg, Ps, batch = _members_and_batch()
    import torch.distributed as dist
    from test_model_gpu import build, to_dev
    from recurrent_fusion_network_amd import parallel as DP
    from recurrent_fusion_network_amd.ensemble import EnsembleDecoder
    DP.init_from_env('gloo')
    dev = torch.device('cuda:0')
    fc, att, labels, masks, top = to_dev(batch, dev)
    ens = EnsembleDecoder([build(cfg, Ps[rank], dev)], process_group=dist.group.WORLD)      # ONE member on this rank
    assert ens.n_total == world
    out = _decode(ens, fc, att)
    q.put((rank, {k: (v.numpy() if torch.is_tensor(v) else [t.numpy() if torch.is_tensor(t) else t for t in v]) for k, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_members_on_two_ranks_equal_the_single_process_ensemble(dev):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 30500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=900) for _ in range(2))
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    cfg, Ps, batch = _members_and_batch()
    from test_model_gpu import build, to_dev
    from recurrent_fusion_network_amd.ensemble import EnsembleDecoder
    fc, att, labels, masks, top = to_dev(batch, dev)
    want = _decode(EnsembleDecoder([build(cfg, p, dev) for p in Ps]), fc, att)
    # the two members disagree, so the ensemble is not either member's own decode
    with torch.no_grad():
        own = [build(cfg, p, dev).sample(fc, att, {'sample_max': 1})[0].cpu() for p in Ps]
    assert not torch.equal(own[0], own[1])
    for rank in (0, 1):
        g = got[rank]
        for k in ('seq', 'seq_lp', 'lp_all', 'bseq', 'bseq_lp'):
            assert torch.equal(torch.from_numpy(g[k]), want[k]), (rank, k)
        assert len(g['top_seq']) == len(want['top_seq'])
        for a, b in zip(g['top_seq'], want['top_seq']):
            assert torch.equal(torch.from_numpy(a), b), rank
        assert g['top_prob'] == want['top_prob'], rank

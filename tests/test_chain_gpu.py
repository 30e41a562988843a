"""The persistent recurrence kernels (csrc/rfn_chain.hip: all steps of a stage-II / decoder recurrence in ONE launch, grid
barrier between dependent phases; opt-in, RFN_PATH_OPT_PERSIST_*) against the three-launches-per-step chain they replace: the same
device bodies on the same tiles in the same k order, so everything must agree BIT FOR BIT -- log-probs, reason heads, loss,
every gradient, greedy ids -- on the reference-pinned golden tiers and at the benchmark shapes, run after run (a stale
cache line in a hand-off between blocks would show up as a sporadic difference)."""
import pytest
import torch

from conftest import load_case
from test_model_gpu import build, to_dev

pytestmark = pytest.mark.gpu


def _step(model, crit, batch):
    fc, att, labels, masks, top = batch
    model.zero_grad()
    log_prob, reason = model(fc, att, labels)
    loss = crit(log_prob, labels[:, 1:], masks[:, 1:], reason, top, 1.0)
    loss.backward()
    out = {'log_prob': log_prob.detach().clone(), 'loss': loss.detach().clone()}
    for j, r in enumerate(reason):
        out['reason%d' % j] = r.detach().clone()
    for k, p in model.named_parameters():
        out['grad:' + k] = p.grad.detach().clone()
    return out


def _assert_same(a, b, what):
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), '%s: %s differs (max abs %g)' % (what, k, float((a[k] - b[k]).abs().max()))


@pytest.mark.parametrize('name', ['mid', 'c2', 'c3', 'tiny0', 'odd'])
@pytest.mark.parametrize('train', [False, True])
@pytest.mark.parametrize('which', [1, 2, 4, 8, 15])
def test_persistent_chains_match_the_per_step_launches_bit_for_bit(dev, name, train, which):
    """Golden tiers: `mid` / `c2` / `c3` take the persistent kernels (hidden sizes are whole K steps of 64), `tiny0` / `odd`
    do not qualify and must quietly run the per-step launches under both settings."""
    import recurrent_fusion_network_amd as R
    N = R._native
    cfg, spec, P, batch, gold = load_case(name)
    if train:
        cfg.drop_prob_lm, cfg.drop_prob_reason = 0.3, 0.2       # the Philox masks ride in the chain's epilogues too
    batch = to_dev(batch, dev)
    crit = R.ReviewNetEnsembleCriterion(cfg)
    ref = build(cfg, P, dev, train=train)
    assert ref.path_flags == 0
    # the chains are built from the three-launch decoder cell of rounds 3-5 (z2h(z) as a per-step product): compare with that
    # form, not with the default (hoisted, two launches, csrc/rfn_deccell.hip), which rounds differently
    ref.path_flags = N.PATH_OPT_DEC_UNHOISTED
    new = build(cfg, P, dev, train=train)
    new.path_flags = which | N.PATH_OPT_DEC_UNHOISTED
    for rnd in range(2):
        torch.manual_seed(5 + rnd)
        want = _step(ref, crit, batch)
        torch.manual_seed(5 + rnd)
        got = _step(new, crit, batch)
        _assert_same(got, want, '%s round %d' % (name, rnd))
    if not train:
        with torch.no_grad():
            a = ref.sample(batch[0], batch[1], {'sample_max': 1})
            b = new.sample(batch[0], batch[1], {'sample_max': 1})
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


@pytest.mark.parametrize('B', [5, 32, 64, 200])
def test_persistent_chains_at_the_c2_shape_run_after_run(dev, B):
    """BASELINE config 2's shape (M = 2, L = 49, D = 512, R = A = E = 512, T1 = T2 = 8, 17 decoder steps) at several batch
    sizes -- ragged 32-row tiles, fewer tiles than blocks, more tiles than blocks, one- and multi-XCD barriers -- 12 runs
    each against ONE run of the per-step launches: every run must reproduce it exactly."""
    import recurrent_fusion_network_amd as R
    from oracle import rfn_oracle as O
    N = R._native
    info = [dict(att_num=49, att_feat_size=512, fc_feat_size=512)] * 2
    cfg = O.make_cfg(info, vocab_size=9487)
    g = torch.Generator(device='cpu').manual_seed(77)
    model = R.RecurrentFusionModel(cfg).to(dev)
    with torch.no_grad():
        for _, p in sorted(model.named_parameters()):
            p.copy_((torch.rand(p.shape, generator=g) * 0.2 - 0.1).to(dev))
    fc, att, labels, masks, top = O.synthetic_batch(cfg, B, seed=3)
    batch = to_dev((fc, att, labels, masks, top), dev)
    crit = R.ReviewNetEnsembleCriterion(cfg)
    model.train()
    model.path_flags = N.PATH_OPT_DEC_UNHOISTED
    want = _step(model, crit, batch)
    model.path_flags = N.PATH_OPT_PERSIST_ALL
    for rnd in range(12):
        _assert_same(_step(model, crit, batch), want, 'B=%d run %d' % (B, rnd))


@pytest.mark.parametrize('name', ['mid', 'c2', 'c3'])
def test_deep_ring_cell_products_match_the_shallow_kernel_bit_for_bit(dev, name):
    """RFN_PATH_OPT_DEEP_CELLS (A/B hook) puts per-step products whose tiles do not outnumber the CUs -- all of them at the
    golden tiers' batch sizes -- on the deep-ring kernel (8 slots, the whole K range of K <= 512 in flight, barrier-free K
    loop; csrc/rfn_cellgemm.hip cell_gemm_deep_k) instead of the 3-slot kernel: same pieces, same k order, so forward, loss,
    every gradient and the greedy decode agree bit for bit."""
    import recurrent_fusion_network_amd as R
    N = R._native
    cfg, spec, P, batch, gold = load_case(name)
    batch = to_dev(batch, dev)
    crit = R.ReviewNetEnsembleCriterion(cfg)
    shallow = build(cfg, P, dev, train=True)
    deep = build(cfg, P, dev, train=True)
    deep.path_flags = N.PATH_OPT_DEEP_CELLS
    _assert_same(_step(deep, crit, batch), _step(shallow, crit, batch), name)
    shallow.eval()
    deep.eval()
    with torch.no_grad():
        a = shallow.sample(batch[0], batch[1], {'sample_max': 1})
        b = deep.sample(batch[0], batch[1], {'sample_max': 1})
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


@pytest.mark.parametrize('name', ['mid', 'c2', 'c3'])
def test_sixteen_row_tiles_in_the_path_match_the_32_row_tiles(dev, name):
    """At the golden tiers' batch sizes every per-step product has fewer 16-row tiles than the chip has CUs, so the library
    takes variant 4 (16-row tiles on the 16x16x4 MFMA shape); RFN_PATH_OPT_NO_SMALL_TILES keeps 32-row tiles.  Same fma chains:
    forward, loss, every gradient and the greedy decode agree bit for bit."""
    import recurrent_fusion_network_amd as R
    N = R._native
    cfg, spec, P, batch, gold = load_case(name)
    batch = to_dev(batch, dev)
    crit = R.ReviewNetEnsembleCriterion(cfg)
    big = build(cfg, P, dev, train=True)
    big.path_flags = N.PATH_OPT_NO_SMALL_TILES
    small = build(cfg, P, dev, train=True)
    _assert_same(_step(small, crit, batch), _step(big, crit, batch), name)
    big.eval()
    small.eval()
    with torch.no_grad():
        a = big.sample(batch[0], batch[1], {'sample_max': 1})
        b = small.sample(batch[0], batch[1], {'sample_max': 1})
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])

"""The decoder cell with z2h hoisted through the attention (csrc/rfn_deccell.hip, C ABI rfn_dec_cell_fwd / rfn_dec_attn_bwd /
rfn_dec_du) against an fp64 restatement of the reference's own, UN-hoisted formulas
(misc/LSTMSoftAttentionCore.py:60-102: z = bmm(att_seq_t, alpha); all_input_sums = i2h(xt) + h2h(pre_h) + z2h(z))."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def N():
    import recurrent_fusion_network_amd._native as n
    return n


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


def maxerr(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def ref_cell(v, Wa, ba, hp, w, bo, Wz, bz, gin, c0, maxout):
    """fp64, the reference's order of operations.  v (Bc, L, R) thought vectors; rows b of the cell read v[b // row_div]
    (the caller expands).  -> alpha, gate activations (as rfn_lstm_fwd leaves them), c1, h1."""
    R = c0.size(1)
    proj = v @ Wa.t() + ba                                          # att_2_att_h(att)              :64-66
    e = (torch.tanh(proj + hp[:, None, :]) @ w) + bo                # h_2_att_h(h) is `hp`          :68-75
    alpha = torch.softmax(e, dim=1)                                 #                                 :76
    z = (v * alpha[:, :, None]).sum(1)                              # bmm(att_seq_t, alpha)         :78-79
    sums = gin + (z @ Wz.t() + bz)                                  # i2h + h2h (given) + z2h(z)    :81
    sig = torch.sigmoid(sums[:, :3 * R])
    g = torch.max(sums[:, 3 * R:4 * R], sums[:, 4 * R:]) if maxout else torch.tanh(sums[:, 3 * R:])
    c1 = sig[:, R:2 * R] * c0 + sig[:, :R] * g
    h1 = sig[:, 2 * R:] * torch.tanh(c1)
    return alpha, sig, g, c1, h1, sums


@pytest.mark.parametrize('B,L,A,R,maxout,row_div', [(6, 8, 512, 512, 0, 1), (5, 8, 64, 96, 1, 1), (4, 5, 30, 18, 0, 1),
                                                    (12, 8, 128, 64, 0, 3), (3, 11, 17, 7, 1, 1), (300, 8, 64, 128, 0, 5),
                                                    (40, 8, 256, 256, 0, 5), (5, 8, 64, 256, 1, 1), (300, 7, 128, 256, 1, 1),
                                                    (70, 3, 512, 768, 0, 1)])
def test_hoisted_decoder_cell_matches_the_references_formulas_in_fp64(dev, B, L, A, R, maxout, row_div):
    n = N()
    NG = 5 if maxout else 4
    GD = NG * R
    Bc = B // row_div
    v = rnd(Bc, L, R, seed=1)
    Wa, ba = rnd(A, R, seed=2, scale=0.1), rnd(A, seed=3, scale=0.1)
    Wz, bz = rnd(GD, R, seed=4, scale=0.1), rnd(GD, seed=5, scale=0.1)
    hp, w, bo = rnd(B, A, seed=6), rnd(A, seed=7, scale=0.3), rnd(1, seed=8)
    gin, c0 = rnd(B, GD, seed=9), rnd(B, R, seed=10)
    if maxout:   # three units whose two candidate chunks tie exactly
        gin[:, 4 * R:4 * R + 3] = gin[:, 3 * R:3 * R + 3]
        Wz[4 * R:4 * R + 3] = Wz[3 * R:3 * R + 3]
        bz[4 * R:4 * R + 3] = bz[3 * R:3 * R + 3]
    # time-major device operands as the path lays them out: proj / U are (L, Bc, .)
    proj_tm = (v.double() @ Wa.double().t() + ba.double()).float().transpose(0, 1).contiguous()      # (L, Bc, A)
    U_tm = (v.double() @ Wz.double().t()).float().transpose(0, 1).contiguous()                       # (L, Bc, GD), no bias
    d = lambda t: t.to(dev).contiguous()
    projd, Ud, hpd, wd, bod, bzd, c0d = d(proj_tm), d(U_tm), d(hp), d(w), d(bo), d(bz), d(c0)
    gd = d(gin)
    c1d, h1d, ald = torch.empty(B, R, device=dev), torch.empty(B, R, device=dev), torch.empty(B, L, device=dev)
    st = n.stream_ptr()
    n.check(n.lib.rfn_dec_cell_fwd(projd.data_ptr(), A, Bc * A, hpd.data_ptr(), wd.data_ptr(), bod.data_ptr(), Ud.data_ptr(), GD,
                                   Bc * GD, bzd.data_ptr(), gd.data_ptr(), GD, c0d.data_ptr(), R, c1d.data_ptr(), R,
                                   h1d.data_ptr(), R, ald.data_ptr(), B, L, A, R, maxout, row_div, 0.0, 0, 0, st), 'dec_cell_fwd')
    # the fp64 reference consumes the f32 projection the kernel was given (the products upstream are tested elsewhere)
    vx = v.double().repeat_interleave(row_div, 0)
    dd = lambda t: t.double()
    alpha, sig, g, c1, h1, sums = ref_cell(vx, dd(Wa), dd(ba), dd(hp), dd(w), dd(bo), dd(Wz), dd(bz), dd(gin), dd(c0), maxout)
    assert maxerr(ald, alpha) < 3e-6
    assert maxerr(gd[:, :3 * R], sig) < 3e-6 and maxerr(gd[:, 3 * R:4 * R], g) < 1e-5
    assert maxerr(c1d, c1) < 1e-5 and maxerr(h1d, h1) < 1e-5
    if maxout:   # chunk 4 keeps the selector, ties split
        s3, s4 = sums[:, 3 * R:4 * R], sums[:, 4 * R:]
        clear = (s3 - s4).abs() > 1e-4
        sel = torch.where(s3 > s4, 1.0, 0.0)
        assert torch.equal(gd[:, 4 * R:].cpu()[clear].double(), sel[clear])
        assert float((gd[:, 4 * R:4 * R + 3] - 0.5).abs().max()) == 0.0

    # ---- backward of the attention from the gate gradients, against autograd through the UN-hoisted formulas -----------------
    if row_div != 1:
        return
    dg = rnd(B, GD, seed=11)
    vr = v.double().requires_grad_(True)
    pr = proj_tm.transpose(0, 1).double().requires_grad_(True)     # d proj is asked for directly (a leaf)
    hr, wr = dd(hp).requires_grad_(True), dd(w).requires_grad_(True)
    e = (torch.tanh(pr + hr[:, None, :]) @ wr) + dd(bo)
    al = torch.softmax(e, dim=1)
    z = (vr.detach() * al[:, :, None]).sum(1)
    ((z @ dd(Wz).t()) * dd(dg)).sum().backward()
    dgd = d(dg)
    dproj = torch.ones(L, B, A, device=dev)
    dhp, dwp = torch.empty(B, A, device=dev), torch.empty(B, A, device=dev)
    n.check(n.lib.rfn_dec_attn_bwd(projd.data_ptr(), A, B * A, hpd.data_ptr(), wd.data_ptr(), ald.data_ptr(), Ud.data_ptr(), GD,
                                   B * GD, dgd.data_ptr(), GD, B, L, A, GD, dproj.data_ptr(), A, B * A, 1, dhp.data_ptr(),
                                   dwp.data_ptr(), st), 'dec_attn_bwd')
    scale = max(1.0, float(pr.grad.abs().max()))
    assert maxerr(dproj.transpose(0, 1), pr.grad + 1.0) < 3e-5 * scale
    assert maxerr(dhp, hr.grad) < 1e-4 * scale
    assert maxerr(dwp.sum(0), wr.grad) < 1e-4 * max(1.0, float(wr.grad.abs().max()))
    # overwrite form
    n.check(n.lib.rfn_dec_attn_bwd(projd.data_ptr(), A, B * A, hpd.data_ptr(), wd.data_ptr(), ald.data_ptr(), Ud.data_ptr(), GD,
                                   B * GD, dgd.data_ptr(), GD, B, L, A, GD, dproj.data_ptr(), A, B * A, 0, dhp.data_ptr(),
                                   dwp.data_ptr(), st), 'dec_attn_bwd')
    assert maxerr(dproj.transpose(0, 1), pr.grad) < 3e-5 * scale


@pytest.mark.parametrize('S,B,L,GD', [(17, 9, 8, 2048), (3, 5, 11, 140), (1, 2, 1, 4), (4, 3, 8, 35)])
def test_du_is_the_sum_over_steps_of_alpha_times_dgates(dev, S, B, L, GD):
    n = N()
    al, dg = rnd(S, B, L, seed=1), rnd(S, B, GD, seed=2)
    ald, dgd = al.to(dev), dg.to(dev)
    dU = torch.full((L, B, GD), 7.0, device=dev)
    n.check(n.lib.rfn_dec_du(ald.data_ptr(), dgd.data_ptr(), S, B, L, GD, dU.data_ptr(), GD, B * GD, n.stream_ptr()), 'dec_du')
    ref = torch.einsum('sbl,sbg->lbg', al.double(), dg.double())
    assert maxerr(dU, ref) < 2e-5


def test_a_rows_result_does_not_depend_on_the_batch_it_sits_in_and_dropout_is_the_published_mask(dev):
    """The launch shape follows the batch size (64-, 128- or 256-unit blocks); a row's bits must not.  With drop_p > 0 the
    kept units are those of rfn_dropout_mask(seed, offset) -- the mask the backward twin regenerates."""
    n = N()
    L, A, R, p, seed, off = 8, 96, 256, 0.25, 1234, (1 << 32) + 3
    GD = 4 * R
    outs = {}
    for B in (2, 40, 600):
        proj, U = rnd(L, B, A, seed=1), rnd(L, B, GD, seed=2, scale=0.2)
        hp, w, bo, bz = rnd(B, A, seed=3), rnd(A, seed=4, scale=0.3), rnd(1, seed=5), rnd(GD, seed=6, scale=0.1)
        gin, c0 = rnd(B, GD, seed=7), rnd(B, R, seed=8)
        # rows 0 and 1 of every batch are the same two rows
        ref_rows = outs.get('rows')
        if ref_rows is None:
            outs['rows'] = ref_rows = (proj[:, :2].clone(), U[:, :2].clone(), hp[:2].clone(), gin[:2].clone(), c0[:2].clone())
        proj[:, :2], U[:, :2], hp[:2], gin[:2], c0[:2] = ref_rows
        d = lambda t: t.to(dev).contiguous()
        projd, Ud, hpd, wd, bod, bzd, gd, c0d = d(proj), d(U), d(hp), d(w), d(bo), d(bz), d(gin), d(c0)
        c1d, h1d, ald = torch.empty(B, R, device=dev), torch.empty(B, R, device=dev), torch.empty(B, L, device=dev)
        n.check(n.lib.rfn_dec_cell_fwd(projd.data_ptr(), A, B * A, hpd.data_ptr(), wd.data_ptr(), bod.data_ptr(), Ud.data_ptr(), GD,
                                       B * GD, bzd.data_ptr(), gd.data_ptr(), GD, c0d.data_ptr(), R, c1d.data_ptr(), R,
                                       h1d.data_ptr(), R, ald.data_ptr(), B, L, A, R, 0, 1, p, seed, off, n.stream_ptr()), 'fwd')
        outs[B] = (gd[:2].clone(), c1d[:2].clone(), ald[:2].clone(), h1d.clone())
        keep = torch.empty(B * R, device=dev)
        n.check(n.lib.rfn_dropout_mask(seed, off, B * R, p, keep.data_ptr(), n.stream_ptr()), 'mask')
        og, tc = gd[:, 2 * R:3 * R], torch.tanh(c1d)
        expect = og * tc * keep.view(B, R) / (1 - p)
        assert maxerr(h1d, expect) < 1e-6
        assert torch.equal(h1d == 0, (keep.view(B, R) == 0) | (og * tc == 0))
    for B in (40, 600):
        for a, b in zip(outs[2][:3], outs[B][:3]):
            assert torch.equal(a, b)


def test_bad_arguments_are_refused(dev):
    n = N()
    t = torch.zeros(64, device=dev)
    p = t.data_ptr()
    st = n.stream_ptr()
    assert n.lib.rfn_dec_cell_fwd(p, 4, 4, p, p, p, p, 16, 16, p, p, 16, p, 4, p, 4, p, 4, p, 1, 2000, 4, 4, 0, 1, 0.0, 0, 0, st) != 0
    assert n.lib.rfn_dec_cell_fwd(p, 4, 4, p, p, p, None, 16, 16, p, p, 16, p, 4, p, 4, p, 4, p, 1, 1, 4, 4, 0, 1, 0.0, 0, 0, st) != 0
    assert n.lib.rfn_dec_attn_bwd(p, 4, 4, p, p, p, p, 16, 16, p, 16, 1, 1025, 4, 16, p, 4, 4, 0, p, p, st) != 0
    assert n.lib.rfn_dec_du(p, p, 1, 1, 0, 8, p, 8, 8, st) != 0


@pytest.mark.parametrize('name', ['mid', 'c2', 'tinymax', 'odd'])
@pytest.mark.parametrize('train', [False, True])
def test_hoisted_and_three_launch_decoder_cells_are_the_same_mathematics(dev, name, train):
    """RFN_PATH_OPT_DEC_UNHOISTED (the decoder cell of rounds 3-5: z2h(z) as a per-step product) against the default (z2h hoisted
    through the attention) on reference-generated tiers -- `tinymax` has 5R-wide maxout gates, `odd` widths that are no multiple
    of 4 (the scalar kernels), `train` adds dropout 0.3 on the decoder (the same Philox masks in both forms).  Same mathematics,
    different association of the z2h sum: log-probs within 2e-5, loss within 1e-5 relative, every gradient within 1e-6 + 3e-4 of
    its tensor's max, greedy ids identical.  (Each form is also checked against the oracle and the goldens on its own.)"""
    import recurrent_fusion_network_amd as R
    from conftest import load_case
    from test_model_gpu import build, to_dev
    N_ = R._native
    cfg, spec, P, batch, gold = load_case(name)
    if train:
        cfg.drop_prob_lm = 0.3
    batch = to_dev(batch, dev)
    fc, att, labels, masks, top = batch
    crit = R.ReviewNetEnsembleCriterion(cfg)
    outs = []
    for flags in (0, N_.PATH_OPT_DEC_UNHOISTED):
        model = build(cfg, P, dev, train=train)
        model.path_flags = flags
        torch.manual_seed(9)                      # the dropout seed both forms draw
        model.zero_grad()
        lp, reason = model(fc, att, labels)
        loss = crit(lp, labels[:, 1:], masks[:, 1:], reason, top, 1.0)
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        model.eval()
        with torch.no_grad():
            seq = model.sample(fc, att, {'sample_max': 1})[0]
        outs.append((lp.detach(), float(loss.detach()), grads, seq))
    (lp0, l0, g0, s0), (lp1, l1, g1, s1) = outs
    assert not torch.equal(lp0, lp1) or name == 'odd', 'the flag must change the arithmetic'
    assert float((lp0 - lp1).abs().max()) < 2e-5
    assert abs(l0 - l1) < 1e-5 * max(1.0, abs(l0))
    for k in g0:
        err = float((g0[k] - g1[k]).abs().max())
        assert err <= 1e-6 + 3e-4 * float(g0[k].abs().max()), (k, err)
    assert torch.equal(s0, s1)

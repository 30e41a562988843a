"""Parity of every primitive HIP operator (called through the C ABI) against an fp64 PyTorch-CPU
restatement of the same math.  fp32 kernels, tolerances are stated per test."""
import ctypes as C

import numpy as np
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


# ------------------------------------------------------------------------------------------------
# GEMM: all four operand layouts, vector and scalar staging, ragged tails, K segments, groups
# ------------------------------------------------------------------------------------------------
GEMM_SHAPES = [
    # M, N, K          (tile tails on every axis; K < 32; K not multiple of 32)
    (256, 512, 512), (130, 70, 100), (64, 64, 32), (3, 51, 16), (200, 2048, 40), (1000, 24, 260),
    (50, 36, 7),     # K % 4 != 0 -> scalar staging
    (33, 45, 64),    # N % 4 != 0 for transposed B -> scalar staging
    (1024, 768, 96),  # >= 384 tiles of 128 -> big tile
]


@pytest.mark.parametrize('ak', [1, 0])
@pytest.mark.parametrize('bk', [1, 0])
@pytest.mark.parametrize('shape', GEMM_SHAPES)
def test_gemm_layouts(dev, shape, ak, bk):
    n = N()
    M, Nn, K = shape
    A = rnd(M, K, seed=1)          # logical A[m,k]
    Bm = rnd(Nn, K, seed=2)        # logical B[n,k]
    bias = rnd(Nn, seed=3)
    ref = A.double() @ Bm.double().t() + bias.double()
    A_st = (A if ak else A.t().contiguous()).to(dev)
    B_st = (Bm if bk else Bm.t().contiguous()).to(dev)
    lda = K if ak else M
    ldb = K if bk else Nn
    Cd = torch.full((M, Nn), float('nan'), device=dev)
    n.gemm(M, Nn, [(Cd, Nn, [(A_st, lda, ak, B_st, ldb, bk, K, bias.to(dev))])])
    tol = 2e-6 * K * 1.0 + 1e-5
    assert maxerr(Cd, ref) < tol
    # accumulate on top of existing C, no bias, with a padded ldc
    Cpad = torch.zeros(M, Nn + 8, device=dev)
    Cpad[:, :Nn] = 1.5
    n.gemm(M, Nn, [(Cpad, Nn + 8, [(A_st, lda, ak, B_st, ldb, bk, K, None)])], accumulate=True)
    assert maxerr(Cpad[:, :Nn], A.double() @ Bm.double().t() + 1.5) < tol
    assert float(Cpad[:, Nn:].abs().max()) == 0.0


def test_gemm_segments_and_groups(dev):
    n = N()
    M, Nn = 96, 160
    Ks = [64, 40, 128]
    groups = []
    refs = []
    for g in range(3):
        segs = []
        ref = torch.zeros(M, Nn, dtype=torch.float64)
        for s, K in enumerate(Ks):
            A, Bm, b = rnd(M, K, seed=10 * g + s), rnd(Nn, K, seed=50 + 10 * g + s), rnd(Nn, seed=90 + s)
            ref += A.double() @ Bm.double().t() + b.double()
            segs.append((A.to(dev), K, 1, Bm.to(dev), K, 1, K, b.to(dev)))
        Cd = torch.empty(M, Nn, device=dev)
        groups.append((Cd, Nn, segs))
        refs.append(ref)
    n.gemm(M, Nn, groups)
    for (Cd, _, _), ref in zip(groups, refs):
        assert maxerr(Cd, ref) < 1e-3


@pytest.mark.parametrize('ak,bk', [(1, 1), (1, 0), (0, 0)])
def test_gemm_split_k_is_deterministic_and_correct(dev, ak, bk):
    """Skinny problems (M = batch) with a workspace are cut along K across blocks; the fixed-order reduce must
    give the same bits on every call and match fp64 within fp32 re-association error."""
    n = N()
    ws = torch.empty(48 << 20, dtype=torch.uint8, device=dev)
    M, Nn = 256, 512
    Ks = [2048, 512, 96]                       # three K segments, the last one ragged vs the split
    segs, ref = [], torch.zeros(M, Nn, dtype=torch.float64)
    keep = []
    for s_, K in enumerate(Ks):
        A, Bm, b = rnd(M, K, seed=s_), rnd(Nn, K, seed=10 + s_), rnd(Nn, seed=20 + s_)
        ref += A.double() @ Bm.double().t() + b.double()
        A_st = (A if ak else A.t().contiguous()).to(dev)
        B_st = (Bm if bk else Bm.t().contiguous()).to(dev)
        bd = b.to(dev)
        keep += [A_st, B_st, bd]
        segs.append((A_st, K if ak else M, ak, B_st, K if bk else Nn, bk, K, bd))
    outs = []
    for rep in range(2):
        Cd = torch.full((M, Nn), 0.25, device=dev)
        n.gemm(M, Nn, [(Cd, Nn, segs)], accumulate=True, ws=ws)
        outs.append(Cd)
    assert torch.equal(outs[0], outs[1])
    assert maxerr(outs[0], ref + 0.25) < 2e-3
    # and the unsplit path agrees to rounding
    C0 = torch.full((M, Nn), 0.25, device=dev)
    n.gemm(M, Nn, [(C0, Nn, segs)], accumulate=True)
    assert maxerr(outs[0], C0) < 1e-3
    # grouped + split, tails on M and N
    M2, N2, K2 = 200, 130, 1024
    groups, refs = [], []
    for g in range(3):
        A, Bm = rnd(M2, K2, seed=30 + g), rnd(N2, K2, seed=40 + g)
        Ad, Bd = A.to(dev), Bm.to(dev)
        keep += [Ad, Bd]
        Cg = torch.empty(M2, N2, device=dev)
        groups.append((Cg, N2, [(Ad, K2, 1, Bd, K2, 1, K2, None)]))
        refs.append(A.double() @ Bm.double().t())
    n.gemm(M2, N2, groups, ws=ws)
    for (Cg, _, _), r in zip(groups, refs):
        assert maxerr(Cg, r) < 1e-3


@pytest.mark.parametrize('M,Nn,K,ak,bk,groups', [
    (4352, 512, 9488, 1, 0, 1),      # logit dX at C3: 136 tiles, the old rule's 544-block launch
    (1088, 512, 9488, 1, 0, 1),      # ... at C2: 36 tiles
    (9472, 512, 1088, 0, 0, 1),      # logit dW at C2: 296 tiles
    (4096, 1536, 4096, 0, 0, 1),     # a heterogeneous encoder's att_2_att_h weight gradient: 384 tiles (K shortened)
    (4096, 2176, 3136, 0, 0, 1),     # ... 544 tiles: between one and two rounds
    (256, 2048, 2048, 1, 0, 4),      # stage-I dH: one full round, 4 groups
])
def test_gemm_medium_products_split_by_the_cost_model(dev, M, Nn, K, ak, bk, groups):
    """Products below two rounds of 128 x 128 tiles take their K split from launch_tile's cost model (tools/split_probe.py):
    every choice must match fp64, be deterministic, not depend on the LDS-lean flag (data-parallel hosts set it: the step
    must stay bit-equal), and a forced different split must agree to re-association error."""
    n = N()
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev).manual_seed(M + Nn + K)
    probs, refs, keep = [], [], []
    for _ in range(groups):
        A = torch.randn((M, K) if ak else (K, M), device=dev, generator=g) * 0.1
        B = torch.randn((Nn, K) if bk else (K, Nn), device=dev, generator=g) * 0.1
        keep += [A, B]
        Am = A if ak else A.t()
        Bm = B.t() if bk else B
        refs.append(Am.double() @ Bm.double())
        probs.append([(A, K if ak else M, ak, B, K if bk else Nn, bk, K, None)])

    def run(flags):
        Cs = [torch.empty(M, Nn, device=dev) for _ in range(groups)]
        n.gemm(M, Nn, [(C, Nn, sg) for C, sg in zip(Cs, probs)], ws=ws, flags=flags)
        return Cs
    base = run(0)
    tol = 3e-6 * K ** 0.5
    for C, r in zip(base, refs):
        assert maxerr(C, r) < tol
    for other in (run(0), run(n.GEMM_OPT_LDS_LEAN)):
        for C, C0 in zip(other, base):
            assert torch.equal(C, C0)
    for forced in (1, 2, 5):
        for C, r in zip(run(forced << 8), refs):
            assert maxerr(C, r) < tol


def test_gemm_weight_gradient_emits_bias_gradient(dev):
    """dW = dY^T X with a_colsum = colsum(dY) riding on the same launch (both tile sizes, vec and scalar)."""
    n = N()
    for rows, Nn, K in [(256, 2048, 512), (300, 70, 36), (37, 51, 16)]:
        dY, X = rnd(rows, Nn, seed=1), rnd(rows, K, seed=2)
        dYd, Xd = dY.to(dev), X.to(dev)
        dW = torch.empty(Nn, K, device=dev)
        db = torch.full((Nn,), float('nan'), device=dev)
        n.gemm(Nn, K, [(dW, K, [(dYd, Nn, 0, Xd, K, 0, rows, None)], db)])
        assert maxerr(dW, dY.double().t() @ X.double()) < 1e-3
        assert maxerr(db, dY.double().sum(0)) < 1e-4


@pytest.mark.parametrize('rows,Nn,K,groups', [(4352, 512, 512, 1), (4352, 2048, 512, 1), (1000, 70, 36, 2),
                                               (4352, 300, 130, 3), (256, 2048, 512, 2)])
def test_gemm_weight_gradient_split_k_keeps_the_bias_rider(dev, rows, Nn, K, groups):
    """Weight gradients of the step-shared weights reduce over S*B rows with few output tiles: they are cut
    along the reduction, and the a_colsum rider is finished by the same fixed-order reduce (deterministic,
    accumulate honoured for both outputs)."""
    n = N()
    ws = torch.empty(48 << 20, dtype=torch.uint8, device=dev)
    keep, probs, refs = [], [], []
    for g in range(groups):
        dY, X = rnd(rows, Nn, seed=1 + g), rnd(rows, K, seed=20 + g)
        dYd, Xd = dY.to(dev), X.to(dev)
        keep += [dYd, Xd]
        dW = torch.full((Nn, K), 0.5, device=dev)
        db = torch.full((Nn,), -2.0, device=dev)
        probs.append((dW, K, [(dYd, Nn, 0, Xd, K, 0, rows, None)], db))
        refs.append((dY.double().t() @ X.double(), dY.double().sum(0)))
    n.gemm(Nn, K, probs, accumulate=True, ws=ws)
    first = [(p[0].clone(), p[3].clone()) for p in probs]
    for (dW, db), (rW, rb) in zip(first, refs):
        assert maxerr(dW, rW + 0.5) < 3e-3
        assert maxerr(db, rb - 2.0) < 1e-3
    for p in probs:
        p[0].fill_(0.5)
        p[3].fill_(-2.0)
    n.gemm(Nn, K, probs, accumulate=True, ws=ws)
    for p, (dW, db) in zip(probs, first):
        assert torch.equal(p[0], dW) and torch.equal(p[3], db)


@pytest.mark.parametrize('rows,Nn,K,groups', [(4352, 512, 512, 1), (2048, 2048, 96, 3), (300, 200, 52, 2), (50176, 512, 256, 2)])
def test_gemm_split_k_finished_in_kernel_keeps_the_bias_rider(dev, rows, Nn, K, groups):
    """rfn_gemm_f32_tk on weight gradients (reduction over `rows`, a_colsum rider): the last K range to arrive sums the
    partial tiles and the partial column sums in K-range order -- same bits as the separate reduce kernel, repeated
    launches on the same counters (left at zero each time), also under accumulate."""
    n = N()
    ws = torch.empty(48 << 20, dtype=torch.uint8, device=dev)
    tickets = torch.zeros(16384, dtype=torch.int32, device=dev)
    dY = [rnd(rows, Nn, seed=10 + g).to(dev) for g in range(groups)]
    X = [rnd(rows, K, seed=20 + g).to(dev) for g in range(groups)]

    def run(tk, acc):
        dW = [torch.full((Nn, K), 0.5, device=dev) for _ in range(groups)]
        db = [torch.full((Nn,), 0.25, device=dev) for _ in range(groups)]
        probs = [(dW[g], K, [(dY[g], Nn, 0, X[g], K, 0, rows, None)], db[g]) for g in range(groups)]
        n.gemm(Nn, K, probs, accumulate=acc, ws=ws, tickets=tk)
        return dW, db

    for acc in (False, True):
        ref_w, ref_b = run(None, acc)
        for rep in range(3):
            got_w, got_b = run(tickets, acc)
            assert int(tickets.abs().sum()) == 0
            for g in range(groups):
                assert torch.equal(got_w[g], ref_w[g]) and torch.equal(got_b[g], ref_b[g]), (acc, rep, g)
    want = dY[0].double().t() @ X[0].double()
    assert maxerr(ref_w[0] - 0.5, want) < 1e-5 + 3e-6 * rows


def test_gemm_half_height_tail_round(dev):
    """Big NT launches whose tile count leaves the last round at most half full process that round as half-height
    tiles inside the same launch -- as quarter tiles when it is at most a quarter full -- (rfn_gemm.hip, TAIL): 3200 tiles =
    400 per XCD = 6 full rounds of 64 + 16 on the LDS-DMA kernel (quarter tiles), other splits on the kernels below, a
    ragged last band, grouped problems -- every output element against fp64, and bit-identical from call to call."""
    n = N()
    M, Nn, K, G = 12800, 2048, 64, 2          # 100 x 16 tiles x 2 groups = 3200
    A = rnd(M, K, seed=1)
    Ad = A.to(dev)
    probs, refs, keep = [], [], []
    for g in range(G):
        Bm, b = rnd(Nn, K, seed=10 + g), rnd(Nn, seed=20 + g)
        Bd, bd = Bm.to(dev), b.to(dev)
        keep += [Bd, bd]
        Cg = torch.full((M, Nn), float('nan'), device=dev)
        probs.append((Cg, Nn, [(Ad, K, 1, Bd, K, 1, K, bd)]))
        refs.append(A.double() @ Bm.double().t() + b.double())
    n.gemm(M, Nn, probs)
    first = [p[0].clone() for p in probs]
    for Cg, r in zip(first, refs):
        assert not bool(torch.isnan(Cg).any())
        assert maxerr(Cg, r) < 2e-4
    n.gemm(M, Nn, probs)
    for p, c in zip(probs, first):
        assert torch.equal(p[0], c)
    # the LDS-DMA kernel (default) and the register-staged round-1 kernel walk k in the same order: bit-identical,
    # half-height tail tiles included (the slot count per CU differs between them, so do their tail rounds)
    for flags in (n.GEMM_OPT_NO_DMA, n.GEMM_OPT_LDS_LEAN):
        for p in probs:
            p[0].fill_(float('nan'))
        n.gemm(M, Nn, probs, flags=flags)
        for p, c in zip(probs, first):
            assert torch.equal(p[0], c)


@pytest.mark.parametrize('ak,bk', [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_gemm_big_tile_variants_are_bit_identical(dev, ak, bk):
    """Interior big-tile problems have three kernels behind one entry point: LDS-DMA staging (default), register
    staging (RFN_GEMM_OPT_NO_DMA) and the single-buffered register-staged tiles data-parallel hosts ask for
    (RFN_GEMM_OPT_LDS_LEAN).  All keep the same k order per output element, so their results must be bit-identical --
    and right (fp64), for every operand layout, with K segments, groups, accumulate and a half-height tail round."""
    n = N()
    M, Nn, G = 12800, 512, 2                 # 100 x 4 tiles x 2 groups = 800 tiles of 128 x 128
    Ks = [64, 96]
    if not (ak and bk):
        M = 1536                               # 12 x 4 x 2 = 96 big tiles would not take the big path: widen N instead
        Nn = 4096
    probs, refs, keep = [], [], []
    for g in range(G):
        segs, ref = [], torch.zeros(M, Nn, dtype=torch.float64)
        for s_, K in enumerate(Ks):
            A, Bm, b = rnd(M, K, seed=7 * g + s_), rnd(Nn, K, seed=50 + 7 * g + s_), rnd(Nn, seed=90 + s_)
            ref += A.double() @ Bm.double().t() + b.double()
            A_st = (A if ak else A.t().contiguous()).to(dev)
            B_st = (Bm if bk else Bm.t().contiguous()).to(dev)
            bd = b.to(dev)
            keep += [A_st, B_st, bd]
            segs.append((A_st, K if ak else M, ak, B_st, K if bk else Nn, bk, K, bd))
        probs.append([None, Nn, segs])
        refs.append(ref)
    outs = {}
    for name, flags in (('dma', 0), ('reg', n.GEMM_OPT_NO_DMA), ('lean', n.GEMM_OPT_LDS_LEAN)):
        res = []
        for p in probs:
            p[0] = torch.full((M, Nn), 0.5, device=dev)
            res.append(p[0])
        n.gemm(M, Nn, [tuple(p) for p in probs], accumulate=True, flags=flags)
        outs[name] = res
    for g in range(G):
        assert maxerr(outs['dma'][g], refs[g] + 0.5) < 2e-4
        assert torch.equal(outs['dma'][g], outs['reg'][g]) and torch.equal(outs['dma'][g], outs['lean'][g])


@pytest.mark.parametrize('ak,bk', [(True, True), (True, False), (False, False)])
@pytest.mark.parametrize('with_ws', [False, True])
def test_gemm_small_tile_dma_is_bit_identical_to_register_staging(dev, ak, bk, with_ws):
    """Interior skinny problems (M = batch rows, 64 x 64 tiles, split over K when a scratch buffer is given) take the LDS-DMA
    ring as well; RFN_GEMM_OPT_NO_DMA selects the register-staged kernel.  Same k order: same bits, and right (fp64)."""
    n = N()
    M, Nn, Ks = 256, 2048, [512, 96]
    segs, ref, keep = [], torch.zeros(M, Nn, dtype=torch.float64), []
    for s_, K in enumerate(Ks):
        A, Bm, b = rnd(M, K, seed=3 + s_), rnd(Nn, K, seed=30 + s_), rnd(Nn, seed=60 + s_)
        ref += A.double() @ Bm.double().t() + b.double()
        A_st = (A if ak else A.t().contiguous()).to(dev)
        B_st = (Bm if bk else Bm.t().contiguous()).to(dev)
        bd = b.to(dev)
        keep += [A_st, B_st, bd]
        segs.append((A_st, K if ak else M, ak, B_st, K if bk else Nn, bk, K, bd))
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=dev) if with_ws else None
    outs = []
    for flags in (0, n.GEMM_OPT_NO_DMA):
        out = torch.full((M, Nn), 0.25, device=dev)
        n.gemm(M, Nn, [(out, Nn, segs)], accumulate=True, ws=ws, flags=flags)
        outs.append(out)
    assert maxerr(outs[0], ref + 0.25) < 2e-4
    assert torch.equal(outs[0], outs[1])


def test_gemm_strided_views_like_the_path(dev):
    """The path feeds column blocks of wider buffers (lda > K, ldc > N): e.g. encoder i's slice of H."""
    n = N()
    B, MR, R, A_ = 70, 4 * 64, 64, 48
    H = rnd(B, MR, seed=5).to(dev)
    W = rnd(A_, R, seed=6).to(dev)
    out = torch.zeros(B, 3 * A_, device=dev)
    n.gemm(B, A_, [(out[:, A_:], 3 * A_, [(H[:, 2 * R:], MR, 1, W, R, 1, R, None)])])
    ref = H[:, 2 * R:3 * R].double().cpu() @ W.double().cpu().t()
    assert maxerr(out[:, A_:2 * A_], ref) < 1e-4
    assert float(out[:, :A_].abs().max()) == 0.0 and float(out[:, 2 * A_:].abs().max()) == 0.0


def test_gemm_rejects_bad_arguments(dev):
    n = N()
    with pytest.raises(n.RfnError):
        n.check(n.lib.rfn_gemm_f32(8, 8, 0, None, 0, None), 'gemm')
    with pytest.raises(n.RfnError):
        n.check(n.lib.rfn_gemm_f32(8, 8, 99, None, 0, None), 'gemm')


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def attn_ref(proj, hp, w, bo, x):
    e = torch.tanh(proj.double() + hp.double()[:, None, :])
    s = e @ w.double() + bo.double()
    al = torch.softmax(s, 1)
    z = (al[:, :, None] * x.double()).sum(1)
    return al, z


@pytest.mark.parametrize('B,L,A,D', [(5, 196, 512, 256), (3, 7, 16, 24), (4, 49, 30, 18), (2, 8, 64, 64), (6, 300, 96, 40)])
def test_attention_forward_backward(dev, B, L, A, D):
    n = N()
    proj, hp, w, bo = rnd(B, L, A, seed=1), rnd(B, A, seed=2), rnd(A, seed=3, scale=0.3), rnd(1, seed=4)
    x, dz = rnd(B, L, D, seed=5), rnd(B, D, seed=6)
    projd, hpd, wd, bod, xd, dzd = [t.to(dev) for t in (proj, hp, w, bo, x, dz)]
    alpha = torch.empty(B, L, device=dev)
    z = torch.empty(B, D, device=dev)
    st = n.stream_ptr()
    n.check(n.lib.rfn_attn_scores_fwd(projd.data_ptr(), L * A, A, hpd.data_ptr(), wd.data_ptr(), bod.data_ptr(), B, L,
                                      A, alpha.data_ptr(), st))
    n.check(n.lib.rfn_attn_context_fwd(xd.data_ptr(), L * D, D, alpha.data_ptr(), B, L, D, z.data_ptr(), D, st))
    # the two-launch form (softmax inside the context kernel) gives the same bits
    raw, alpha2, z2 = torch.empty(B, L, device=dev), torch.empty(B, L, device=dev), torch.empty(B, D, device=dev)
    n.check(n.lib.rfn_attn_fwd(projd.data_ptr(), L * A, A, hpd.data_ptr(), wd.data_ptr(), bod.data_ptr(),
                               xd.data_ptr(), L * D, D, B, L, A, D, raw.data_ptr(), alpha2.data_ptr(), z2.data_ptr(), D,
                               st))
    assert torch.equal(alpha2, alpha) and torch.equal(z2, z)
    assert n.lib.rfn_attn_fwd(projd.data_ptr(), L * A, A, hpd.data_ptr(), wd.data_ptr(), bod.data_ptr(),
                              xd.data_ptr(), L * D, D, B, L, A, D, alpha2.data_ptr(), alpha2.data_ptr(),
                              z2.data_ptr(), D, st) != 0   # scratch must not alias alpha
    # fp64 autograd reference
    pr, hr, wr, xr = [t.double().requires_grad_(True) for t in (proj, hp, w, x)]
    al_ref, z_ref = attn_ref(pr, hr, wr, bo, xr)
    assert maxerr(alpha, al_ref) < 2e-6
    assert maxerr(z, z_ref) < 1e-5
    z_ref.backward(dz.double())
    dal = torch.empty(B, L, device=dev)
    n.check(n.lib.rfn_attn_context_bwd_dalpha(xd.data_ptr(), L * D, D, dzd.data_ptr(), D, B, L, D, dal.data_ptr(), st))
    dproj = torch.empty(B, L, A, device=dev)
    dhp = torch.empty(B, A, device=dev)
    dwp = torch.empty(B, A, device=dev)
    n.check(n.lib.rfn_attn_scores_bwd(projd.data_ptr(), L * A, A, hpd.data_ptr(), wd.data_ptr(), alpha.data_ptr(),
                                      dal.data_ptr(), B, L, A, dproj.data_ptr(), L * A, A, 0, dhp.data_ptr(),
                                      dwp.data_ptr(), st))
    assert maxerr(dproj, pr.grad) < 2e-5
    assert maxerr(dhp, hr.grad) < 1e-4
    assert maxerr(dwp.sum(0), wr.grad) < 1e-4 * max(1.0, float(wr.grad.abs().max()))
    # the one-launch form (dalpha kept in LDS) gives the same bits, in place too
    dproj_f, dhp_f, dwp_f = torch.empty(B, L, A, device=dev), torch.empty(B, A, device=dev), torch.empty(B, A, device=dev)
    n.check(n.lib.rfn_attn_bwd(projd.data_ptr(), L * A, A, hpd.data_ptr(), wd.data_ptr(), alpha.data_ptr(),
                               xd.data_ptr(), L * D, D, dzd.data_ptr(), D, B, L, A, D, dproj_f.data_ptr(), L * A, A, 0,
                               dhp_f.data_ptr(), dwp_f.data_ptr(), st))
    assert torch.equal(dproj_f, dproj) and torch.equal(dhp_f, dhp) and torch.equal(dwp_f, dwp)
    inpl_f = projd.clone()
    n.check(n.lib.rfn_attn_bwd(inpl_f.data_ptr(), L * A, A, hpd.data_ptr(), wd.data_ptr(), alpha.data_ptr(),
                               xd.data_ptr(), L * D, D, dzd.data_ptr(), D, B, L, A, D, inpl_f.data_ptr(), L * A, A, 0,
                               dhp_f.data_ptr(), dwp_f.data_ptr(), st))
    assert torch.equal(inpl_f, dproj)
    # d att_seq through the context path only (the projection path is a GEMM)
    dx = torch.zeros(B, L, D, device=dev)
    n.check(n.lib.rfn_attn_context_bwd_dseq(alpha.data_ptr(), dzd.data_ptr(), D, B, L, D, dx.data_ptr(), L * D, D, st))
    assert maxerr(dx, al_ref.detach()[:, :, None] * dz.double()[:, None, :]) < 1e-5
    # in-place + accumulate variants
    acc = torch.ones(B, L, A, device=dev)
    n.check(n.lib.rfn_attn_scores_bwd(projd.data_ptr(), L * A, A, hpd.data_ptr(), wd.data_ptr(), alpha.data_ptr(),
                                      dal.data_ptr(), B, L, A, acc.data_ptr(), L * A, A, 1, dhp.data_ptr(),
                                      dwp.data_ptr(), st))
    assert maxerr(acc, pr.grad + 1.0) < 2e-5
    inpl = projd.clone()
    n.check(n.lib.rfn_attn_scores_bwd(inpl.data_ptr(), L * A, A, hpd.data_ptr(), wd.data_ptr(), alpha.data_ptr(),
                                      dal.data_ptr(), B, L, A, inpl.data_ptr(), L * A, A, 0, dhp.data_ptr(),
                                      dwp.data_ptr(), st))
    assert maxerr(inpl, pr.grad) < 2e-5


def test_stage1_attention_grouped_over_encoders_matches_single_calls(dev):
    """rfn_attn_fwd_grouped / rfn_attn_bwd_grouped (blockIdx.z / .y = encoder) against one call per encoder: same bits."""
    n = N()
    G, B, L, A, D = 3, 5, 50, 64, 96
    st = n.stream_ptr()
    proj = [rnd(B, L, A, seed=10 + g).to(dev) for g in range(G)]
    hp = [rnd(B, A, seed=20 + g).to(dev) for g in range(G)]
    w = [rnd(A, seed=30 + g, scale=0.3).to(dev) for g in range(G)]
    bo = [rnd(1, seed=40 + g).to(dev) for g in range(G)]
    x = [rnd(B, L, D, seed=50 + g).to(dev) for g in range(G)]
    dz = [rnd(B, D, seed=60 + g).to(dev) for g in range(G)]
    new = lambda *shape: [torch.empty(*shape, device=dev) for _ in range(G)]  # noqa: E731
    raw, al, z = new(B, L), new(B, L), new(B, D)
    n.check(n.lib.rfn_attn_fwd_grouped(G, n.ptr_array(proj), L * A, A, n.ptr_array(hp), n.ptr_array(w), n.ptr_array(bo),
                                       n.ptr_array(x), L * D, D, B, L, A, D, n.ptr_array(raw), n.ptr_array(al),
                                       n.ptr_array(z), D, st))
    dp, dhp, dwp = new(B, L, A), new(B, A), new(B, A)
    n.check(n.lib.rfn_attn_bwd_grouped(G, n.ptr_array(proj), L * A, A, n.ptr_array(hp), n.ptr_array(w), n.ptr_array(al),
                                       n.ptr_array(x), L * D, D, n.ptr_array(dz), D, B, L, A, D, n.ptr_array(dp), L * A,
                                       A, 0, n.ptr_array(dhp), n.ptr_array(dwp), st))
    for g in range(G):
        raw1, al1, z1 = torch.empty(B, L, device=dev), torch.empty(B, L, device=dev), torch.empty(B, D, device=dev)
        n.check(n.lib.rfn_attn_fwd(proj[g].data_ptr(), L * A, A, hp[g].data_ptr(), w[g].data_ptr(), bo[g].data_ptr(),
                                   x[g].data_ptr(), L * D, D, B, L, A, D, raw1.data_ptr(), al1.data_ptr(), z1.data_ptr(),
                                   D, st))
        assert torch.equal(al1, al[g]) and torch.equal(z1, z[g])
        dp1, dhp1, dwp1 = torch.empty(B, L, A, device=dev), torch.empty(B, A, device=dev), torch.empty(B, A, device=dev)
        n.check(n.lib.rfn_attn_bwd(proj[g].data_ptr(), L * A, A, hp[g].data_ptr(), w[g].data_ptr(), al1.data_ptr(),
                                   x[g].data_ptr(), L * D, D, dz[g].data_ptr(), D, B, L, A, D, dp1.data_ptr(), L * A, A,
                                   0, dhp1.data_ptr(), dwp1.data_ptr(), st))
        assert torch.equal(dp1, dp[g]) and torch.equal(dhp1, dhp[g]) and torch.equal(dwp1, dwp[g])
        al_ref, z_ref = attn_ref(proj[g].cpu(), hp[g].cpu(), w[g].cpu(), bo[g].cpu(), x[g].cpu())
        assert maxerr(al[g], al_ref) < 2e-6 and maxerr(z[g], z_ref) < 1e-5


@pytest.mark.parametrize('Ls,Ds,A', [((50, 17, 64, 9), (96, 160, 36, 200), 64), ((196, 64, 49), (2048, 1536, 2208), 512),
                                     ((5, 7), (18, 30), 30)])
def test_stage1_attention_over_heterogeneous_encoders_matches_single_calls(dev, Ls, Ds, A):
    """rfn_attn_fwd_het / rfn_attn_bwd_het: encoders with different (L, D) maps in one launch (the reference's shipped
    feat_array.py:240-244 mix) against one call per encoder: same bits, and fp64 parity of alpha / z."""
    import ctypes as C
    n = N()
    G, B = len(Ls), 5
    st = n.stream_ptr()
    proj = [rnd(B, Ls[g], A, seed=10 + g).to(dev) for g in range(G)]
    hp = [rnd(B, A, seed=20 + g).to(dev) for g in range(G)]
    w = [rnd(A, seed=30 + g, scale=0.3).to(dev) for g in range(G)]
    bo = [rnd(1, seed=40 + g).to(dev) for g in range(G)]
    x = [rnd(B, Ls[g], Ds[g], seed=50 + g).to(dev) for g in range(G)]
    dz = [rnd(B, Ds[g], seed=60 + g).to(dev) for g in range(G)]
    raw = [torch.empty(B, Ls[g], device=dev) for g in range(G)]
    al = [torch.empty(B, Ls[g], device=dev) for g in range(G)]
    z = [torch.empty(B, Ds[g], device=dev) for g in range(G)]
    La, Da = (C.c_int * G)(*Ls), (C.c_int * G)(*Ds)
    n.check(n.lib.rfn_attn_fwd_het(G, n.ptr_array(proj), n.ptr_array(hp), n.ptr_array(w), n.ptr_array(bo), n.ptr_array(x), B,
                                   La, A, Da, n.ptr_array(raw), n.ptr_array(al), n.ptr_array(z), st), 'rfn_attn_fwd_het')
    dp = [torch.empty(B, Ls[g], A, device=dev) for g in range(G)]
    dhp = [torch.empty(B, A, device=dev) for g in range(G)]
    dwp = [torch.empty(B, A, device=dev) for g in range(G)]
    n.check(n.lib.rfn_attn_bwd_het(G, n.ptr_array(proj), n.ptr_array(hp), n.ptr_array(w), n.ptr_array(al), n.ptr_array(x),
                                   n.ptr_array(dz), B, La, A, Da, n.ptr_array(dp), 0, n.ptr_array(dhp), n.ptr_array(dwp), st),
            'rfn_attn_bwd_het')
    for g in range(G):
        L, D = Ls[g], Ds[g]
        raw1, al1, z1 = torch.empty(B, L, device=dev), torch.empty(B, L, device=dev), torch.empty(B, D, device=dev)
        n.check(n.lib.rfn_attn_fwd(proj[g].data_ptr(), L * A, A, hp[g].data_ptr(), w[g].data_ptr(), bo[g].data_ptr(),
                                   x[g].data_ptr(), L * D, D, B, L, A, D, raw1.data_ptr(), al1.data_ptr(), z1.data_ptr(),
                                   D, st))
        assert torch.equal(al1, al[g]) and torch.equal(z1, z[g])
        dp1, dhp1, dwp1 = torch.empty(B, L, A, device=dev), torch.empty(B, A, device=dev), torch.empty(B, A, device=dev)
        n.check(n.lib.rfn_attn_bwd(proj[g].data_ptr(), L * A, A, hp[g].data_ptr(), w[g].data_ptr(), al1.data_ptr(),
                                   x[g].data_ptr(), L * D, D, dz[g].data_ptr(), D, B, L, A, D, dp1.data_ptr(), L * A, A,
                                   0, dhp1.data_ptr(), dwp1.data_ptr(), st))
        assert torch.equal(dp1, dp[g]) and torch.equal(dhp1, dhp[g]) and torch.equal(dwp1, dwp[g])
        al_ref, z_ref = attn_ref(proj[g].cpu(), hp[g].cpu(), w[g].cpu(), bo[g].cpu(), x[g].cpu())
        assert maxerr(al[g], al_ref) < 2e-6 and maxerr(z[g], z_ref) < 2e-5
    # in place (dproj = proj) and accumulate, as the path uses it
    inpl = [t.clone() for t in proj]
    n.check(n.lib.rfn_attn_bwd_het(G, n.ptr_array(inpl), n.ptr_array(hp), n.ptr_array(w), n.ptr_array(al), n.ptr_array(x),
                                   n.ptr_array(dz), B, La, A, Da, n.ptr_array(inpl), 0, n.ptr_array(dhp), n.ptr_array(dwp), st),
            'rfn_attn_bwd_het (in place)')
    for g in range(G):
        assert torch.equal(inpl[g], dp[g])


def test_attention_time_major_strides(dev):
    """Stage II / decoder read thoughts stored (step, batch, feature): stride_b = R, stride_l = B*R."""
    n = N()
    B, L, A, D = 6, 8, 32, 32
    proj_tm, x_tm = rnd(L, B, A, seed=1), rnd(L, B, D, seed=2)
    hp, w, bo = rnd(B, A, seed=3), rnd(A, seed=4, scale=0.3), rnd(1, seed=5)
    al_ref, z_ref = attn_ref(proj_tm.transpose(0, 1), hp, w, bo, x_tm.transpose(0, 1))
    alpha = torch.empty(B, L, device=dev)
    z = torch.empty(B, D, device=dev)
    pd, xd, hpd, wd, bod = [t.to(dev) for t in (proj_tm, x_tm, hp, w, bo)]  # keep the device tensors alive
    n.check(n.lib.rfn_attn_scores_fwd(pd.data_ptr(), A, B * A, hpd.data_ptr(), wd.data_ptr(), bod.data_ptr(), B, L, A,
                                      alpha.data_ptr(), n.stream_ptr()))
    n.check(n.lib.rfn_attn_context_fwd(xd.data_ptr(), D, B * D, alpha.data_ptr(), B, L, D, z.data_ptr(), D,
                                       n.stream_ptr()))
    assert maxerr(alpha, al_ref) < 2e-6 and maxerr(z, z_ref) < 1e-5


@pytest.mark.parametrize('G,B,L,A,D', [(4, 6, 8, 512, 512), (1, 5, 8, 64, 96), (3, 4, 5, 30, 18), (2, 3, 32, 17, 7), (2, 3, 300, 24, 20)])
def test_fused_small_attention_groups_match_fp64(dev, G, B, L, A, D):
    """rfn_attn_small_fwd/bwd: G encoders in one launch over time-major (step, batch, G*feature) thoughts, as
    stage II lays them out; backward overwrites the projections in place and accumulates d thoughts."""
    n = N()
    proj = [rnd(B, L, A, seed=10 + g) for g in range(G)]           # (b, l) addressed with strides like P2
    hp = [rnd(B, A, seed=20 + g) for g in range(G)]
    w = [rnd(A, seed=30 + g, scale=0.3) for g in range(G)]
    bo = [rnd(1, seed=40 + g) for g in range(G)]
    x_tm = rnd(L, B, G * D, seed=50)                                # thoughts: Hs[1 + l][b, g*D:(g+1)*D]
    dz = rnd(B, G * D, seed=51)
    pd = [t.to(dev) for t in proj]
    hpd = [t.to(dev) for t in hp]
    wd = [t.to(dev) for t in w]
    bod = [t.to(dev) for t in bo]
    xd, dzd = x_tm.to(dev), dz.to(dev)
    alpha = torch.empty(G, B, L, device=dev)
    z = torch.empty(B, G * D, device=dev)
    st = n.stream_ptr()
    arr = lambda ts: n.ptr_array(ts)
    xs = [xd[:, :, g * D:] for g in range(G)]
    a_p, a_hp, a_w, a_b, a_x = arr(pd), arr(hpd), arr(wd), arr(bod), arr(xs)
    a_al, a_z = arr([alpha[g] for g in range(G)]), arr([z[:, g * D:] for g in range(G)])
    n.check(n.lib.rfn_attn_small_fwd(G, a_p, L * A, A, a_hp, a_w, a_b, a_x, G * D, B * G * D, B, L, A, D, a_al, a_z,
                                     G * D, st))
    refs = []
    for g in range(G):
        pr, hr, wr = [t.double().requires_grad_(True) for t in (proj[g], hp[g], w[g])]
        xr = x_tm[:, :, g * D:(g + 1) * D].transpose(0, 1).double().requires_grad_(True)
        al_ref, z_ref = attn_ref(pr, hr, wr, bo[g], xr)
        assert maxerr(alpha[g], al_ref) < 2e-6
        assert maxerr(z[:, g * D:(g + 1) * D], z_ref) < 1e-5
        z_ref.backward(dz[:, g * D:(g + 1) * D].double())
        refs.append((pr.grad, hr.grad, wr.grad, xr.grad))
    dx = torch.ones(L, B, G * D, device=dev)                        # accumulated into
    dhp = torch.empty(G, B, A, device=dev)
    dwp = torch.empty(G, B, A, device=dev)
    inpl = [t.clone() for t in pd]
    a_pi = arr(inpl)
    a_dz = arr([dzd[:, g * D:] for g in range(G)])
    a_dhp, a_dwp = arr([dhp[g] for g in range(G)]), arr([dwp[g] for g in range(G)])
    a_dx = arr([dx[:, :, g * D:] for g in range(G)])
    n.check(n.lib.rfn_attn_small_bwd(G, a_pi, L * A, A, a_hp, a_w, a_al, a_x, G * D, B * G * D, a_dz, G * D, B, L, A, D,
                                     a_pi, L * A, A, 0, a_dhp, a_dwp, a_dx, st))
    for g in range(G):
        gp, gh, gw, gx = refs[g]
        assert maxerr(inpl[g], gp) < 2e-5
        assert maxerr(dhp[g], gh) < 1e-4
        assert maxerr(dwp[g].sum(0), gw) < 1e-4 * max(1.0, float(gw.abs().max()))
        assert maxerr(dx[:, :, g * D:(g + 1) * D].transpose(0, 1), gx + 1.0) < 1e-5
    # separate, accumulated dproj (decoder form) and no d att_seq
    acc = [torch.ones(B, L, A, device=dev) for _ in range(G)]
    a_acc = arr(acc)
    n.check(n.lib.rfn_attn_small_bwd(G, a_p, L * A, A, a_hp, a_w, a_al, a_x, G * D, B * G * D, a_dz, G * D, B, L, A, D,
                                     a_acc, L * A, A, 1, a_dhp, a_dwp, None, st))
    for g in range(G):
        assert maxerr(acc[g], refs[g][0] + 1.0) < 2e-5
    # rejects L beyond the fused kernel's bound
    assert n.lib.rfn_attn_small_fwd(G, a_p, L * A, A, a_hp, a_w, a_b, a_x, G * D, B * G * D, B, 1025, A, D, a_al, a_z,
                                    G * D, st) != 0


def test_state_mean_over_encoders_and_backward(dev):
    """(h, c) mean over the M encoder slices in one launch (sum in slice order, then true division) and the
    broadcast of its gradient back to every slice."""
    n = N()
    B, R, M = 7, 24, 3
    H, Cc = rnd(B, M * R, seed=1), rnd(B, M * R, seed=2)
    Hd, Cd = H.to(dev), Cc.to(dev)
    h, c = torch.empty(B, R, device=dev), torch.empty(B, R, device=dev)
    n.check(n.lib.rfn_mean_over_groups(2, n.ptr_array([Hd, Cd]), M * R, R, M, n.ptr_array([h, c]), R, B, R,
                                       n.stream_ptr()))
    for out, src in ((h, H), (c, Cc)):
        ref = src[:, :R].clone()
        for i in range(1, M):
            ref = ref + src[:, i * R:(i + 1) * R]
        assert torch.equal(out.cpu(), ref / float(M))
    dh, dc = rnd(B, R, seed=3), rnd(B, R, seed=4)
    dhd, dcd = dh.to(dev), dc.to(dev)
    dH, dC = torch.ones(B, M * R, device=dev), torch.full((B, M * R), float('nan'), device=dev)
    beta = (C.c_float * 2)(1.0, 0.0)
    inv = 1.0 / M
    n.check(n.lib.rfn_bcast_to_groups(2, inv, n.ptr_array([dhd, dcd]), R, beta, n.ptr_array([dH, dC]), M * R, R, M, B,
                                      R, n.stream_ptr()))
    inv32 = torch.tensor(inv, dtype=torch.float32)
    assert torch.equal(dH.cpu(), (inv32 * dh).repeat(1, M) + 1.0)
    assert torch.equal(dC.cpu(), (inv32 * dc).repeat(1, M))


# ------------------------------------------------------------------------------------------------
# LSTM epilogue
# ------------------------------------------------------------------------------------------------
def test_lstm_forward_backward(dev):
    n = N()
    B, R = 7, 48
    sums, c0 = rnd(B, 4 * R, seed=1), rnd(B, R, seed=2)
    dh, dcn = rnd(B, R, seed=3), rnd(B, R, seed=4)
    sr, cr = sums.double().requires_grad_(True), c0.double().requires_grad_(True)
    sig = torch.sigmoid(sr[:, :3 * R])
    g = torch.tanh(sr[:, 3 * R:])
    c1 = sig[:, R:2 * R] * cr + sig[:, :R] * g
    h1 = sig[:, 2 * R:] * torch.tanh(c1)
    (h1 * dh.double()).sum().add((c1 * dcn.double()).sum()).backward()
    gd, c0d = sums.to(dev), c0.to(dev)
    c1d, h1d = torch.empty(B, R, device=dev), torch.empty(B, R, device=dev)
    st = n.stream_ptr()
    n.check(n.lib.rfn_lstm_fwd(gd.data_ptr(), 4 * R, c0d.data_ptr(), R, c1d.data_ptr(), R, h1d.data_ptr(), R, B, R,
                               0, 0.0, 0, 0, st))
    assert maxerr(h1d, h1) < 1e-6 and maxerr(c1d, c1) < 1e-6
    dcp = torch.empty(B, R, device=dev)
    dhd, dcnd = dh.to(dev), dcn.to(dev)
    n.check(n.lib.rfn_lstm_bwd(gd.data_ptr(), 4 * R, c0d.data_ptr(), R, c1d.data_ptr(), R, dhd.data_ptr(), R,
                               dcnd.data_ptr(), R, dcp.data_ptr(), R, B, R, 0, 0.0, 0, 0, st))
    assert maxerr(gd, sr.grad) < 2e-6 and maxerr(dcp, cr.grad) < 2e-6


def test_lstm_maxout_forward_backward(dev):
    """maxout cells (LSTMSoftMultiAttentionFeatArrayNoInputCore.py:60-62, LSTMSoftAttentionCore.py:89-91): 5R gates,
    in_transform = max(chunk 3, chunk 4) without tanh."""
    n = N()
    B, R = 5, 40
    sums, c0 = rnd(B, 5 * R, seed=1), rnd(B, R, seed=2)
    sums[:, 4 * R:4 * R + 7] = sums[:, 3 * R:3 * R + 7]        # ties: torch.max(a, b) splits their gradient evenly
    dh, dcn = rnd(B, R, seed=3), rnd(B, R, seed=4)
    sr, cr = sums.double().requires_grad_(True), c0.double().requires_grad_(True)
    sig = torch.sigmoid(sr[:, :3 * R])
    g = torch.max(sr[:, 3 * R:4 * R], sr[:, 4 * R:])
    c1 = sig[:, R:2 * R] * cr + sig[:, :R] * g
    h1 = sig[:, 2 * R:] * torch.tanh(c1)
    (h1 * dh.double()).sum().add((c1 * dcn.double()).sum()).backward()
    gd, c0d = sums.to(dev), c0.to(dev)
    c1d, h1d = torch.empty(B, R, device=dev), torch.empty(B, R, device=dev)
    st = n.stream_ptr()
    n.check(n.lib.rfn_lstm_fwd(gd.data_ptr(), 5 * R, c0d.data_ptr(), R, c1d.data_ptr(), R, h1d.data_ptr(), R, B, R,
                               1, 0.0, 0, 0, st))
    assert maxerr(h1d, h1) < 1e-6 and maxerr(c1d, c1) < 1e-6
    dcp = torch.empty(B, R, device=dev)
    dhd, dcnd = dh.to(dev), dcn.to(dev)
    n.check(n.lib.rfn_lstm_bwd(gd.data_ptr(), 5 * R, c0d.data_ptr(), R, c1d.data_ptr(), R, dhd.data_ptr(), R,
                               dcnd.data_ptr(), R, dcp.data_ptr(), R, B, R, 1, 0.0, 0, 0, st))
    assert maxerr(gd, sr.grad) < 2e-6 and maxerr(dcp, cr.grad) < 2e-6


def test_lstm_dropout_mask_is_regenerated(dev):
    n = N()
    B, R, p = 64, 256, 0.3
    sums, c0 = rnd(B, 4 * R, seed=1), rnd(B, R, seed=2)
    outs = []
    c0d = c0.to(dev)
    for seed in (11, 11, 12):
        gd = sums.to(dev).clone()
        c1d, h1d = torch.empty(B, R, device=dev), torch.empty(B, R, device=dev)
        n.check(n.lib.rfn_lstm_fwd(gd.data_ptr(), 4 * R, c0d.data_ptr(), R, c1d.data_ptr(), R, h1d.data_ptr(),
                                   R, B, R, 0, p, seed, 5, n.stream_ptr()))
        outs.append((gd, c1d, h1d))
    assert torch.equal(outs[0][2], outs[1][2]) and not torch.equal(outs[0][2], outs[2][2])
    keep = (outs[0][2] != 0).float().mean().item()
    assert abs(keep - (1 - p)) < 0.02
    # backward applies the same mask: dgates of dropped units get no o-gate gradient
    gd, c1d, h1d = outs[0]
    dh = torch.ones(B, R, device=dev)
    dcp = torch.empty(B, R, device=dev)
    act = gd.clone()
    n.check(n.lib.rfn_lstm_bwd(gd.data_ptr(), 4 * R, c0d.data_ptr(), R, c1d.data_ptr(), R, dh.data_ptr(), R,
                               None, R, dcp.data_ptr(), R, B, R, 0, p, 11, 5, n.stream_ptr()))
    dropped = (h1d == 0) & (act[:, 2 * R:3 * R] * torch.tanh(c1d) != 0)
    assert float(gd[:, 2 * R:3 * R][dropped].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------
# small kernels
# ------------------------------------------------------------------------------------------------
def test_colsum_embed_logsoftmax_max(dev):
    n = N()
    st = n.stream_ptr()
    X = rnd(301, 77, seed=1)
    out = torch.full((77,), 2.0, device=dev)
    Xd = X.to(dev)
    n.check(n.lib.rfn_colsum_f32(Xd.data_ptr(), 77, 301, 77, out.data_ptr(), 1, st))
    assert maxerr(out, X.double().sum(0) + 2.0) < 1e-4
    # embedding gather + deterministic scatter, ids laid out (B, ld) read as (step, batch) rows
    V1, E, B, S = 51, 20, 5, 4
    W = rnd(V1, E, seed=2)
    ids = torch.randint(0, V1, (B, S + 2), generator=torch.Generator().manual_seed(3))
    xs = torch.empty(S * B, E, device=dev)
    idd = ids.to(dev)
    Wd = W.to(dev)
    n.check(n.lib.rfn_embed_fwd(Wd.data_ptr(), E, V1, idd.data_ptr(), B, S + 2, 1, S * B, xs.data_ptr(), E, st))
    ref = W[ids[:, :S].t().reshape(-1)]
    assert maxerr(xs, ref) == 0.0
    dxs = rnd(S * B, E, seed=4)
    dW = torch.full((V1, E), float('nan'), device=dev)
    dxsd = dxs.to(dev)
    n.check(n.lib.rfn_embed_bwd(dxsd.data_ptr(), E, idd.data_ptr(), B, S + 2, 1, S * B, E, V1, dW.data_ptr(), st))
    refW = torch.zeros(V1, E, dtype=torch.float64).index_add_(0, ids[:, :S].t().reshape(-1), dxs.double())
    assert maxerr(dW, refW) < 1e-5
    # log-softmax with the (step,batch) -> (batch,step) row mapping, and its backward
    V1 = 9488
    lg = rnd(S * B, V1, seed=5, scale=3.0)
    lp = torch.empty(B, S, V1, device=dev)
    lgd = lg.to(dev)
    n.check(n.lib.rfn_log_softmax_fwd(lgd.data_ptr(), V1, S * B, V1, B, S * V1, V1, lp.data_ptr(), st))
    ref = torch.log_softmax(lg.double(), 1).view(S, B, V1).transpose(0, 1)
    assert maxerr(lp, ref) < 5e-6
    g = rnd(B, S, V1, seed=6)
    dl = torch.empty(S * B, V1, device=dev)
    gdv = g.to(dev)
    n.check(n.lib.rfn_log_softmax_bwd(gdv.data_ptr(), lp.data_ptr(), S * B, V1, B, S * V1, V1, dl.data_ptr(), V1, st))
    gr = g.double().transpose(0, 1).reshape(S * B, V1)
    refd = gr - torch.softmax(lg.double(), 1) * gr.sum(1, keepdim=True)
    assert maxerr(dl, refd) < 2e-4
    # max over steps
    Xs = rnd(6, 9, 33, seed=7)
    mo = torch.empty(9, 33, device=dev)
    arg = torch.empty(9, 33, dtype=torch.int32, device=dev)
    Xsd = Xs.to(dev)
    n.check(n.lib.rfn_max_over_steps_fwd(Xsd.data_ptr(), 6, 9, 33, mo.data_ptr(), arg.data_ptr(), st))
    mref, aref = Xs.max(0)
    assert maxerr(mo, mref) == 0.0 and torch.equal(arg.cpu().long(), aref)
    dmo = rnd(9, 33, seed=8)
    dX = torch.empty(6, 9, 33, device=dev)
    dmod = dmo.to(dev)
    n.check(n.lib.rfn_max_over_steps_bwd(dmod.data_ptr(), arg.data_ptr(), 6, 9, 33, dX.data_ptr(), st))
    refdX = torch.zeros(6, 9, 33).scatter_(0, aref.unsqueeze(0), dmo.unsqueeze(0))
    assert maxerr(dX, refdX) == 0.0
    # axpby / div
    y = rnd(5, 12, seed=9).to(dev)
    y0 = y.clone()
    xx = rnd(5, 4, seed=10).to(dev)
    n.check(n.lib.rfn_axpby_2d(2.0, xx.data_ptr(), 4, 1.0, y[:, 4:].data_ptr(), 12, 5, 4, st))
    assert maxerr(y[:, 4:8], y0[:, 4:8] + 2 * xx) < 1e-6 and torch.equal(y[:, :4], y0[:, :4])
    n.check(n.lib.rfn_div_2d(y.data_ptr(), 12, 5, 12, 3.0, st))
    y0[:, 4:8] += 2 * xx
    # IEEE division, compared with the CPU result (torch's GPU scalar division multiplies by 1/3)
    assert torch.equal(y.cpu(), y0.cpu() / 3.0)


@pytest.mark.parametrize('eps', [0.0, 0.1])
def test_criteria_match_torch(dev, eps):
    import torch.nn.functional as F
    n = N()
    B, T, V1, K = 6, 5, 301, 50
    lp = torch.log_softmax(rnd(B, T, V1, seed=1), 2)
    g = torch.Generator().manual_seed(2)
    labels = torch.randint(0, V1, (B, T + 2), generator=g)
    mask = (torch.rand(B, T + 2, generator=g) > 0.3).float()
    target, mk = labels[:, 1:], mask[:, 1:]          # strided views, as train.py passes them
    pred = rnd(B, K, seed=3)
    top = -torch.ones(B, K, dtype=torch.long)
    for b in range(B):
        k = b % 4                                     # includes a row with no targets
        top[b, :k] = torch.randperm(K, generator=g)[:k]
    top[1, 1] = top[1, 0]                             # duplicated target id
    lpr = lp.double().requires_grad_(True)
    pr = pred.double().requires_grad_(True)
    oh = torch.zeros(B, T, V1, dtype=torch.float64).scatter_(2, target[:, :T].unsqueeze(2), 1.0)
    q = oh * (1 - eps) + eps / V1 if eps > 0 else oh
    ref = (-(lpr * q).sum(2) * mk[:, :T].double()).sum() / B + 0.7 * F.multilabel_margin_loss(pr, top)
    ref.backward()
    from recurrent_fusion_network_amd.criteria import _XEFn
    lpd = lp.to(dev).requires_grad_(True)
    pd = pred.to(dev).requires_grad_(True)
    loss = _XEFn.apply(eps, 0.7, target.to(dev), mk.to(dev), top.to(dev), lpd, pd)
    loss.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-5 * max(1.0, abs(float(ref.detach())))
    assert maxerr(lpd.grad, lpr.grad) < 1e-7
    assert maxerr(pd.grad, pr.grad) < 1e-7


def test_adam_multi_bucket_launch_is_bit_identical_to_per_bucket_calls(dev):
    """rfn_adam_step_multi: every flat bucket of the optimizer in one launch, same bits as one rfn_adam_step per bucket
    (vector-aligned buckets, a ragged one, an unaligned view)."""
    import ctypes as C
    n = N()
    st = n.stream_ptr()
    sizes = [4096 * 300, 10007, 64, 4 * 12345 + 4]
    base = [torch.randn(s_ + 4, device=dev, generator=torch.Generator(device=dev).manual_seed(k)) * 0.1 for k, s_ in enumerate(sizes)]
    views = [b[:s_] for b, s_ in zip(base, sizes)]
    views[3] = base[3][1:1 + sizes[3]]             # 4-byte offset: takes the scalar path
    gs = [torch.randn(s_, device=dev, generator=torch.Generator(device=dev).manual_seed(10 + k)) * 2.0 for k, s_ in enumerate(sizes)]

    def fresh():
        return ([v.clone() for v in views], [torch.zeros(s_, device=dev) for s_ in sizes], [torch.zeros(s_, device=dev) for s_ in sizes])
    p1, m1, v1 = fresh()
    p2, m2, v2 = fresh()
    for step in (1, 2, 3):
        for k in range(len(sizes)):
            n.check(n.lib.rfn_adam_step(p1[k].data_ptr(), gs[k].data_ptr(), m1[k].data_ptr(), v1[k].data_ptr(), sizes[k], 5e-4,
                                        0.9, 0.999, 1e-8, 1e-5, 1.0, 0.5, step, st))
        n.check(n.lib.rfn_adam_step_multi(len(sizes), n.ptr_array(p2), n.ptr_array(gs), n.ptr_array(m2), n.ptr_array(v2),
                                          (C.c_int64 * len(sizes))(*sizes), 5e-4, 0.9, 0.999, 1e-8, 1e-5, 1.0, 0.5, step, st))
    for k in range(len(sizes)):
        assert torch.equal(p1[k], p2[k]) and torch.equal(m1[k], m2[k]) and torch.equal(v1[k], v2[k])
    assert n.lib.rfn_adam_step_multi(17, n.ptr_array(p2 * 5), n.ptr_array(gs * 5), n.ptr_array(m2 * 5), n.ptr_array(v2 * 5),
                                     (C.c_int64 * 20)(*(sizes * 5)), 5e-4, 0.9, 0.999, 1e-8, 0.0, 1.0, 1.0, 1, st) != 0


def test_adam_and_greedy_pick(dev):
    n = N()
    st = n.stream_ptr()
    nel = 10007
    p, g = rnd(nel, seed=1), rnd(nel, seed=2, scale=2.0)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5)
    pd, m, v = p.to(dev), torch.zeros(nel, device=dev), torch.zeros(nel, device=dev)
    for step in (1, 2, 3):
        pr.grad = (g * 0.5 * step).clamp(-1, 1)
        opt.step()
        gstep = (g * step).to(dev)
        n.check(n.lib.rfn_adam_step(pd.data_ptr(), gstep.data_ptr(), m.data_ptr(), v.data_ptr(), nel, 5e-4,
                                    0.9, 0.999, 1e-8, 1e-5, 1.0, 0.5, step, st))
    assert maxerr(pd, pr) < 2e-6
    # greedy pick: first maximum, unfinished bookkeeping, unmasked next ids
    B, V1 = 5, 9488
    lp = torch.log_softmax(rnd(B, V1, seed=3), 1)
    lp[2, 0] = 5.0          # row 2 picks END
    lp[3, 17] = lp[3, 400] = 6.0   # tie -> first index
    lpd = lp.to(dev)
    nxt = torch.empty(B, dtype=torch.long, device=dev)
    seq = torch.zeros(B, 4, dtype=torch.long, device=dev)
    slp = torch.zeros(B, 4, device=dev)
    unf = torch.zeros(3, B, dtype=torch.int32, device=dev)
    n.check(n.lib.rfn_greedy_pick(lpd.data_ptr(), V1, B, V1, 1, nxt.data_ptr(), seq[:, 0].data_ptr(), 4,
                                  slp[:, 0].data_ptr(), 4, None, unf[1].data_ptr(), st))
    val, idx = lp.max(1)
    assert torch.equal(nxt.cpu(), idx) and int(nxt[3]) == 17 and int(nxt[2]) == 0
    assert torch.equal(unf[1].cpu(), (idx > 0).int())
    assert torch.equal(seq[:, 0].cpu(), idx * (idx > 0).long()) and maxerr(slp[:, 0], val) == 0.0
    lp2 = lp.clone()
    lp2[2, 0] = -50.0       # row 2 would continue, but it is already finished
    lp2d = lp2.to(dev)
    n.check(n.lib.rfn_greedy_pick(lp2d.data_ptr(), V1, B, V1, 2, nxt.data_ptr(), seq[:, 1].data_ptr(), 4,
                                  slp[:, 1].data_ptr(), 4, unf[1].data_ptr(), unf[2].data_ptr(), st))
    assert int(unf[2, 2]) == 0 and int(seq[2, 1]) == 0 and int(nxt[2]) == int(lp2[2].argmax())


def _beam_reference(logps, W, S):
    """Pure-Python restatement of misc/RecurrentFusionModel.py:451-531 driven by a fixed log-prob table per step
    (rows are re-gathered exactly as the recurrent state would be)."""
    V1 = logps[0].shape[1]
    beam_seq = np.zeros((S, W), dtype=np.int64)
    beam_lp = np.zeros((S, W), dtype=np.float32)
    beam_sum = np.zeros(W, dtype=np.float32)
    done, src = [], np.arange(W)
    for t in range(1, S + 1):
        lp = logps[t - 1][src]                       # state of row r descends from row src[r]
        ixs = np.argsort(-lp, axis=1, kind='stable')
        cols, rows = min(W, V1), (1 if t == 1 else W)
        cands = []
        for c in range(cols):
            for q in range(rows):
                if t > 1 and beam_seq[t - 2, q] == 0:
                    continue
                local = lp[q, ixs[q, c]]
                cands.append(dict(c=int(ixs[q, c]), q=q, p=np.float32(beam_sum[q] + local), r=local))
        if not cands:
            break
        cands = sorted(cands, key=lambda x: -x['p'])
        prev_seq, prev_lp = beam_seq.copy(), beam_lp.copy()
        new_src = np.arange(W)
        for vix in range(min(W, len(cands))):
            v = cands[vix]
            beam_seq[:t - 1, vix] = prev_seq[:t - 1, v['q']]
            beam_lp[:t - 1, vix] = prev_lp[:t - 1, v['q']]
            new_src[vix] = v['q']
            beam_seq[t - 1, vix], beam_lp[t - 1, vix], beam_sum[vix] = v['c'], v['r'], v['p']
            if v['c'] == 0 or t == S:
                done.append((beam_seq[:, vix].copy(), beam_lp[:, vix].copy(), float(beam_sum[vix])))
        src = new_src                                  # rows >= len(cands) keep their own state
    return done


@pytest.mark.parametrize('W,V1,S,ties', [(3, 7, 5, False), (5, 40, 6, False), (4, 3, 4, False), (8, 500, 6, False),
                                         (16, 2000, 4, False), (1, 50, 4, False), (2, 100, 5, True),
                                         (5, 9488, 6, True), (9, 700, 5, True)])
def test_beam_step_kernel_matches_python_bookkeeping(dev, W, V1, S, ties):
    """rfn_beam_step against the reference's bookkeeping with END tokens frequent enough to finish beams early
    (and, for the tiny vocabulary, to exhaust every candidate -> the early break); every register-list width of
    the top-k phase, tied log-probs, the full vocabulary."""
    n = N()
    NB = 4
    g = torch.Generator().manual_seed(W * 100 + V1)
    tables = [torch.log_softmax(torch.randn(NB * W, V1, generator=g) * 2.0, 1) for _ in range(S)]
    if ties:                                             # many equal log-probs: ties go to the lowest column id
        tables = [torch.round(tb * 2.0) / 2.0 for tb in tables]
    for tb in tables:
        tb[:, 0] += 1.5                                  # make END (id 0) likely
    st = n.stream_ptr()
    bs = torch.zeros(S, NB, W, dtype=torch.long, device=dev)
    bl = torch.zeros(S, NB, W, device=dev)
    bsum = torch.zeros(NB, W, device=dev)
    order = torch.arange(NB * W, dtype=torch.int32, device=dev)
    ids = torch.zeros(NB * W, dtype=torch.long, device=dev)
    MAXD = W * S
    dseq = torch.zeros(NB, MAXD, S, dtype=torch.long, device=dev)
    dlp = torch.zeros(NB, MAXD, S, device=dev)
    dp = torch.zeros(NB, MAXD, device=dev)
    dn = torch.zeros(NB, dtype=torch.int32, device=dev)
    act = torch.ones(NB, dtype=torch.int32, device=dev)
    src = torch.arange(NB * W, device=dev)                 # which original row each state row descends from
    for t in range(1, S + 1):
        cur = tables[t - 1].to(dev)[src].contiguous()
        n.check(n.lib.rfn_beam_step(cur.data_ptr(), V1, V1, W, S, t, NB, MAXD, bs.data_ptr(), bl.data_ptr(),
                                    bsum.data_ptr(), order.data_ptr(), ids.data_ptr(), dseq.data_ptr(), dlp.data_ptr(),
                                    dp.data_ptr(), dn.data_ptr(), act.data_ptr(), st))
        src = order.long()                                 # next step's row r descends from row order[r]
    for k in range(NB):
        ref = _beam_reference([tb[k * W:(k + 1) * W].numpy() for tb in tables], W, S)
        assert int(dn[k]) == len(ref), (k, int(dn[k]), len(ref))
        for j, (rs, rl, rp) in enumerate(ref):
            assert np.array_equal(dseq[k, j].cpu().numpy(), rs)
            assert np.allclose(dlp[k, j].cpu().numpy(), rl, atol=1e-6)
            assert abs(float(dp[k, j]) - rp) < 1e-5


def test_log_softmax_topk_of_rows_with_nan_logits_stays_in_bounds(dev):
    """a row whose logits are NaN offers no comparable candidate to the final merge; the kernel must write
    (-inf, token 0) placeholders instead of indexing its candidate lists with -1.  Healthy rows beside it are untouched."""
    n = N()
    V1, W, rows = 1000, 5, 6
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(rows, V1, generator=g)
    logits[1] = float('nan')                      # a whole row of NaN
    logits[4, 3:] = float('nan')                  # fewer comparable entries than W (three finite ones... made NaN by the sum)
    lg = logits.to(dev)
    topv = torch.full((rows, W), 7.0, device=dev)
    topi = torch.full((rows, W), 123456, dtype=torch.int32, device=dev)
    n.check(n.lib.rfn_log_softmax_topk(lg.data_ptr(), V1, rows, V1, W, topv.data_ptr(), topi.data_ptr(), n.stream_ptr()))
    torch.cuda.synchronize()
    ti, tv = topi.cpu(), topv.cpu()
    assert int(ti.min()) >= 0 and int(ti.max()) < V1                      # every written token id is a valid row index
    for r in (1, 4):
        assert bool(((tv[r] == float('-inf')) | torch.isnan(tv[r])).all()), tv[r]
    lp = torch.log_softmax(logits.double(), 1)
    for r in (0, 2, 3, 5):
        order = torch.sort(lp[r], descending=True, stable=True).indices[:W]
        assert torch.equal(ti[r].long(), order)


@pytest.mark.parametrize('W,V1,rows,ties', [(5, 9488, 37, False), (3, 301, 12, True), (16, 9488, 6, False), (8, 50, 9, True),
                                             (5, 10, 4, False), (24, 9488, 30, False), (32, 301, 40, True)])
def test_log_softmax_topk_and_the_beam_step_it_feeds(dev, W, V1, rows, ties):
    """rfn_log_softmax_topk: the W best entries of every row of log_softmax(logits), ordered (value descending, token
    ascending), from the same log-prob bits rfn_log_softmax_fwd writes -- and rfn_beam_step_topk on those lists does exactly
    what rfn_beam_step does on the full rows (vectorised and scalar row forms, ties, W up to 16, W > V+1); the lists alone up
    to W = 32."""
    n = N()
    st = n.stream_ptr()
    g = torch.Generator().manual_seed(W * 1000 + V1)
    logits = torch.randn(rows, V1, generator=g) * 3.0
    if ties:
        logits = torch.round(logits)
    lg = logits.to(dev)
    lp = torch.empty(rows, V1, device=dev)
    n.check(n.lib.rfn_log_softmax_fwd(lg.data_ptr(), V1, rows, V1, rows, V1, 0, lp.data_ptr(), st))
    assert maxerr(lp, torch.log_softmax(logits.double(), 1)) < 2e-6
    topv = torch.full((rows, W), float('nan'), device=dev)
    topi = torch.full((rows, W), -1, dtype=torch.int32, device=dev)
    n.check(n.lib.rfn_log_softmax_topk(lg.data_ptr(), V1, rows, V1, W, topv.data_ptr(), topi.data_ptr(), st))
    cols = min(W, V1)
    # reference order: stable sort of the log-prob bits, descending (ties keep the lower token first)
    order = torch.sort(lp.cpu(), dim=1, descending=True, stable=True).indices[:, :cols]
    assert torch.equal(topi[:, :cols].cpu().long(), order)
    assert torch.equal(topv[:, :cols].cpu(), lp.cpu().gather(1, order))
    # one beam step from the lists == one beam step from the rows (NB images x W beams need rows == NB * W)
    NB = rows // W
    if NB < 1 or W > 16:      # the full-row form cuts a row over 16 / W waves: W <= 16 (wider beams: tests/test_model_gpu.py)
        return
    S, MAXD = 4, W * 4

    def fresh():
        return dict(bs=torch.zeros(S, NB, W, dtype=torch.long, device=dev), bl=torch.zeros(S, NB, W, device=dev),
                    bsum=torch.zeros(NB, W, device=dev), order=torch.zeros(NB * W, dtype=torch.int32, device=dev),
                    ids=torch.zeros(NB * W, dtype=torch.long, device=dev),
                    dseq=torch.zeros(NB, MAXD, S, dtype=torch.long, device=dev), dlp=torch.zeros(NB, MAXD, S, device=dev),
                    dp=torch.zeros(NB, MAXD, device=dev), dn=torch.zeros(NB, dtype=torch.int32, device=dev),
                    act=torch.ones(NB, dtype=torch.int32, device=dev))
    a, b = fresh(), fresh()
    for t in (1, 2, 3):
        n.check(n.lib.rfn_beam_step(lp.data_ptr(), V1, V1, W, S, t, NB, MAXD, *[a[k].data_ptr() for k in
                                    ('bs', 'bl', 'bsum', 'order', 'ids', 'dseq', 'dlp', 'dp', 'dn', 'act')], st))
        n.check(n.lib.rfn_beam_step_topk(topv.data_ptr(), topi.data_ptr(), V1, W, S, t, NB, MAXD, *[b[k].data_ptr() for k in
                                         ('bs', 'bl', 'bsum', 'order', 'ids', 'dseq', 'dlp', 'dp', 'dn', 'act')], st))
        for k in a:
            assert torch.equal(a[k], b[k]), (t, k)


def test_gemm_randomized_shapes_layouts_segments_groups(dev):
    """60 random problems: all operand layouts, ragged / tiny / large dims, 1-4 K segments of different lengths
    (different per group), 1-5 groups, bias on some segments, accumulate, padded ldc/lda/ldb, split-K scratch on
    and off -- every dispatch branch of rfn_gemm_f32_ws against fp64."""
    import random
    n = N()
    rng = random.Random(1234)
    ws = torch.empty(48 << 20, dtype=torch.uint8, device=dev)
    ws2 = torch.empty(48 << 20, dtype=torch.uint8, device=dev)
    tickets = torch.zeros(16384, dtype=torch.int32, device=dev)
    g = torch.Generator().manual_seed(99)
    dims_m = [1, 3, 64, 100, 128, 256, 300, 1024]
    dims_n = [1, 20, 51, 64, 128, 130, 512, 2048]
    dims_k = [1, 7, 16, 32, 40, 96, 256, 1000, 2048]
    for case in range(60):
        M, Nn = rng.choice(dims_m), rng.choice(dims_n)
        ak, bk = rng.randint(0, 1), rng.randint(0, 1)
        ngroups, nseg = rng.randint(1, 5), rng.randint(1, 4)
        acc, use_ws = rng.random() < 0.4, rng.random() < 0.6
        pad = rng.choice([0, 0, 4, 5])
        problems, refs, keep = [], [], []
        for _ in range(ngroups):
            ldc = Nn + pad
            Cfull = torch.randn(M, ldc, generator=g)
            ref = Cfull[:, :Nn].double().clone() if acc else torch.zeros(M, Nn, dtype=torch.float64)
            segs = []
            for _ in range(nseg):
                K = rng.choice(dims_k)
                A, Bm = torch.randn(M, K, generator=g), torch.randn(Nn, K, generator=g)
                ref += A.double() @ Bm.double().t()
                bias = None
                if rng.random() < 0.5:
                    bias = torch.randn(Nn, generator=g)
                    ref += bias.double()
                # storage with padded leading dimensions
                if ak:
                    Ast = torch.zeros(M, K + pad); Ast[:, :K] = A; lda = K + pad
                else:
                    Ast = torch.zeros(K, M + pad); Ast[:, :M] = A.t(); lda = M + pad
                if bk:
                    Bst = torch.zeros(Nn, K + pad); Bst[:, :K] = Bm; ldb = K + pad
                else:
                    Bst = torch.zeros(K, Nn + pad); Bst[:, :Nn] = Bm.t(); ldb = Nn + pad
                Ad, Bd = Ast.to(dev), Bst.to(dev)
                bd = None if bias is None else bias.to(dev)
                keep += [Ad, Bd, bd]
                segs.append((Ad, lda, ak, Bd, ldb, bk, K, bd))
            Cd = Cfull.to(dev)
            problems.append((Cd, ldc, segs))
            refs.append((ref, Cfull))
        # the same launch with split-K finished inside the kernel (rfn_gemm_f32_tk): bit-identical, counters left at zero
        if use_ws:
            twin = [(Cfull.to(dev), ldc, segs) for (Cd, ldc, segs), (ref, Cfull) in zip(problems, refs)]
            n.gemm(M, Nn, twin, accumulate=acc, ws=ws2, tickets=tickets)
            assert int(tickets.abs().sum()) == 0, 'a tile counter was left non-zero'
        n.gemm(M, Nn, problems, accumulate=acc, ws=ws if use_ws else None)
        if use_ws:
            for (Cd, _, _), (Ct, _, _) in zip(problems, twin):
                assert torch.equal(Cd, Ct), (case, 'in-kernel split-K finish differs from the reduce kernel')
        for (Cd, ldc, segs), (ref, Cfull) in zip(problems, refs):
            ktot = sum(s_[6] for s_ in segs)
            tol = 1e-5 + 3e-6 * ktot * 3.0
            err = maxerr(Cd[:, :Nn], ref)
            assert err < tol, (case, M, Nn, ak, bk, ngroups, nseg, acc, use_ws, pad, err, tol)
            if pad:
                assert torch.equal(Cd[:, Nn:].cpu(), Cfull[:, Nn:]), 'wrote outside the N columns'


@pytest.mark.parametrize('use_ppo', [0, 1])
def test_rl_reward_criterion_matches_oracle(dev, use_ppo):
    """ReviewNetRewardCriterion (misc/utils.py:50-84) incl. the PPO-clip surrogate, finished rows (seq == 0) and a
    logprobs_all that holds one more step than the sampled sequence."""
    from types import SimpleNamespace
    import recurrent_fusion_network_amd as R
    from oracle import rfn_oracle as O
    B, T, V1, K = 6, 5, 301, 40
    g = torch.Generator().manual_seed(5 + use_ppo)
    lp_all = torch.log_softmax(torch.randn(B, T + 1, V1, generator=g), 2)
    seq = torch.randint(1, V1, (B, T), generator=g)
    seq[1, 2:] = 0
    seq[4, 0:] = 0
    inp = lp_all[:, :T].gather(2, seq.unsqueeze(2)).squeeze(2).clone()
    old = inp + 0.3 * torch.randn(B, T, generator=g)
    reward = torch.randn(B, 1, generator=g).expand(B, T).contiguous() * 2.0
    preds = [torch.randn(B, K, generator=g) for _ in range(2)]
    top = -torch.ones(B, K, dtype=torch.long)
    top[:, :3] = torch.stack([torch.randperm(K, generator=g)[:3] for _ in range(B)])
    cfg = SimpleNamespace(use_ppo=use_ppo, ppo_clip=0.2, use_label_smoothing=0, label_smoothing_epsilon=0.1)
    # oracle (fp64 autograd)
    ir, lr = inp.double().requires_grad_(True), lp_all.double().requires_grad_(True)
    pr = [p.double().requires_grad_(True) for p in preds]
    ref = O.rl_criterion(cfg, ir, seq, reward.double(), lr, 0.01, pr, top, 1.0, old.double())
    ref.backward()
    crit = R.ReviewNetRewardCriterion(cfg)
    idv, ldv = inp.to(dev).requires_grad_(True), lp_all.to(dev).requires_grad_(True)
    pdv = [p.to(dev).requires_grad_(True) for p in preds]
    loss = crit(idv, seq.to(dev), reward.to(dev), ldv, 0.01, pdv, top.to(dev), 1.0, old.to(dev), cfg)
    loss.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-5 * max(1.0, abs(float(ref.detach())))
    assert maxerr(idv.grad, ir.grad) < 1e-6
    assert maxerr(ldv.grad, lr.grad) < 1e-7
    for a, b in zip(pdv, pr):
        assert maxerr(a.grad, b.grad) < 1e-7


def test_multinomial_pick_is_the_inverse_cdf_of_its_uniform(dev):
    """rfn_multinomial_pick (sample(sample_max=0), scheduled sampling; misc/RecurrentFusionModel.py:623-631, 260-270):
    the drawn index is the inverse CDF of exp(logp / T) at the caller's uniform -- checked against an fp64 cumulative
    sum --, tokens without mass are never drawn, the coin mask keeps the other rows' tokens, and the empirical
    frequencies of many draws follow the distribution."""
    n = N()
    g = torch.Generator().manual_seed(0)
    B, V1 = 64, 9488
    logits = torch.randn(B, V1 + 5, generator=g) * 3.0
    logits[:, 100:200] = float('-inf')                       # a block of impossible tokens
    logp = torch.log_softmax(logits[:, :V1].double(), 1).float()
    pad = torch.full((B, V1 + 5), float('nan'))
    pad[:, :V1] = logp                                       # row stride > V1, as a column of (B, S, V1)
    for temp in (1.0, 0.7):
        u = torch.rand(B, generator=g)
        u[0], u[1] = 0.0, 0.99999994                         # both ends of the interval
        ids = torch.full((B, 3), -7, dtype=torch.long, device=dev)
        n.check(n.lib.rfn_multinomial_pick(pad.to(dev).data_ptr(), V1 + 5, B, V1, 1.0 / temp, u.to(dev).data_ptr(), None,
                                           1.0, ids[:, 1].data_ptr(), 3, n.stream_ptr()))
        got = ids.cpu()
        assert bool((got[:, 0] == -7).all()) and bool((got[:, 2] == -7).all())      # only the addressed column
        p = torch.exp(logp.double() / temp)
        cdf = torch.cumsum(p, 1)
        tgt = u.double() * cdf[:, -1]
        for b in range(B):
            v = int(got[b, 1])
            assert 0 <= v < V1 and not (100 <= v < 200)
            lo = float(cdf[b, v - 1]) if v > 0 else 0.0
            tol = 2e-6 * float(cdf[b, -1])                   # fp32 partial sums vs the fp64 cumulative sum
            assert lo - tol <= float(tgt[b]) <= float(cdf[b, v]) + tol, (b, v, lo, float(tgt[b]), float(cdf[b, v]))
    # scheduled-sampling mask: rows with coin >= keep_prob keep their token
    keep = torch.tensor([0.1, 0.9] * (B // 2))
    ids = torch.full((B,), 5, dtype=torch.long, device=dev)
    u = torch.rand(B, generator=g)
    n.check(n.lib.rfn_multinomial_pick(pad.to(dev).data_ptr(), V1 + 5, B, V1, 1.0, u.to(dev).data_ptr(),
                                       keep.to(dev).data_ptr(), 0.5, ids.data_ptr(), 1, n.stream_ptr()))
    assert bool((ids.cpu()[1::2] == 5).all()) and not bool((ids.cpu()[0::2] == 5).all())
    # empirical frequencies on a small vocabulary: 200 000 draws of one distribution
    V, R = 13, 200000
    lp = torch.log_softmax(torch.randn(V, generator=g) * 1.5, 0)
    rows = lp.repeat(R, 1).contiguous().to(dev)
    out = torch.empty(R, dtype=torch.long, device=dev)
    n.check(n.lib.rfn_multinomial_pick(rows.data_ptr(), V, R, V, 1.0, torch.rand(R, generator=g).to(dev).data_ptr(), None,
                                       1.0, out.data_ptr(), 1, n.stream_ptr()))
    freq = torch.bincount(out.cpu(), minlength=V).double() / R
    assert float((freq - torch.exp(lp.double())).abs().max()) < 5e-3


def test_gemm_big_tile_fuzz_across_kernels(dev):
    """Random interior big-tile problems (rows / columns multiples of 128, K segments multiples of 32, 1-3 groups, 1-3
    segments, every operand layout, with and without accumulate, wide leading dimensions): the LDS-DMA kernel, the
    register-staged kernel and the lean tiles must agree bit for bit and match fp64 -- including launches with a
    half-height tail round and K ranges shorter than the DMA ring."""
    n = N()
    rng = np.random.default_rng(11)
    for case in range(14):
        ak, bk = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        G = int(rng.integers(1, 4))
        # at least 384 tiles of 128 x 128 so that the big-tile dispatch is taken
        tm, tn = int(rng.integers(1, 40)), int(rng.integers(1, 12))
        while tm * tn * G < 384:
            tm += int(rng.integers(1, 24))
        M, Nn = 128 * tm, 128 * tn
        Ks = [32 * int(rng.integers(1, 5)) for _ in range(int(rng.integers(1, 4)))]
        pad_a, pad_b, pad_c = 4 * int(rng.integers(0, 3)), 4 * int(rng.integers(0, 3)), 4 * int(rng.integers(0, 3))
        acc = bool(rng.integers(0, 2))
        probs, refs, keep = [], [], []
        for g in range(G):
            segs, ref = [], torch.zeros(M, Nn, dtype=torch.float64)
            for s_, K in enumerate(Ks):
                A, Bm = rnd(M, K, seed=1000 * case + 10 * g + s_), rnd(Nn, K, seed=5000 + 1000 * case + 10 * g + s_)
                b = rnd(Nn, seed=9000 + case + s_) if rng.integers(0, 2) else None
                ref += A.double() @ Bm.double().t() + (b.double() if b is not None else 0.0)
                if ak:
                    A_st = torch.zeros(M, K + pad_a)
                    A_st[:, :K] = A
                    lda = K + pad_a
                else:
                    A_st = torch.zeros(K, M + pad_a)
                    A_st[:, :M] = A.t()
                    lda = M + pad_a
                if bk:
                    B_st = torch.zeros(Nn, K + pad_b)
                    B_st[:, :K] = Bm
                    ldb = K + pad_b
                else:
                    B_st = torch.zeros(K, Nn + pad_b)
                    B_st[:, :Nn] = Bm.t()
                    ldb = Nn + pad_b
                A_d, B_d, b_d = A_st.to(dev), B_st.to(dev), (b.to(dev) if b is not None else None)
                keep += [A_d, B_d, b_d]
                segs.append((A_d, lda, ak, B_d, ldb, bk, K, b_d))
            probs.append([None, Nn + pad_c, segs])
            refs.append(ref + (0.25 if acc else 0.0))
        outs = {}
        for name, flags in (('dma', 0), ('reg', n.GEMM_OPT_NO_DMA), ('lean', n.GEMM_OPT_LDS_LEAN)):
            res = []
            for p in probs:
                p[0] = torch.full((M, Nn + pad_c), 0.25 if acc else float('nan'), device=dev)
                res.append(p[0])
            n.gemm(M, Nn, [tuple(p) for p in probs], accumulate=acc, flags=flags)
            outs[name] = res
        for g in range(G):
            got = outs['dma'][g][:, :Nn]
            assert maxerr(got, refs[g]) < 1e-4 * max(1.0, sum(Ks) / 64), (case, ak, bk, M, Nn, Ks, G, acc)
            assert torch.equal(outs['dma'][g][:, :Nn], outs['reg'][g][:, :Nn]), (case, 'reg')
            assert torch.equal(outs['dma'][g][:, :Nn], outs['lean'][g][:, :Nn]), (case, 'lean')
            if pad_c and acc:
                assert bool((outs['dma'][g][:, Nn:] == 0.25).all())

"""The f32 GEMM on the bf16 matrix cores (csrc/rfn_gemm_x3.hip, RFN_GEMM_OPT_BF16X3): the plane images are an exact
three-way split in the documented layout, and the six-product GEMM is as close to an f64 product as an f32 product is.
Tolerances are relative to sum_k |a||b| (the scale of an f32 dot product's forward error): 2^-24 is one rounding."""
import pytest
import torch

pytestmark = pytest.mark.gpu

U = 2.0 ** -24


def _decode(img, rows, K):
    """image (uint8) -> three f32 planes [3][rows_pad][K_pad], following include/rfn.h's layout."""
    rows_pad, k_pad = (rows + 255) // 256 * 256, (K + 31) // 32 * 32
    nrb, nkc = rows_pad // 16, k_pad // 32
    assert img.numel() == rows_pad * k_pad * 6
    w = img.view(torch.int16).view(nkc, nrb, 3, 4, 16, 8)            # [kc][rb][plane][l / 16][l % 16][j]
    f = (w.to(torch.int32) << 16).view(torch.float32)
    return f.permute(2, 1, 4, 0, 3, 5).reshape(3, rows_pad, k_pad)   # [plane][rb, l % 16][kc, l / 16, j]


@pytest.mark.parametrize('k_fast', [True, False])
def test_plane_images_are_an_exact_split(dev, k_fast):
    import recurrent_fusion_network_amd._native as N
    g = torch.Generator(device='cpu').manual_seed(3)
    rows, K = 77, 45                      # ragged: neither a multiple of 32 / 16
    x = torch.randn(rows, K, generator=g) * torch.exp(4 * torch.randn(rows, K, generator=g))
    x[0, 0], x[1, 1], x[2, 2], x[3, 3] = 0.0, 1e-30, -3.0e38, 1.0 + 2.0 ** -23
    x[4, 4], x[5, 5] = 2.0 ** -100 * (1 + 2.0 ** -23), -(2.0 ** -127)    # low planes / the value itself denormal in bf16
    x[6, 6] = 3.4e38                                                    # rounds past the largest bf16: truncated plane
    src = x.to(dev) if k_fast else x.t().contiguous().to(dev)
    img = N.x3_image([src], rows, K, k_fast=k_fast)
    pl = _decode(img, rows, K).cpu()
    s = (pl[2].double() + pl[1].double() + pl[0].double())
    assert torch.equal(s[:rows, :K].float(), x), 'x0 + x1 + x2 must reproduce every finite f32 exactly'
    assert torch.equal(s[:rows, :K], x.double())
    assert float(s[rows:].abs().max()) == 0.0 and float(s[:, K:].abs().max()) == 0.0       # zero padding
    # plane magnitudes fall by 2^-8 each (round to nearest): |x1| <= 2^-7 |x0|, |x2| <= 2^-15 |x0| (2^-9, 2^-17 unless x0 was truncated)
    a0 = pl[0][:rows, :K].abs().double()
    assert bool((pl[1][:rows, :K].abs().double() <= a0 * 2.0 ** -7 + 1e-300).all())
    assert bool((pl[2][:rows, :K].abs().double() <= a0 * 2.0 ** -15 + 1e-300).all())


def test_plane_image_of_row_groups_and_non_finite_values(dev):
    import recurrent_fusion_network_amd._native as N
    g = torch.Generator(device='cpu').manual_seed(4)
    mats = [torch.randn(64, 40, generator=g) for _ in range(3)]
    img = N.x3_image([m.to(dev) for m in mats], 64, 40)
    pl = _decode(img, 192, 40).cpu()
    s = pl.double().sum(0)
    assert torch.equal(s[:192, :40].float(), torch.cat(mats, 0))
    assert float(s[192:].abs().max()) == 0.0
    bad = torch.tensor([[float('inf'), float('-inf'), float('nan'), 1.0]] * 32).to(dev)
    pl = _decode(N.x3_image([bad], 32, 4), 32, 4).cpu()
    assert bool(torch.isinf(pl[0][0, 0])) and bool(torch.isinf(pl[0][0, 1])) and bool(torch.isnan(pl[0][0, 2]))
    assert float(pl[1][:, :2].abs().max()) == 0.0 and float(pl[2][:, :3].abs().max()) == 0.0   # no inf - inf = nan planes


def _check(dev, M, N_, K, gm=None, gn=None, bias=False, accumulate=False, splitk=1, seed=0, wide=False):
    import recurrent_fusion_network_amd._native as N
    g = torch.Generator(device='cpu').manual_seed(seed)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N_, K, generator=g)
    if wide:
        a = a * torch.exp(3 * torch.randn(M, K, generator=g))
        b = b * torch.exp(3 * torch.randn(N_, K, generator=g))
    a, b = a.to(dev), b.to(dev)
    gm_, gn_ = gm or M, gn or N_
    ngm, ngn = -(-M // gm_), -(-N_ // gn_)
    outs = [torch.randn(min(gm_, M), min(gn_, N_), generator=g).to(dev) for _ in range(ngm * ngn)]
    prev = [o.clone() for o in outs]
    biases = [torch.randn(min(gn_, N_), generator=g).to(dev) for _ in range(ngm * ngn)] if bias else None
    ia, ib = N.x3_image([a], M, K), N.x3_image([b], N_, K)
    N.x3_gemm(M, N_, K, ia, ib, outs, gm=gm_, gn=gn_, ldc=min(gn_, N_), bias=biases, accumulate=accumulate, splitk=splitk)
    ref = a.double() @ b.double().t()
    mag = a.double().abs() @ b.double().abs().t()
    f32 = (a @ b.t()).double()                  # an f32 library product of the same operands, for scale
    worst = 0.0
    for i in range(ngm):
        for j in range(ngn):
            k = i * ngn + j
            r = ref[i * gm_:(i + 1) * gm_, j * gn_:(j + 1) * gn_]
            want = r + (biases[k].double() if bias else 0) + (prev[k].double() if accumulate else 0)
            err = (outs[k].double() - want).abs() / (mag[i * gm_:(i + 1) * gm_, j * gn_:(j + 1) * gn_] + 1e-30)
            worst = max(worst, float(err.max()))
    f32_worst = float(((f32 - ref).abs() / (mag + 1e-30)).max())
    return worst, f32_worst


@pytest.mark.parametrize('shape', [(300, 520, 70), (256, 256, 16), (1, 1, 1), (513, 257, 1000), (40, 2300, 333)])
def test_gemm_ragged_shapes_against_f64(dev, shape):
    worst, f32_worst = _check(dev, *shape, seed=sum(shape))
    assert worst <= 6 * U, (worst / U, f32_worst / U)


def test_gemm_output_groups_bias_accumulate(dev):
    # projection-style: one row group, three column groups of 256 with their own bias vectors
    assert _check(dev, 700, 768, 200, gn=256, bias=True, seed=1)[0] <= 8 * U
    # weight-gradient-style: row groups of 256, one column group, accumulated onto the previous contents
    assert _check(dev, 512, 300, 4000, gm=256, accumulate=True, seed=2)[0] <= 8 * U
    # both at once
    assert _check(dev, 512, 512, 64, gm=256, gn=256, bias=True, accumulate=True, seed=3)[0] <= 8 * U


@pytest.mark.parametrize('splitk', [2, 3, 7])
def test_gemm_split_k_is_deterministic_and_accurate(dev, splitk):
    import recurrent_fusion_network_amd._native as N
    worst, _ = _check(dev, 512, 260, 5000, gm=256, splitk=splitk, seed=5, bias=True)    # 313 pieces of K: uneven slices
    assert worst <= 6 * U
    g = torch.Generator(device='cpu').manual_seed(9)
    a, b = torch.randn(256, 3000, generator=g).to(dev), torch.randn(256, 3000, generator=g).to(dev)
    ia, ib = N.x3_image([a], 256, 3000), N.x3_image([b], 256, 3000)
    o1, o2 = torch.empty(256, 256, device=dev), torch.empty(256, 256, device=dev)
    N.x3_gemm(256, 256, 3000, ia, ib, [o1], splitk=splitk)
    N.x3_gemm(256, 256, 3000, ia, ib, [o2], splitk=splitk)
    assert torch.equal(o1, o2)


def test_gemm_wide_range_operands_no_worse_than_an_f32_product(dev):
    worst, f32_worst = _check(dev, 256, 256, 2048, seed=11, wide=True)
    assert worst <= max(2.0 * f32_worst, 8 * U), (worst / U, f32_worst / U)


def test_gemm_tail_round_of_quarter_tiles(dev):
    """More tiles than two rounds of the chip with a last round at most a quarter full: the remaining tiles are done as
    four quarter tiles each (csrc/rfn_gemm_x3.hip: main_tiles).  Row independence: the same rows through a short launch
    (no tail) give the same bits."""
    import recurrent_fusion_network_amd._native as N
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    tiles_n = 16
    tiles_m = (2 * cus + 16) // tiles_n + (1 if (2 * cus + 16) % tiles_n else 0)
    M, N_, K = tiles_m * 256 - 100, tiles_n * 256, 48
    g = torch.Generator(device='cpu').manual_seed(13)
    a, b = torch.randn(M, K, generator=g).to(dev), torch.randn(N_, K, generator=g).to(dev)
    ia, ib = N.x3_image([a], M, K), N.x3_image([b], N_, K)
    out = torch.empty(M, N_, device=dev)
    N.x3_gemm(M, N_, K, ia, ib, [out])
    ref = a.double() @ b.double().t()
    mag = a.double().abs() @ b.double().abs().t()
    assert float(((out.double() - ref).abs() / mag).max()) <= 16 * U     # the max over 3.4e7 outputs of a short (K = 48) product
    last = a[M - 300:].contiguous()
    out2 = torch.empty(300, N_, device=dev)
    N.x3_gemm(300, N_, K, N.x3_image([last], 300, K), ib, [out2])
    assert torch.equal(out2, out[M - 300:])


def test_gemm_rejects_groups_that_straddle_tiles(dev):
    import recurrent_fusion_network_amd._native as N
    a = torch.randn(256, 32, device=dev)
    ia = N.x3_image([a], 256, 32)
    outs = [torch.empty(256, 100, device=dev) for _ in range(3)]
    with pytest.raises(N.RfnError):
        N.x3_gemm(256, 256, 32, ia, ia, outs, gn=100)


def _decode_ks(img, K, cols):
    """k-slow image (uint8) -> three f32 planes [3][k_pad][m_pad]: element (k, plane, m) at ((k * 3 + plane) * Mp + m) * 2."""
    mp, k_pad = (cols + 255) // 256 * 256, (K + 31) // 32 * 32
    assert img.numel() == mp * k_pad * 6
    w = img.view(torch.int16).view(k_pad, 3, mp)
    return (w.to(torch.int32) << 16).view(torch.float32).permute(1, 0, 2)


def test_k_slow_image_is_an_exact_split_in_the_documented_layout(dev):
    import recurrent_fusion_network_amd._native as N
    g = torch.Generator(device='cpu').manual_seed(21)
    K, cols = 70, 36
    mats = [(torch.randn(K, cols, generator=g) * torch.exp(3 * torch.randn(K, cols, generator=g))) for _ in range(3)]
    img = N.x3_image_ks([m.to(dev) for m in mats], K, cols)
    pl = _decode_ks(img, K, 3 * cols).cpu().double().sum(0)
    assert torch.equal(pl[:K, :3 * cols].float(), torch.cat(mats, 1))
    assert float(pl[K:].abs().max()) == 0.0 and float(pl[:, 3 * cols:].abs().max()) == 0.0


def test_k_slow_image_with_more_than_65535_reduction_rows(dev):
    """K = B*L of the weight-gradient operands exceeds HIP's 65535 cap on gridDim.y from B = 335 at L = 196 (bench.py
    --batch 512): the split kernel carries k on grid.x."""
    import recurrent_fusion_network_amd._native as N
    K, cols = 70000, 8
    m = torch.randn(K, cols, generator=torch.Generator(device='cpu').manual_seed(3)).to(dev)
    img = N.x3_image_ks([m], K, cols)
    pl = _decode_ks(img, K, cols).double().sum(0)
    assert torch.equal(pl[:K, :cols].float(), m)
    assert float(pl[K:].abs().max()) == 0.0 and float(pl[:, cols:].abs().max()) == 0.0


@pytest.mark.parametrize('shape,splitk', [((300, 520, 70), 1), ((512, 260, 5000), 3), ((256, 256, 32), 1), ((40, 2300, 333), 2)])
def test_k_slow_gemm_against_f64(dev, shape, splitk):
    """C = A^T B with both operands stored reduction-index-major ([K][M], [K][N]): the weight-gradient orientation.  The MFMA
    operands come from transposing LDS reads of swizzled k-slow tiles."""
    import recurrent_fusion_network_amd._native as N
    M, N_, K = shape
    g = torch.Generator(device='cpu').manual_seed(sum(shape))
    a, b = torch.randn(K, M, generator=g).to(dev), torch.randn(K, N_, generator=g).to(dev)
    out = torch.full((M, N_), float('nan'), device=dev)
    N.x3_gemm(M, N_, K, N.x3_image_ks([a], K, M), N.x3_image_ks([b], K, N_), [out], splitk=splitk, k_slow=True)
    ref = a.double().t() @ b.double()
    mag = a.double().abs().t() @ b.double().abs()
    assert float(((out.double() - ref).abs() / mag).max()) <= 6 * U
    # the same product from fragment-order images of the transposed operands: same arithmetic per output element
    out2 = torch.empty(M, N_, device=dev)
    N.x3_gemm(M, N_, K, N.x3_image([a], M, K, k_fast=False), N.x3_image([b], N_, K, k_fast=False), [out2], splitk=splitk)
    assert torch.equal(out, out2)


def test_attention_backward_writes_the_plane_image_of_its_f32_result(dev):
    """rfn_attn_bwd_grouped_ks: the same launch as rfn_attn_bwd_grouped with d proj delivered as bf16 planes into k-slow
    images (one per encoder, this call's A columns at column ks_col0).  The planes must decode to the f32 values the plain
    launch writes (to one rounding: separate instantiations contract 1 - t * t differently), at rows k = b * L + l, and leave the other columns alone; dhproj / dw_part agree to rounding."""
    import recurrent_fusion_network_amd._native as N
    G, B, L, A, D, T = 2, 5, 50, 64, 96, 3           # image columns: T steps of A
    g_ = torch.Generator(device='cpu').manual_seed(31)
    rnd = lambda *s: torch.randn(*s, generator=g_).to(dev)  # noqa: E731
    proj, hp, w = [rnd(B, L, A) for _ in range(G)], [rnd(B, A) for _ in range(G)], [0.3 * rnd(A) for _ in range(G)]
    x, dz = [rnd(B, L, D) for _ in range(G)], [rnd(B, D) for _ in range(G)]
    al = [torch.softmax(rnd(B, L), 1).contiguous() for _ in range(G)]
    st = N.stream_ptr()
    new = lambda *shape: [torch.empty(*shape, device=dev) for _ in range(G)]  # noqa: E731
    dp, dhp, dwp = new(B, L, A), new(B, A), new(B, A)
    N.check(N.lib.rfn_attn_bwd_grouped(G, N.ptr_array(proj), L * A, A, N.ptr_array(hp), N.ptr_array(w), N.ptr_array(al),
                                       N.ptr_array(x), L * D, D, N.ptr_array(dz), D, B, L, A, D, N.ptr_array(dp), L * A, A,
                                       0, N.ptr_array(dhp), N.ptr_array(dwp), st), 'rfn_attn_bwd_grouped')
    K, cols = B * L, T * A
    mp = (cols + 255) // 256 * 256
    imgs = [torch.zeros(N.lib.rfn_x3_image_bytes(cols, K), dtype=torch.uint8, device=dev) for _ in range(G)]
    dhp2, dwp2 = new(B, A), new(B, A)
    N.check(N.lib.rfn_attn_bwd_grouped_ks(G, N.ptr_array(proj), L * A, A, N.ptr_array(hp), N.ptr_array(w), N.ptr_array(al),
                                          N.ptr_array(x), L * D, D, N.ptr_array(dz), D, B, L, A, D, N.ptr_array(imgs), mp, A,
                                          N.ptr_array(dhp2), N.ptr_array(dwp2), st), 'rfn_attn_bwd_grouped_ks')
    for g in range(G):
        s = _decode_ks(imgs[g], K, cols).double().sum(0)             # [k_pad][mp]
        got, want = s[:K, A:2 * A].float(), dp[g].reshape(K, A)
        # the two instantiations contract 1 - t * t differently (fma or not): one rounding of the largest terms apart
        assert float((got - want).abs().max()) <= 1.2e-7 * float(want.abs().max())
        assert float(s[:, :A].abs().max()) == 0.0 and float(s[:, 2 * A:].abs().max()) == 0.0 and float(s[K:].abs().max()) == 0.0
        for u, v in ((dhp[g], dhp2[g]), (dwp[g], dwp2[g])):            # column sums of those values: same caveat
            assert float((u - v).abs().max()) <= 1e-6 * float(u.abs().max())


def test_gemm_random_shapes_both_image_kinds(dev):
    """40 random problems (M, N up to ~900, K up to ~2000, any remainder, 1-5 K slices): fragment-order images and k-slow
    images of the same operands give the same bits, and both are right against f64."""
    import random
    import recurrent_fusion_network_amd._native as N
    rng = random.Random(5)
    g = torch.Generator(device='cpu').manual_seed(5)
    for case in range(40):
        M, N_ = rng.randint(1, 900), 4 * rng.randint(1, 225)
        K = rng.choice([1, 7, 31, 32, 33, 64, 100, 257, rng.randint(1, 2000)])
        steps = (K + 31) // 32
        splitk = rng.randint(1, min(5, steps))
        a, b = torch.randn(K, M, generator=g).to(dev), torch.randn(K, N_, generator=g).to(dev)      # both reduction-index-major
        out_f, out_k = torch.full((M, N_), float('nan'), device=dev), torch.full((M, N_), float('nan'), device=dev)
        N.x3_gemm(M, N_, K, N.x3_image([a], M, K, k_fast=False), N.x3_image([b], N_, K, k_fast=False), [out_f], splitk=splitk)
        if M % 4 == 0:
            N.x3_gemm(M, N_, K, N.x3_image_ks([a], K, M), N.x3_image_ks([b], K, N_), [out_k], splitk=splitk, k_slow=True)
            assert torch.equal(out_f, out_k), (case, M, N_, K, splitk)
        ref = a.double().t() @ b.double()
        mag = a.double().abs().t() @ b.double().abs()
        err = float(((out_f.double() - ref).abs() / (mag + 1e-30)).max())
        assert err <= 8 * U, (case, M, N_, K, splitk, err / U)


def test_greedy_decode_stays_with_the_exact_path_and_as_close_to_fp64(dev):
    """The flip-rate record in miniature (tools/x3_flip_rate.py, profiles/r04_x3_flip_rate*.json): on 256 random rows of a
    C2-shaped model, greedy decoding with the plane GEMM picks exactly the tokens the exact-f32 path picks; where either differs
    from an fp64 decode of the same model, the fp64 top1-top2 margin at that step is inside the f32 noise of the path; and the
    two modes sit at the same distance from the fp64 log-probs."""
    import recurrent_fusion_network_amd as R
    import recurrent_fusion_network_amd._native as N
    from oracle import rfn_oracle as O
    info = [dict(att_num=49, att_feat_size=512, fc_feat_size=512)] * 2
    cfg = O.make_cfg(info, vocab_size=9487, rnn_size=512, input_encoding_size=512, att_hid_size=512, num_review_steps_0=8,
                     num_review_steps=8, top_words_count=1000, seq_length=16)
    P = O.seeded_params(cfg, 77)
    rows = 256
    fc, att, _, _, _ = O.synthetic_batch(cfg, rows, seed=78)
    with torch.no_grad():
        seq64, _, lp64, _ = O.sample_greedy(cfg, {k: v.double() for k, v in P.items()}, [f.double() for f in fc],
                                            [x.double() for x in att])
    S, T = cfg.seq_length, lp64.size(1)
    ids64 = torch.zeros(rows, S, dtype=torch.long)
    ids64[:, :seq64.size(1)] = seq64
    top2 = lp64.topk(2, dim=2).values
    margin = top2[:, :, 0] - top2[:, :, 1]
    model = R.RecurrentFusionModel(cfg)
    model.load_state_dict(P)
    model = model.to(dev).eval()
    got, dist = {}, {}
    for name, flags in (('exact', 0), ('x3', N.GEMM_OPT_BF16X3 | N.GEMM_OPT_BF16X3_ANY_SIZE)):
        model.gemm_flags = flags
        with torch.no_grad():
            seq, _, lp, _ = model.sample([f.to(dev) for f in fc], [x.to(dev) for x in att], {'sample_max': 1})
        ids = torch.zeros(rows, S, dtype=torch.long)
        ids[:, :seq.size(1)] = seq.cpu()
        got[name] = ids
        differ = ids != ids64
        first = torch.where(differ.any(1), differ.float().argmax(1), torch.full((rows,), S))
        for r in torch.nonzero(differ.any(1)).flatten().tolist():      # a flip against fp64 needs a margin inside the noise
            assert float(margin[r, int(first[r])]) < 1e-5, (name, r, int(first[r]), float(margin[r, int(first[r])]))
        Tm = min(T, lp.size(1))
        ok = torch.arange(Tm)[None, :] <= first[:, None]
        d = (lp.cpu().double()[:, :Tm] - lp64[:, :Tm]).abs().amax(2)[ok]
        dist[name] = (float(d.median()), float(d.max()))
        assert dist[name][1] < 2e-5, (name, dist[name])
    assert torch.equal(got['exact'], got['x3'])
    assert dist['x3'][0] < 1.25 * dist['exact'][0] and dist['x3'][1] < 1.5 * dist['exact'][1], dist

"""Diagnostic (round 3): do the MFMA-bound per-encoder gate GEMMs of a stage-I step hide under the HBM-bound attention
kernels of ANOTHER encoder when the two run on two HIP streams?  (The M cells of a step are independent, so encoder i's
gate GEMM could run beside encoder i+1's attention.)  Chains of 16 launches each, alone and together."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import recurrent_fusion_network_amd._native as N

dev = torch.device('cuda:0')
B, L, D, A, R = 256, 196, 2048, 512, 512
X = [torch.randn(B, L, D, device=dev) for _ in range(3)]
alpha = torch.softmax(torch.randn(B, L, device=dev), 1).contiguous()
z = torch.empty(B, D, device=dev)
H = torch.randn(B, 4 * R, device=dev)
Z = torch.randn(B, D, device=dev)
WH = torch.randn(4 * R, 4 * R, device=dev) * 0.05
WZ = torch.randn(4 * R, D, device=dev) * 0.05
G = torch.empty(4, B, 4 * R, device=dev)
ws = torch.empty(48 << 20, dtype=torch.uint8, device=dev)
P1 = torch.randn(B * L, A, device=dev)
hp = torch.randn(B, A, device=dev)
w = torch.randn(A, device=dev) * 0.1
dz = torch.randn(B, D, device=dev)
dhp = torch.empty(B, A, device=dev)
dwp = torch.empty(B, A, device=dev)
lib = N.lib


def chain_gemm(n, groups=1):
    for _ in range(n):
        N.gemm(B, 4 * R, [(G[g], 4 * R, [(H, 4 * R, 1, WH, 4 * R, 1, 4 * R, None), (Z, D, 1, WZ, D, 1, D, None)]) for g in range(groups)],
               ws=ws)


def chain_ctx(n):
    st = N.stream_ptr()
    for i in range(n):
        N.check(lib.rfn_attn_context_fwd(X[i % 3].data_ptr(), L * D, D, alpha.data_ptr(), B, L, D, z.data_ptr(), D, st))


def chain_bwd(n):
    st = N.stream_ptr()
    for i in range(n):
        N.check(lib.rfn_attn_bwd(P1.data_ptr(), L * A, A, hp.data_ptr(), w.data_ptr(), alpha.data_ptr(), X[i % 3].data_ptr(), L * D, D,
                                 dz.data_ptr(), D, B, L, A, D, P1.data_ptr(), L * A, A, 0, dhp.data_ptr(), dwp.data_ptr(), st))


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


side = torch.cuda.Stream()


def both(a, b):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    a()
    with torch.cuda.stream(side):
        b()
    main.wait_stream(side)


for gname, g in (('16 x gate GEMM (1 encoder, 4.3 GF)', lambda: chain_gemm(16, 1)), ('16 x gate GEMM (4 encoders, 17 GF)', lambda: chain_gemm(16, 4))):
    tg = timeit(g)
    for cname, c in (('16 x attn_context_fwd (411 MB)', lambda: chain_ctx(16)), ('16 x attn_bwd fused (616 MB r+w)', lambda: chain_bwd(16))):
        tc, tb = timeit(c), timeit(lambda: both(c, g))
        print('%-36s %.2f ms | %-34s %.2f ms | together %.2f ms | hidden %.2f ms = %.0f %% of the GEMM chain' % (
            gname, tg, cname, tc, tb, tg + tc - tb, 100 * (tg + tc - tb) / tg), flush=True)

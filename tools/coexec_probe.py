"""Diagnostic: how fast does a dependent chain of recurrence kernels run on a second HIP stream while the big grouped
projection GEMM occupies the chip on the first?  (Stream overlap only pays if the chain keeps close to its stand-alone
speed.  The round-2 runs recorded in profiles/r02_overlap_probes.md also varied how many GEMM blocks a CU hosts through a
probe-only build option, RFN_GEMM_COOP_LDS_KB, which padded the GEMM's LDS request; the product has no such knob.)  Prints, per chain kind: chain alone, GEMM alone, both together, and the overlap that was realised."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import recurrent_fusion_network_amd._native as N

dev = torch.device('cuda:0')
B, L, D, A, T1, R = 256, 196, 2048, 512, 8, 512
X = [torch.randn(B, L, D, device=dev) for _ in range(4)]
W = [torch.randn(A, D, device=dev) * 0.1 for _ in range(T1)]
P = torch.empty(B * L, T1 * A, device=dev)
P2 = [torch.randn(B * L, T1 * A, device=dev) for _ in range(2)]
dW = [torch.empty(A, D, device=dev) for _ in range(T1)]
hp = torch.randn(B, A, device=dev)
w = torch.randn(A, device=dev) * 0.1
bo = torch.zeros(1, device=dev)
alpha = torch.softmax(torch.randn(B, L, device=dev), 1).contiguous()
z = torch.empty(B, D, device=dev)
dz = torch.randn(B, D, device=dev)
dhp = torch.empty(B, A, device=dev)
dwp = torch.empty(B, A, device=dev)
H = torch.randn(B, 4 * R, device=dev)
Wg = torch.randn(4 * R, 4 * R, device=dev) * 0.05
G = torch.empty(B, 4 * R, device=dev)
Wh = torch.randn(A, R, device=dev) * 0.05
ws = torch.empty(48 << 20, dtype=torch.uint8, device=dev)
c0 = torch.randn(B, R, device=dev)
c1 = torch.empty(B, R, device=dev)
h1 = torch.empty(B, R, device=dev)
lib = N.lib


def gemm_nt():
    N.gemm(B * L, A, [(P[:, t * A:], T1 * A, [(X[0], D, 1, W[t], D, 1, D, None)]) for t in range(T1)])


def gemm_tn():
    N.gemm(A, D, [(dW[t], D, [(P2[0][:, t * A:], T1 * A, 0, X[0], D, 0, B * L, None)]) for t in range(T1)])


def chain_tiny(n):
    st = N.stream_ptr()
    for _ in range(n):
        N.check(lib.rfn_lstm_fwd(G.data_ptr(), 4 * R, c0.data_ptr(), R, c1.data_ptr(), R, h1.data_ptr(), R, B, R, 0, 0.0, 0, 0, st))


def chain_gemm(n):       # the stage-I gate GEMM (M = 256 rows, N = 2048, K = 2048): split-K + reduce
    for _ in range(n):
        N.gemm(B, 4 * R, [(G, 4 * R, [(H, 4 * R, 1, Wg, 4 * R, 1, 4 * R, None)])], ws=ws)


def chain_ctx(n):        # HBM-bound: 411 MB of features per launch
    st = N.stream_ptr()
    for i in range(n):
        N.check(lib.rfn_attn_context_fwd(X[1 + i % 3].data_ptr(), L * D, D, alpha.data_ptr(), B, L, D, z.data_ptr(), D, st))


def chain_bwd(n):        # HBM-bound fused attention backward of one (step, encoder)
    st = N.stream_ptr()
    for i in range(n):
        p = P2[1]
        N.check(lib.rfn_attn_bwd(p.data_ptr(), L * T1 * A, T1 * A, hp.data_ptr(), w.data_ptr(), alpha.data_ptr(),
                                 X[1 + i % 3].data_ptr(), L * D, D, dz.data_ptr(), D, B, L, A, D, p.data_ptr(), L * T1 * A,
                                 T1 * A, 0, dhp.data_ptr(), dwp.data_ptr(), st))


def timeit(fn, reps=3):
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


def both(g, c):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    g()
    with torch.cuda.stream(side):
        c()
    main.wait_stream(side)


for gname, g in (('NT proj', gemm_nt), ('TN dW', gemm_tn)):
    tg = timeit(g)
    for cname, c in (('200 x lstm_fwd', lambda: chain_tiny(200)), ('40 x gate GEMM M=256', lambda: chain_gemm(40)),
                     ('12 x attn_context_fwd', lambda: chain_ctx(12)), ('8 x attn_bwd (fused)', lambda: chain_bwd(8))):
        tc, tb = timeit(c), timeit(lambda: both(g, c))
        print('%-8s %.2f ms | %-22s alone %.2f ms | together %.2f ms | hidden %.2f ms (%.0f %% of the chain)' % (
            gname, tg, cname, tc, tb, tg + tc - tb, 100 * (tg + tc - tb) / tc))

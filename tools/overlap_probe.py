"""Diagnostic: do the MFMA-bound big GEMM and the HBM-bound attention kernels overlap when issued on two streams?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import recurrent_fusion_network_amd._native as N

dev = torch.device('cuda:0')
B, L, D, A, T1 = 256, 196, 2048, 512, 8
X = torch.randn(B, L, D, device=dev)
W = [torch.randn(A, D, device=dev) * 0.1 for _ in range(T1)]
P = torch.empty(B * L, T1 * A, device=dev)
dW = [torch.empty(A, D, device=dev) for _ in range(T1)]
hp = torch.randn(B, A, device=dev); w = torch.randn(A, device=dev) * 0.1; bo = torch.zeros(1, device=dev)
alpha = torch.empty(B, L, device=dev); z = torch.empty(B, D, device=dev); dz = torch.randn(B, D, device=dev)
dal = torch.empty(B, L, device=dev); dhp = torch.empty(B, A, device=dev); dwp = torch.empty(B, A, device=dev)

def gemm_nt(groups):   # projection for `groups` step weights
    N.gemm(B * L, A, [(P[:, t * A:], T1 * A, [(X, D, 1, W[t], D, 1, D, None)]) for t in range(groups)])
def gemm_tn(groups):
    N.gemm(A, D, [(dW[t], D, [(P[:, t * A:], T1 * A, 0, X, D, 0, B * L, None)]) for t in range(groups)])
def attn_fwd(n):
    st = N.stream_ptr()
    for _ in range(n):
        N.check(N.lib.rfn_attn_scores_fwd(P.data_ptr(), L * T1 * A, T1 * A, hp.data_ptr(), w.data_ptr(), bo.data_ptr(), B, L, A, alpha.data_ptr(), st))
        N.check(N.lib.rfn_attn_context_fwd(X.data_ptr(), L * D, D, alpha.data_ptr(), B, L, D, z.data_ptr(), D, st))
def attn_bwd(n):
    st = N.stream_ptr()
    for _ in range(n):
        N.check(N.lib.rfn_attn_context_bwd_dalpha(X.data_ptr(), L * D, D, dz.data_ptr(), D, B, L, D, dal.data_ptr(), st))
        N.check(N.lib.rfn_attn_scores_bwd(P.data_ptr(), L * T1 * A, T1 * A, hp.data_ptr(), w.data_ptr(), alpha.data_ptr(), dal.data_ptr(), B, L, A,
                                          P.data_ptr(), L * T1 * A, T1 * A, 0, dhp.data_ptr(), dwp.data_ptr(), st))

def timeit(fn, reps=3):
    torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3

s1 = torch.cuda.Stream()
def both(g, a):
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s1):
        s1.wait_event(ev); g()
    a()
    torch.cuda.current_stream().wait_stream(s1)

gemm_nt(8); gemm_tn(8); attn_fwd(1); attn_bwd(1)
for name, g, a in [('fwd: NT proj (8 groups) + 32 x (scores+context)', lambda: gemm_nt(8), lambda: attn_fwd(32)),
                   ('bwd: TN dW (8 groups) + 32 x (dalpha+scores_bwd)', lambda: gemm_tn(8), lambda: attn_bwd(32)),
                   ('bwd: TN dW (4 groups) + 16 x (dalpha+scores_bwd)', lambda: gemm_tn(4), lambda: attn_bwd(16))]:
    tg, ta, tb = timeit(g), timeit(a), timeit(lambda: both(g, a))
    print('%-52s gemm %.2f ms  attn %.2f ms  sum %.2f  concurrent %.2f ms  (saved %.2f)' % (name, tg, ta, tg + ta, tb, tg + ta - tb))

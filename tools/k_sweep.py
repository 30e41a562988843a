"""Time against K of the medium launches whose K is short (the K = B stage-I weight gradients: 8 groups of 2048 x 2048, TN;
the dz products: 4 groups of 256 x 2048, NN, split-K): T(K) = a + b K.  `a` is what a launch pays for its tiles' pro- and
epilogues, i.e. the most a cross-tile prefetch / persistent-tile form could recover.   python tools/k_sweep.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from recurrent_fusion_network_amd import _native as nv
dev = 'cuda'
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
def run(name, M, N, ak, bk, ng, Ks):
    rows = []
    for K in Ks:
        probs, keep = [], []
        for g in range(ng):
            C = torch.empty(M, N, device=dev)
            A = torch.randn((M, K) if ak else (K, M), device=dev); Bm = torch.randn((N, K) if bk else (K, N), device=dev)
            keep += [A, Bm, C]
            probs.append((C, N, [(A, K if ak else M, ak, Bm, K if bk else N, bk, K, None)]))
        for _ in range(10): nv.gemm(M, N, probs, ws=ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): nv.gemm(M, N, probs, ws=ws)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 50
        rows.append((K, us, 2.0 * M * N * K * ng / us / 1e6))
    (k0, t0, _), (k1, t1, _) = rows[0], rows[-1]
    b = (t1 - t0) / (k1 - k0); a = t0 - b * k0
    print('%s: %s | fit: a = %.1f us, b = %.1f ns per k (= %.0f TF marginal)' % (
        name, '  '.join('K=%d %.1f us (%.0f TF)' % r for r in rows), a, b * 1e3, 2.0 * M * N * ng / b / 1e6))
run('part-A weight gradient, 8 x (2048 x 2048), TN', 2048, 2048, 0, 0, 8, [256, 512, 1024, 2048])
run('dz, 4 x (256 x 2048), NN', 256, 2048, 1, 0, 4, [1024, 2048, 4096, 8192])
run('stage-I gates, 4 x (256 x 2048), NT', 256, 2048, 1, 1, 4, [2048, 4096, 8192])
run('logit dX, 4352 x 512, NN', 4352, 512, 1, 0, 1, [4096, 9488, 18976])

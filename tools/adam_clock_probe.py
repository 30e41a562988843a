"""The fused clamp+Adam launch over 390 M parameters (the C3 model) behind different neighbours: back to back, behind 600 tiny
launches, behind 20 ms of idle, behind 7 ms of tiny launches + 3.3 ms of the B = 32 weight-gradient GEMMs (what a 32-caption
shard's step puts in front of it).  profiles/r04_small_shards.md section 3.   python tools/adam_clock_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from recurrent_fusion_network_amd import _native as nv
dev = 'cuda'
n = 390_000_000
p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
v.abs_()
st = nv.stream_ptr()
L, D, A, T, B = 196, 2048, 512, 8, 32
BL = B * L
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
X = torch.randn(BL, D, device=dev); P = torch.randn(T, BL, A, device=dev); dW = [torch.empty(A, D, device=dev) for _ in range(T)]
tn = [(dW[t], D, [(P[t], A, 0, X, D, 0, BL, None)]) for t in range(T)]
small = torch.zeros(64, device=dev)
def adam(): nv.check(nv.lib.rfn_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, 5e-4, 0.9, 0.999, 1e-8, 1e-5, 1.0, 1.0, 3, st))
def tiny(k):
    for _ in range(k): small.add_(1.0)
def shard_like():
    tiny(600)
    for _ in range(4): nv.gemm(A, D, tn, ws=ws)
def idle():
    torch.cuda.synchronize(); time.sleep(0.02)
def t(pre, reps=8):
    tot = 0
    for i in range(reps + 2):
        pre(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); adam(); e1.record(); torch.cuda.synchronize()
        if i >= 2: tot += e0.elapsed_time(e1)
    return tot / reps
def b2b():
    adam()
print('clamp+Adam over %d M parameters (%.1f GB moved): behind itself %.3f ms | behind 600 tiny launches %.3f | behind 20 ms idle %.3f | '
      'behind 600 tiny launches + 4 B=32 weight-gradient GEMMs %.3f | behind 4 such GEMMs only %.3f'
      % (n // 1_000_000, n * 28 / 1e9, t(b2b), t(lambda: tiny(600)), t(idle), t(shard_like), t(lambda: [nv.gemm(A, D, tn, ws=ws) for _ in range(4)])))

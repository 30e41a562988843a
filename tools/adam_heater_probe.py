"""Why is the identical clamp+Adam launch (10.9 GB moved) 2.0 ms inside a B = 256 step and 2.35-2.45 ms standalone or inside a
B = 32 step?  The counters say the bytes are the same (profiles/r05_pmc_step.md: 2 x FETCH 6.23 GB, WRITE 4.67 GB, L2 hit 55 %
at both batch sizes).  This probe puts k back-to-back dense f32 weight-gradient GEMMs of the B = 256 step (5.9 ms each, the
chip at its power limit) in front of the launch: if its time falls with the length of that heater, what the small shard
lacks is the sustained-load power state, not anything in the kernel.   python tools/adam_heater_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from recurrent_fusion_network_amd import _native as nv
dev = 'cuda'
n = 390_000_000
p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
v.abs_()
st = nv.stream_ptr()
L, D, A, T = 196, 2048, 512, 8
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)


def heater_of(B):
    BL = B * L
    X = torch.randn(BL, D, device=dev); P = torch.randn(T, BL, A, device=dev); dW = [torch.empty(A, D, device=dev) for _ in range(T)]
    return [(dW[t], D, [(P[t], A, 0, X, D, 0, BL, None)]) for t in range(T)]


big, small = heater_of(256), heater_of(32)
stream_src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
stream_dst = torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def adam():
    nv.check(nv.lib.rfn_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, 5e-4, 0.9, 0.999, 1e-8, 1e-5, 1.0, 1.0, 3, st))


def timed(pre, reps=6):
    ts = []
    for i in range(reps + 2):
        pre(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); adam(); e1.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(e0.elapsed_time(e1))
    return sum(ts) / len(ts), min(ts)


rows = [('nothing (behind the previous measurement)', lambda: None)]
for k in (1, 2, 4, 8):
    rows.append(('%d x the B=256 weight-gradient GEMM (%.0f ms of dense f32 MFMA)' % (k, 5.9 * k), lambda k=k: [nv.gemm(A, D, big, ws=ws) for _ in range(k)]))
for k in (4, 32):
    rows.append(('%d x the B=32 weight-gradient GEMM (%.1f ms)' % (k, 0.8 * k), lambda k=k: [nv.gemm(A, D, small, ws=ws) for _ in range(k)]))
for k in (4, 16):
    rows.append(('%d x a 1 GiB device copy (HBM-bound heater, %.1f ms)' % (k, 0.4 * k), lambda k=k: [stream_dst.copy_(stream_src) for _ in range(k)]))
print('clamp+Adam over %d M parameters (%.1f GB moved), launch timed by HIP events; mean / min ms over 6 runs' % (n // 1_000_000, n * 28 / 1e9))
for name, pre in rows:
    mean, best = timed(pre)
    print('  behind %-72s %.3f / %.3f' % (name, mean, best))

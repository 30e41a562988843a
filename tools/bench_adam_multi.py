"""Times rfn_adam_step_multi on the C3 bucket mix (decoder 54 MB, core 262 MB, four encoders a 277 MB + 34 MB: 1.56 GB per
stream, 7 streams) and prints the achieved HBM rate.  Diagnostic only (csrc/rfn_misc.hip ADAM_CONTIG / ADAM_NT A/B)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recurrent_fusion_network_amd import _native as n  # noqa: E402

dev = torch.device('cuda:0')
MB = 1000 * 1000 // 4
sizes = [54 * MB, 262 * MB] + [277 * MB] * 4 + [34 * MB] * 4
sizes = [s // 4 * 4 for s in sizes]
ps = [torch.randn(s, device=dev) * 0.1 for s in sizes]
gs = [torch.randn(s, device=dev) * 0.01 for s in sizes]
ms = [torch.zeros(s, device=dev) for s in sizes]
vs = [torch.zeros(s, device=dev) for s in sizes]
st = n.stream_ptr()
arr = (C.c_int64 * len(sizes))(*sizes)
ptrs = (n.ptr_array(ps), n.ptr_array(gs), n.ptr_array(ms), n.ptr_array(vs))


def step(k):
    n.check(n.lib.rfn_adam_step_multi(len(sizes), *ptrs, arr, 5e-4, 0.9, 0.999, 1e-8, 1e-5, 1.0, 1.0, k, st))


for k in range(1, 4):
    step(k)
torch.cuda.synchronize()
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(4, 14):
        step(k)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    tot = sum(sizes)
    print('rfn_adam_step_multi, %d buckets, %.2f GB per stream: %.3f ms = %.2f TB/s (28 B per parameter)' % (
        len(sizes), tot * 4 / 1e9, t, 28.0 * tot / t / 1e9))

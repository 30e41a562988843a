"""Times rfn_adam_step on a C3-sized flat bucket (350 M parameters = 1.4 GB per stream, 7 streams) and prints the
achieved HBM rate (28 B per parameter).  Diagnostic only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recurrent_fusion_network_amd import _native as n  # noqa: E402

dev = torch.device('cuda:0')
N = 350 * 1000 * 1000
p = torch.randn(N, device=dev) * 0.1
g = torch.randn(N, device=dev) * 0.01
m = torch.zeros(N, device=dev)
v = torch.zeros(N, device=dev)
st = n.stream_ptr()


def step(k):
    n.check(n.lib.rfn_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), N, 5e-4, 0.9, 0.999, 1e-8, 1e-5,
                                1.0, 1.0, k, st))


for k in range(1, 4):
    step(k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(4, 14):
    step(k)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print('rfn_adam_step: %.3f ms for %d M parameters = %.2f TB/s (28 B per parameter)' % (ms, N // 1000000, 28.0 * N / ms / 1e9))

"""Fixed cost of the two long stage-I products against the batch: the hoisted att_2_att_h projection (NT, M = B*L) and its
weight gradient (TN, K = B*L) at B = 16 ... 256, (a) back to back, (b) each launch behind a 1.6 GB streaming kernel (what
precedes them in the train step: HBM-bound attention / Adam).  Fits time = X + flops / rate.
    python tools/gemm_fixed_cost.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from recurrent_fusion_network_amd import _native as nv  # noqa: E402

dev = 'cuda'
L, D, A, T = 196, 2048, 512, 8


def timed(fn, reps, pre=None):
    for _ in range(3):
        if pre:
            pre()
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        if pre:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3


def main():
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    big = torch.empty(400 << 20, dtype=torch.float32, device=dev)
    flush = lambda: big.mul_(1.0001)  # noqa: E731
    Wt = [torch.randn(A, D, device=dev) * 0.1 for _ in range(T)]
    rows = []
    for B in (16, 32, 64, 128, 256):
        BL = B * L
        X = torch.randn(BL, D, device=dev)
        P = torch.empty(T, BL, A, device=dev)
        dW = [torch.empty(A, D, device=dev) for _ in range(T)]
        nt = [(P[t], A, [(X, D, 1, Wt[t], D, 1, D, None)]) for t in range(T)]
        tn = [(dW[t], D, [(P[t], A, 0, X, D, 0, BL, None)]) for t in range(T)]
        flops = 2.0 * BL * D * A * T
        r = [B, flops]
        for probs, M_, N_ in ((nt, BL, A), (tn, A, D)):
            f = lambda: nv.gemm(M_, N_, probs, ws=ws)  # noqa: E731
            r += [timed(f, 10), timed(f, 10, flush)]
        rows.append(r)
        print('B=%3d  %.4f TF | NT back-to-back %7.1f us (%5.1f TF)  behind a stream %7.1f us | TN %7.1f us (%5.1f TF)  behind a stream %7.1f us'
              % (B, flops / 1e12, r[2], flops / r[2] / 1e6, r[3], r[4], flops / r[4] / 1e6, r[5]), flush=True)
    for col, name in ((2, 'NT b2b'), (3, 'NT cold'), (4, 'TN b2b'), (5, 'TN cold')):
        (b0, f0), (b1, f1) = (rows[1][col], rows[1][1]), (rows[-1][col], rows[-1][1])
        rate = (f1 - f0) / (b1 - b0)          # flop per us
        print('%s: rate %.1f TF, fixed cost %.0f us (fit through B=32 and B=256)' % (name, rate / 1e6, b0 - f0 / rate))


if __name__ == '__main__':
    main()

"""Do the per-step products of stages I / II pay for COLD weights?  Every step of those stages has weights of its own, read once
per forward and once per backward of a train step, so inside the step they come from HBM (or the memory-side cache), not from
L2 -- unlike tools/bench_cellgemm.py, which launches the same product on the same weights over and over.  This probe times
rfn_cell_gemm at BASELINE config 2's shapes (B = 64) on ONE weight set (warm) and rotating through `sets` weight sets whose
total size exceeds every cache (cold), back to back.

    python tools/cold_weights_probe.py [--reps 256] [--sets 64] [--batch 64]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import recurrent_fusion_network_amd._native as N  # noqa: E402

dev = torch.device('cuda:0')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=256)
    ap.add_argument('--sets', type=int, default=64)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--variants', type=lambda x: [int(v) for v in x.split(',')], default=[0], help='tile variants to force (0 = the library picks)')
    a = ap.parse_args()
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.1  # noqa: E731
    R, M = 512, a.batch
    cases = [  # name, outputs [(N, [K...])], lstm?, b_kfast
        ('stage II K3 (C2: 2 x z_2_h -> lstm)', [(2048, [512, 512])], True, 1),
        ('stage II K1 (C2: 2 hp | g = h2h(h))', [(512, [512])] * 2 + [(2048, [512])], False, 1),
        ('stage I gates (C2: 2 enc, K = 1024 + 512 -> lstm)', [(2048, [1024, 512])] * 2, True, 1),
        ('stage II bwd Kb1 (C2: dhrec, 2 dz; K = 2048)', [(512, [2048])] * 3, False, 0),
        ('stage I bwd X (C2: 2 slabs N = 1024, 2 dz; K = 2048)', [(1024, [2048])] * 2 + [(512, [2048])] * 2, False, 0),
        ('stage II K3 (C3: 4 x z_2_h -> lstm)', [(2048, [512] * 4)], True, 1),
    ]
    st = N.stream_ptr()
    for name, outs_spec, lstm, bkf in cases:
        wbytes = sum(4.0 * n * sum(ks) for n, ks in outs_spec)
        argsets, keep = [], []
        As = [[rnd(M, k) for k in ks] for n, ks in outs_spec]      # the activations are the same (warm) in both runs
        for s in range(a.sets):
            outs = []
            for (n, ks), Ao in zip(outs_spec, As):
                C = torch.zeros(M, n, device=dev)
                segs = []
                for k, A in zip(ks, Ao):
                    W = rnd(n, k) if bkf else rnd(k, n)
                    segs.append((A, k, W, k if bkf else n, bkf, k, rnd(n)))
                o = dict(C=C, ldc=n, N=n, accumulate=1 if lstm else 0, segs=segs)
                if lstm:
                    o['lstm'] = (rnd(M, R), R, torch.empty(M, R, device=dev), R, torch.empty(M, R, device=dev), R, 7)
                outs.append(o)
            argsets.append((outs, N.cell_gemm_args(outs)))

        def run(which, variant=0):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for warm in range(2):
                if warm == 1:
                    e0.record()
                for r in range(a.reps):
                    outs, arr = argsets[which(r)]
                    N.check(N.lib.rfn_cell_gemm(M, len(outs), arr, R, 0.0, 0, variant, st))
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / a.reps
        for v in a.variants:
            warm = run(lambda r: 0, v)
            cold = run(lambda r: r % a.sets, v)
            print('%-56s weights %5.1f MB | variant %d | warm %6.2f us | cold %6.2f us (%.2f TB/s of weights)' % (
                name, wbytes / 1e6, v, warm, cold, wbytes / cold / 1e6), flush=True)


if __name__ == '__main__':
    main()

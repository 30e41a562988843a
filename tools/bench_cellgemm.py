"""Times rfn_cell_gemm (csrc/rfn_cellgemm.hip) per tile variant at the per-step shapes of the recurrences, next to the
launch sequence it replaces (64x64-tile GEMM + split-K reduce [+ lstm_fwd]).  HIP events over back-to-back launches.

    python tools/bench_cellgemm.py [--reps 200]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import recurrent_fusion_network_amd._native as N  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps       # microseconds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=200)
    ap.add_argument('--floor', action='store_true')
    ap.add_argument('--small', action='store_true')
    ap.add_argument('--batch', type=int, default=0, help='run the B = 256 cases at this batch size instead')
    ap.add_argument('--slots', type=lambda x: [int(v) for v in x.split(',')], default=[2, 3, 4, 6])
    ap.add_argument('--variants', type=lambda x: [int(v) for v in x.split(',')], default=[1, 2, 3])
    a = ap.parse_args()
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.1  # noqa: E731
    ws = torch.empty(48 << 20, dtype=torch.uint8, device=dev)
    R = 512
    cases = [  # name, M, outputs [(N, [K...])], lstm?, b_kfast
        ('decoder K1: hp | g += h2h(h)', 256, [(512, [512]), (2048, [512])], False, 1),
        ('decoder K3: g += z2h(z) -> lstm', 256, [(2048, [512])], True, 1),
        ('stage II K1: 4 hp | g = h2h(h)', 256, [(512, [512])] * 4 + [(2048, [512])], False, 1),
        ('stage II K3: g += sum z_2_h -> lstm', 256, [(2048, [512] * 4)], True, 1),
        ('decoder K3 at B=64', 64, [(2048, [512])], True, 1),
        ('decoder K1 at B=64', 64, [(512, [512]), (2048, [512])], False, 1),
        ('decoder bwd Kb1 at B=64', 64, [(512, [2048]), (512, [2048])], False, 0),
        ('decoder bwd Kb2 at B=64', 64, [(512, [512])], False, 0),
        ('decoder K3 at B=640 (beam)', 640, [(2048, [512])], True, 1),
        ('decoder bwd: dz | dhrec = dg . W', 256, [(512, [2048]), (512, [2048])], False, 0),
        ('decoder bwd: dhrec += dhp . W', 256, [(512, [512])], False, 0),
        ('stage II bwd: dhrec, 4 dz', 256, [(512, [2048])] * 5, False, 0),
    ]
    if a.batch:
        cases = [('B=%d %s' % (a.batch, nm), a.batch, o, l, k) for nm, m, o, l, k in cases if m == 256]
    if a.small:      # the small-shard regime (B = 32 / 64 per GPU; BASELINE config 2): grids far below 256 CUs
        cases = []
        for Ms in (32, 64):
            cases += [
                ('B=%d decoder K1: hp | g += h2h(h)' % Ms, Ms, [(512, [512]), (2048, [512])], False, 1),
                ('B=%d decoder K3 -> lstm' % Ms, Ms, [(2048, [512])], True, 1),
                ('B=%d stage II K1: 4 hp | g' % Ms, Ms, [(512, [512])] * 4 + [(2048, [512])], False, 1),
                ('B=%d stage II K3 -> lstm' % Ms, Ms, [(2048, [512] * 4)], True, 1),
                ('B=%d decoder bwd Kb1' % Ms, Ms, [(512, [2048]), (512, [2048])], False, 0),
                ('B=%d decoder bwd Kb2' % Ms, Ms, [(512, [512])], False, 0),
                ('B=%d stage II bwd: dhrec, 4 dz' % Ms, Ms, [(512, [2048])] * 5, False, 0),
                ('B=%d stage I gates 4 enc K=4096 -> lstm' % Ms, Ms, [(2048, [2048, 2048])], True, 1),
                ('B=%d stage I dz 4 enc N=2048 K=2048' % Ms, Ms, [(2048, [2048])] * 4, False, 0),
                ('B=%d stage I gates 2 enc K=1024+512 -> lstm (C2)' % Ms, Ms, [(2048, [1024, 512])] * 2, True, 1),
            ]
    if a.floor:      # fixed cost vs K slope: one output N = 2048 at M = 256, both epilogues
        cases = [('K=%d %s' % (k, 'lstm' if l else 'store'), 256, [(2048, [k])], l, 1) for l in (False, True)
                 for k in (64, 256, 512, 1024, 2048)]
    for name, M, outs_spec, lstm, bkf in cases:
        flops = sum(2.0 * M * n * sum(ks) for n, ks in outs_spec)
        outs, probs = [], []
        for n, ks in outs_spec:
            C = torch.zeros(M, n, device=dev)
            segs, psegs = [], []
            for k in ks:
                A = rnd(M, k)
                W = rnd(n, k) if bkf else rnd(k, n)
                b = rnd(n)
                segs.append((A, k, W, k if bkf else n, bkf, k, b))
                psegs.append((A, k, 1, W, k if bkf else n, bkf, k, b))
            o = dict(C=C, ldc=n, N=n, accumulate=1 if lstm else 0, segs=segs)
            if lstm:
                cp, cn, hn = rnd(M, R), torch.empty(M, R, device=dev), torch.empty(M, R, device=dev)
                o['lstm'] = (cp, R, cn, R, hn, R, 7)
            outs.append(o)
            probs.append((C, n, psegs))
        line = '%-40s %6.2f GF |' % (name, flops / 1e9)

        def old():
            for C, n, psegs in probs:
                N.gemm(M, n, [(C, n, psegs)], accumulate=lstm, ws=ws)
            if lstm:
                o = outs[0]
                cp, _, cn, _, hn, _, _ = o['lstm']
                N.check(N.lib.rfn_lstm_fwd(o['C'].data_ptr(), 4 * R, cp.data_ptr(), R, cn.data_ptr(), R, hn.data_ptr(), R, M, R,
                                           0, 0.0, 0, 0, N.stream_ptr()))
        line += ' old %6.1f us |' % timeit(old, a.reps)
        arr, st = N.cell_gemm_args(outs), N.stream_ptr()
        for v in [0] + [t + 16 * sl for t in a.variants for sl in a.slots]:
            if N.lib.rfn_cell_gemm(M, len(outs), arr, R, 0.0, 0, v, st) != 0:
                line += ' v%d/%d   -- ' % (v & 15, v >> 4)
                continue
            t = timeit(lambda: N.lib.rfn_cell_gemm(M, len(outs), arr, R, 0.0, 0, v, st), a.reps)
            line += ' v%d/%d %5.1f' % (v & 15, v >> 4, t)
        print(line, flush=True)


if __name__ == '__main__':
    main()

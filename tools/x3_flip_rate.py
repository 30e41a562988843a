"""How often does greedy decoding disagree with an fp64 decode of the same model -- for the exact-f32 path and for the
opt-in bf16x3 path (RFN_GEMM_OPT_BF16X3)?  (a RATE over thousands of rows, not another fixture.)

C2-sized model (M = 2, L = 49, D = 512, R = A = E = 512, V+1 = 9488, seeded uniform(+-0.1) weights), N random rows.  Three
greedy decodes of every row: the fp64 restatement (oracle/rfn_oracle.py in float64 on the host cores), the HIP path in exact
f32, the HIP path with the stage-I projections on bf16 planes.  Per mode: rows whose token ids differ anywhere from the fp64
ids, the step of first divergence, the fp64 top1-top2 margin at that step, and quantiles of max_v |log-prob - fp64 log-prob|
over all (row, step) pairs whose prefixes still agree.  One JSON object on stdout.

    python tools/x3_flip_rate.py [--workload c2|c3] [--rows 4096] [--chunk 512] > profiles/r04_x3_flip_rate.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench as HB  # noqa: E402
import recurrent_fusion_network_amd as R  # noqa: E402
import recurrent_fusion_network_amd._native as N  # noqa: E402
from oracle import rfn_oracle as O  # noqa: E402


def quantiles(x):
    x = x.double().flatten().sort().values
    pick = lambda q: float(x[min(x.numel() - 1, int(q * x.numel()))])  # noqa: E731
    return {'n': int(x.numel()), 'p50': pick(0.5), 'p90': pick(0.9), 'p99': pick(0.99), 'p999': pick(0.999), 'max': float(x[-1])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=4096)
    ap.add_argument('--chunk', type=int, default=512)
    ap.add_argument('--seed', type=int, default=31)
    ap.add_argument('--workload', default='c2', choices=['c2', 'c3'], help='model size: C2 (M=2, L=49, D=512) or C3 (M=4, L=196, D=2048)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    cfg = HB.make_cfg(HB.WORKLOADS[a.workload])
    P = O.seeded_params(cfg, a.seed)
    P64 = {k: v.double() for k, v in P.items()}
    model = R.RecurrentFusionModel(cfg)
    model.load_state_dict(P)
    model = model.to(dev).eval()
    modes = {'exact_f32': 0, 'bf16x3': N.GEMM_OPT_BF16X3 | N.GEMM_OPT_BF16X3_ANY_SIZE}
    S = cfg.seq_length
    stats = {m: dict(rows_differ=0, first_step=[], margin_at_flip=[], dlogp=[]) for m in modes}
    stats['x3_vs_exact'] = dict(rows_differ=0)
    margins_all = []
    t_cpu = 0.0
    for c0 in range(0, a.rows, a.chunk):
        nb = min(a.chunk, a.rows - c0)
        fc, att, _, _, _ = O.synthetic_batch(cfg, nb, seed=1000 + c0)
        t0 = time.perf_counter()
        with torch.no_grad():
            seq64, _, lp64, _ = O.sample_greedy(cfg, P64, [f.double() for f in fc], [x.double() for x in att])
        t_cpu += time.perf_counter() - t0
        T = lp64.size(1)
        ids64 = torch.zeros(nb, S, dtype=torch.long)
        ids64[:, :seq64.size(1)] = seq64
        top2 = lp64.topk(2, dim=2).values
        margin = top2[:, :, 0] - top2[:, :, 1]                       # (nb, T): fp64 top1 - top2 at every step
        margins_all.append(margin[:, :min(T, S)].flatten())
        got = {}
        for m, flags in modes.items():
            model.gemm_flags = flags
            with torch.no_grad():
                seq, _, lp, _ = model.sample([f.to(dev) for f in fc], [x.to(dev) for x in att], {'sample_max': 1})
            ids = torch.zeros(nb, S, dtype=torch.long)
            ids[:, :seq.size(1)] = seq.cpu()
            got[m] = ids
            lp = lp.cpu().double()
            Tm = min(T, lp.size(1))
            differ = ids != ids64                                                  # (nb, S)
            first = torch.where(differ.any(1), differ.float().argmax(1), torch.full((nb,), S))    # first differing token
            st = stats[m]
            st['rows_differ'] += int(differ.any(1).sum())
            for r in torch.nonzero(differ.any(1)).flatten().tolist():
                f = int(first[r])
                st['first_step'].append(f)
                st['margin_at_flip'].append(float(margin[r, f]) if f < T else None)
            # log-prob distance on every (row, step) whose inputs were still identical: step t is fed token t-1
            ok = torch.arange(Tm)[None, :] <= first[:, None]
            d = (lp[:, :Tm] - lp64[:, :Tm]).abs().amax(2)
            st['dlogp'].append(d[ok])
        stats['x3_vs_exact']['rows_differ'] += int((got['exact_f32'] != got['bf16x3']).any(1).sum())
        sys.stderr.write('rows %d..%d done (fp64 on the host: %.1f s so far)\n' % (c0, c0 + nb, t_cpu))
    w = HB.WORKLOADS[a.workload]
    out = {'model': '%s-sized RecurrentFusionModel (M=%d, L=%d, D=%d, R=A=E=512, T1=T2=8, V+1=9488), seeded uniform(+-0.1) weights' % (w['desc'], w['M'], w['L'], w['D']),
           'rows': a.rows, 'seq_length': S, 'reference': 'oracle/rfn_oracle.py in float64 on %d host threads, %.1f s' % (torch.get_num_threads(), t_cpu),
           'fp64_margin_top1_top2': quantiles(torch.cat(margins_all)),
           'fp64_margin_smallest': float(torch.cat(margins_all).min())}
    for m in modes:
        st = stats[m]
        out[m] = {'rows_with_ids_differing_from_fp64': st['rows_differ'], 'rate': st['rows_differ'] / a.rows,
                  'first_differing_step': sorted(st['first_step']), 'fp64_margin_at_that_step': st['margin_at_flip'],
                  'max_abs_dlogprob_vs_fp64_per_row_step': quantiles(torch.cat(st['dlogp']))}
    out['bf16x3_vs_exact_f32_rows_differing'] = stats['x3_vs_exact']['rows_differ']
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

"""Per-phase breakdown of one C3 train step from a `rocprofv3 --kernel-trace` CSV of bench.py.

Phases are delimited by marker kernels of the fixed launch schedule (csrc/rfn_path.hip): the grouped feature
projections, the stage-I forward recurrence, stage II + decoder forward + criterion, the logit-layer backward, decoder +
stage-II backward, the stage-I backward recurrence, the stage-I weight gradients, clamp+Adam.  Prints span, kernel time
and the heaviest kernels of each phase for the last complete step in the trace.

    python tools/trace_phases.py gpurun_out/prof_x/..._kernel_trace.csv [--top 6]
"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r'\(.*', '', name.replace('(anonymous namespace)::', '')).replace('void ', '')
    name = name.replace('rfn_gemm_kernel', 'gemm')
    return name[:64]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('csv')
    ap.add_argument('--top', type=int, default=6)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows]
    adam = [i for i, e in enumerate(ev) if e[2].startswith(('adam_k', 'adam_multi_k'))]
    groups, cur = [], [adam[0]]
    for x, y in zip(adam, adam[1:]):
        if y - x < 5:
            cur.append(y)
        else:
            groups.append(cur)
            cur = [y]
    groups.append(cur)
    if len(groups) < 3:
        raise SystemExit('need at least two complete steps in the trace')
    step = ev[groups[-3][-1] + 1:groups[-2][-1] + 1]
    t0 = step[0][0]

    def first(pat, start=0):
        for i in range(start, len(step)):
            if re.search(pat, step[i][2]):
                return i
        return len(step)

    def last(pat):
        for i in range(len(step) - 1, -1, -1):
            if re.search(pat, step[i][2]):
                return i
        return 0

    i_s1f = first(r'attn_scores_raw_k')
    # the h_2_att_h GEMM (+ reduce) of step 0 precedes the first score kernel: back up over the small kernels
    while i_s1f > 0 and step[i_s1f - 1][1] - step[i_s1f - 1][0] < 100000 and 'copyBuffer' not in step[i_s1f - 1][2]:
        i_s1f -= 1
    i_mid = first(r'mean_groups_k')
    i_lb = first(r'log_softmax_bwd_k')
    i_db = first(r'lstm_bwd_k', i_lb)
    i_s1b = first(r'attn_scores_bwd_k|attn_dalpha_k', i_db)
    while i_s1b > i_db and not re.search(r'lstm_bwd', step[i_s1b][2]):
        i_s1b -= 1
    i_wg = last(r'attn_scores_bwd_k|attn_dalpha_k') + 1
    while i_wg < len(step) and not re.search(r'fill_small_k', step[i_wg][2]):
        i_wg += 1
    i_wg += 1
    i_ad = first(r'adam_(multi_)?k')
    bounds = [('feature projections', 0, i_s1f), ('stage I forward recurrence', i_s1f, i_mid),
              ('stage II + decoder forward + criterion', i_mid, i_lb), ('logit layer backward', i_lb, i_db),
              ('decoder + stage II backward', i_db, i_s1b), ('stage I backward recurrence', i_s1b, i_wg),
              ('stage I weight gradients', i_wg, i_ad), ('clamp + Adam', i_ad, len(step))]
    total = (step[-1][1] - t0) / 1e6
    print('step: %d launches, %.2f ms' % (len(step), total))
    for name, lo, hi in bounds:
        seg = step[lo:hi]
        if not seg:
            continue
        span = (seg[-1][1] - seg[0][0]) / 1e6
        agg = collections.OrderedDict()
        for s, e, n in seg:
            x = agg.setdefault(n, [0, 0.0])
            x[0] += 1
            x[1] += (e - s) / 1e6
        busy = sum(v[1] for v in agg.values())
        print('== %-40s span %6.2f ms  kernel time %6.2f ms  %4d launches' % (name, span, busy, len(seg)))
        for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
            print('      %-64s n=%-4d %7.3f ms' % (n, c, t))


if __name__ == '__main__':
    main()

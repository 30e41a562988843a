"""Every launch of the kernels matching <substring> in the LAST complete train step of a `rocprofv3 --kernel-trace` CSV, with
the launches around it: start offset in the step, duration, gap to the previous launch's end, blocks.
    python tools/trace_window.py <kernel_trace.csv> <substring> [context=2] [min_us=0]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
pat, ctx = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 2
min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0


def short(n):
    return re.sub(r'\(.*', '', n.replace('(anonymous namespace)::', '')).replace('void ', '').replace('rfn_gemm_kernel', 'gemm')[:70]


ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']),
       int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) for r in rows]
adam = [i for i, e in enumerate(ev) if e[2].startswith(('adam_k', 'adam_multi_k'))]
step = ev[adam[-2] + 1:adam[-1] + 1] if len(adam) >= 2 else ev
t0 = step[0][0]
for i, e in enumerate(step):
    if pat not in e[2] or (e[1] - e[0]) / 1e3 < min_us:
        continue
    for j in range(max(0, i - ctx), min(len(step), i + ctx + 1)):
        s, t, n, b = step[j]
        gap = (s - step[j - 1][1]) / 1e3 if j > 0 else 0.0
        print('%s +%9.1f us  %9.1f us  gap %6.1f  %6d blk  %s' % ('>>' if j == i else '  ', (s - t0) / 1e3, (t - s) / 1e3, gap, b, n))
    print()

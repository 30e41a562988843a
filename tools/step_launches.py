"""One train step of a `rocprofv3 --kernel-trace` CSV of bench.py as a table: launches, span, kernel time, and per kernel
(with its grid size, so that the launches sharing an instantiation -- e.g. the 128x128 NT GEMM serving the hoisted
projection AND the stage-I gates -- are separate rows) count, total and mean duration.

    python tools/step_launches.py <kernel_trace.csv>
"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def short(n):
    return re.sub(r'\(.*', '', n.replace('(anonymous namespace)::', '')).replace('void ', '').replace('rfn_gemm_kernel', 'gemm')[:78]


ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']),
       int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) for r in rows]
adam = [i for i, e in enumerate(ev) if e[2].startswith(('adam_k', 'adam_multi_k'))]
groups, cur = [], [adam[0]]
for x, y in zip(adam, adam[1:]):
    if y - x < 5:
        cur.append(y)
    else:
        groups.append(cur)
        cur = [y]
groups.append(cur)
step = ev[groups[-3][-1] + 1:groups[-2][-1] + 1]
print('step: %d launches, span %.3f ms, kernel time %.3f ms, rfn_gemm_reduce_k launches: %d' % (
    len(step), (step[-1][1] - step[0][0]) / 1e6, sum(e[1] - e[0] for e in step) / 1e6,
    sum(1 for e in step if e[2].startswith('rfn_gemm_reduce_k'))))
agg = collections.OrderedDict()
for e in step:
    agg.setdefault((e[2], e[3]), []).append((e[1] - e[0]) / 1e3)
print('%-80s %8s %4s %10s %9s' % ('kernel', 'blocks', 'n', 'total us', 'mean us'))
for (name, blocks), ds in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print('%-80s %8d %4d %10.1f %9.1f' % (name, blocks, len(ds), sum(ds), sum(ds) / len(ds)))

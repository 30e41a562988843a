"""Mean device duration per (kernel, grid size) of a rocprofv3 --kernel-trace CSV, in launch order of first appearance."""
import collections
import csv
import re
import sys

agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(.*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '')).replace('void ', '')[:70]
    key = (name, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])))
    x = agg.setdefault(key, [0, 0.0])
    x[0] += 1
    x[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for (name, blocks), (n, t) in agg.items():
    if pat in name:
        print('%-72s blocks %6d  n %5d  mean %7.2f us' % (name, blocks, n, t / n))

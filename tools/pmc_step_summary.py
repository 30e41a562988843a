"""Per-kernel HBM-side traffic of a whole train step from the passes of tools/run_step_pmc.sh.
    python tools/pmc_step_summary.py gpurun_out/pmc_step/<tag> [substring ...]
One row per (kernel, grid): launches per step, mean duration under the counter pass, 2 x FETCH_SIZE (gfx950 tallies a
128-B request of a wide streaming read at 64 B: doubled, MI355X guide), WRITE_SIZE, their sum / duration, L2 hit rate,
clock (GRBM_GUI_ACTIVE / 8 XCDs / duration).  Counter units are KiB."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
wanted = sys.argv[2:] or ['attn_', 'adam', 'log_softmax', 'rfn_gemm_kernel<128', 'cell_gemm', 'colsum_grouped', 'embed']
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ('fetch', 'write', 'l2', 'clk'):
    for f in glob.glob(os.path.join(root, sub, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0]
            name = name.replace('void ', '')
            k = (name[:72], r.get('Grid_Size', ''))
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
            if 'Start_Timestamp' in r:
                agg[k]['_dur_' + sub].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
print('%-74s %9s %5s %9s %11s %10s %8s %7s %6s' % ('kernel', 'grid', 'n', 'us', '2xFETCH MB', 'WRITE MB', 'TB/s', 'L2 hit', 'GHz'))
rows = []
for k, c in agg.items():
    if not any(w in k[0] for w in wanted):
        continue
    avg = lambda n: (sum(c[n]) / len(c[n])) if c[n] else float('nan')  # noqa: E731
    dur = avg('_dur_fetch')
    fetch = 2 * avg('FETCH_SIZE') * 1024
    write = avg('WRITE_SIZE') * 1024
    hit = avg('TCC_HIT_sum') / (avg('TCC_HIT_sum') + avg('TCC_MISS_sum')) if c['TCC_HIT_sum'] else float('nan')
    clk = avg('GRBM_GUI_ACTIVE') / 8.0 / avg('_dur_clk') if c['GRBM_GUI_ACTIVE'] else float('nan')
    rows.append((-(fetch + write if fetch == fetch else 0) * len(c['FETCH_SIZE']), '%-74s %9s %5d %9.1f %11.1f %10.1f %8.2f %7.1f %6.2f' % (
        k[0], k[1], len(c['FETCH_SIZE']), dur / 1e3, fetch / 1e6, write / 1e6, (fetch + write) / dur / 1e3 if dur == dur else float('nan'),
        100 * hit, clk)))
for _, line in sorted(rows):
    print(line)

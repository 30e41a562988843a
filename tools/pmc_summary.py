"""Summarises the rocprofv3 --pmc passes of tools/run_gemm_pmc.sh: per kernel (big GEMM launches only) the average
duration, effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), matrix-pipe occupancy
(SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)), wait fractions, and the TCC byte counters
(FETCH_SIZE reports 1/2 of the bytes on gfx950: doubled here; KiB units)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
wanted = sys.argv[2] if len(sys.argv) > 2 else 'rfn_gemm_kernel'     # kernel-name substring


def load(sub):
    rows = []
    for f in glob.glob(os.path.join(root, sub, '**', '*counter_collection.csv'), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def key(r):
    return (r['Kernel_Name'][:60], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X', ''))


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ('sq', 'fetch', 'write', 'l2'):
    for r in load(sub):
        k = (r['Kernel_Name'].split('(')[0][-70:], r.get('Grid_Size', ''))
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        if 'Start_Timestamp' in r and r['Counter_Name'] in ('GRBM_GUI_ACTIVE',):
            agg[k]['_dur'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
for k, c in agg.items():
    if 'GRBM_GUI_ACTIVE' not in c or wanted not in k[0]:
        continue
    avg = lambda n: sum(c[n]) / max(1, len(c[n]))  # noqa: E731
    gui = avg('GRBM_GUI_ACTIVE') / 8.0
    if gui < 1e6:
        continue
    line = '%s grid=%s n=%d | cycles/XCD %.2fM' % (k[0][-48:], k[1], len(c['GRBM_GUI_ACTIVE']), gui / 1e6)
    if c['_dur']:
        dur = avg('_dur')
        line += ' | %.3f ms | clock %.3f GHz' % (dur / 1e6, gui / dur)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        line += ' | MFMA busy %.1f %%' % (100.0 * avg('SQ_VALU_MFMA_BUSY_CYCLES') / (gui * 1024))
    if 'SQ_WAVE_CYCLES' in c:
        wc = avg('SQ_WAVE_CYCLES')
        line += ' | WAIT_ANY %.1f %% WAIT_INST %.1f %%' % (100 * avg('SQ_WAIT_ANY') / wc, 100 * avg('SQ_WAIT_INST_ANY') / wc)
    if 'SQ_LDS_BANK_CONFLICT' in c:
        line += ' | LDS conflict cycles %.0f of %.3g' % (avg('SQ_LDS_BANK_CONFLICT'), avg('SQ_LDS_IDX_ACTIVE'))
    if 'FETCH_SIZE' in c:
        line += ' | fetch %.2f GB (2 x FETCH_SIZE)' % (2 * avg('FETCH_SIZE') * 1024 / 1e9)
    if 'WRITE_SIZE' in c:
        line += ' | write %.2f GB' % (avg('WRITE_SIZE') * 1024 / 1e9)
    if 'TCC_HIT_sum' in c:
        line += ' | L2 hit %.1f %%' % (100 * avg('TCC_HIT_sum') / (avg('TCC_HIT_sum') + avg('TCC_MISS_sum')))
    print(line)

"""Every launch of the step's dominant GEMM kernels in a `rocprofv3 --kernel-trace` CSV of bench.py, with its duration:
the four projection / four weight-gradient launches per step (exact-f32: rfn_gemm_kernel at grids 12800 x 256 and 512 x 256
blocks x threads; bf16x3: x3_gemm_k at 3264 and 256 blocks of 512 threads).   python tools/dominant_launches.py trace.csv"""
import csv
import sys

FLOP = 2.0 * 256 * 196 * 2048 * 512 * 8          # one encoder: f32 multiply-adds of the product
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print('kernel,launch,duration_ms,TFLOP_per_s_f32_equivalent')
count = {}
for r in rows:
    name = r['Kernel_Name']
    blocks = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])) if 'Grid_Size_X' in r else 0
    zs = int(r.get('Grid_Size_Z', 1) or 1)
    label = None
    if 'rfn_gemm_kernel' in name and blocks == 12800:
        label = 'exact NT projection (12800 blocks)'
    elif 'rfn_gemm_kernel' in name and blocks == 512 and '128, 128, false, false' in name:
        label = 'exact TN weight gradient (512 blocks)'
    elif 'x3_gemm_k' in name and zs == 1 and blocks > 3000:
        label = 'bf16x3 NT projection (%d blocks)' % blocks
    elif 'x3_gemm_k' in name and zs > 1:
        label = 'bf16x3 TN weight gradient (%d blocks x %d K slices)' % (blocks, zs)
    ms = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    if label is None or ms < 2.0:      # other launches share the grids; the dominant ones take 3-6 ms
        continue
    count[label] = count.get(label, 0) + 1
    print('%s,%d,%.4f,%.1f' % (label, count[label], ms, FLOP / ms / 1e9))

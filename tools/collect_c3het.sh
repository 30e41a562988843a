#!/bin/bash
# c3het (the reference's shipped five encoders) and its permutation with the 196 x 2048 map last: launch table, phase table and
# the neighbourhood of the 196 x 2048 projection launch (grid 12800) in the last traced step.   bash tools/collect_c3het.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/c3het
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in c3het c3het_last c3het; do
  tag=$wl; [ -e $O/${tag}_phases.txt ] && tag=${wl}_again
  rocprofv3 --kernel-trace --stats -d $O/$tag -o t --output-format csv -- python3 $R/bench.py --workload $wl --no-cpu-baseline --no-alt-line --steps 6 --settle 1 > $O/$tag.log 2>&1
  python3 $R/tools/step_launches.py $O/$tag/t_kernel_trace.csv > $O/${tag}_step_launches.txt 2>&1
  python3 $R/tools/trace_phases.py $O/$tag/t_kernel_trace.csv --top 12 > $O/${tag}_phases.txt 2>&1
  python3 $R/tools/trace_window.py $O/$tag/t_kernel_trace.csv "gemm<128, 128, true, true, true, 1, 32, true, 256, true, 2>" 3 1000 > $O/${tag}_proj_window.txt 2>&1
  python3 $R/tools/trace_window.py $O/$tag/t_kernel_trace.csv "gemm<128, 128, false, false, true, 1, 16, true, 256, false, 3>" 2 1000 > $O/${tag}_wgrad_window.txt 2>&1
  tail -1 $O/$tag.log | cut -c1-220
  rm -rf $O/$tag
done

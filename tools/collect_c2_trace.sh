#!/bin/bash
# One-step launch table + phase table of a small workload (default --workload c2) from a kernel trace.
#   bash tools/collect_c2_trace.sh <tag> [bench args...]      -> gpurun_out/c2trace/<tag>_{step_launches,phases}.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r06}; shift
ARGS=${@:---workload c2}
O=$R/gpurun_out/c2trace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/$TAG -o t --output-format csv -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-alt-line --steps 6 --settle 1 > $O/$TAG.log 2>&1
python3 $R/tools/step_launches.py $O/$TAG/t_kernel_trace.csv > $O/${TAG}_step_launches.txt 2>&1
python3 $R/tools/trace_phases.py $O/$TAG/t_kernel_trace.csv --top 12 > $O/${TAG}_phases.txt 2>&1
cp $O/$TAG/t_kernel_stats.csv $O/${TAG}_kernel_stats.csv 2>/dev/null
rm -rf $O/$TAG

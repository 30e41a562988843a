#!/bin/bash
# PMC passes over the WHOLE train step as bench.py runs it (the HBM-bound kernels in their current
# grouped form, counted where they run -- behind their real neighbours).  Separate passes per the MI355X guide:
# FETCH_SIZE, WRITE_SIZE (the TCC slots do not hold both), L2 hit / miss, clock.  Usage:
#     bash tools/run_step_pmc.sh <tag> [bench.py args ...]     e.g.  run_step_pmc.sh b256 ; run_step_pmc.sh b32 --batch 32
# Outputs gpurun_out/pmc_step/<tag>/{fetch,write,l2,clk}; summary: python tools/pmc_step_summary.py gpurun_out/pmc_step/<tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-b256}; shift
O=$R/gpurun_out/pmc_step/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-alt-line --steps 3 --warmup 2 --settle 0 $*"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o t -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o t -- python3 $R/bench.py $ARGS > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2 -o t -- python3 $R/bench.py $ARGS > $O/l2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/clk -o t -- python3 $R/bench.py $ARGS > $O/clk.log 2>&1
cd $R
python3 tools/pmc_step_summary.py $O > $O/summary.txt 2>&1
# the raw per-dispatch tables are tens of MB: keep the summary, drop the rest
find $O -name "*.csv" -size +2M -delete
cat $O/summary.txt

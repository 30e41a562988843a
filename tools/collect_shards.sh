#!/bin/bash
# Kernel traces of the small-shard regime (what strong scaling and BASELINE config 2 run): one-step launch tables and phase
# tables at B = 32 / 64 per GPU and for --workload c2.  Outputs under gpurun_out/shards/<tag>/ (copy into profiles/).
#   bash tools/collect_shards.sh [tag]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
O=$R/gpurun_out/shards/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in "b32:--batch 32" "b64:--batch 64" "b128:--batch 128" "c2:--workload c2"; do
  name=${spec%%:*}; args=${spec#*:}
  rocprofv3 --kernel-trace --stats -d $O/$name -o t --output-format csv -- python3 $R/bench.py $args --no-cpu-baseline --no-alt-line --steps 6 --settle 1 > $O/$name.log 2>&1
  python3 $R/tools/step_launches.py $O/$name/t_kernel_trace.csv > $O/${name}_step_launches.txt 2>&1
  python3 $R/tools/trace_phases.py $O/$name/t_kernel_trace.csv --top 10 > $O/${name}_phases.txt 2>&1
  cp $O/$name/t_kernel_stats.csv $O/${name}_kernel_stats.csv 2>/dev/null
  rm -rf $O/$name
done
cd $R
: > $O/lines.jsonl
for args in "--batch 32" "--batch 64" "--batch 128" "--workload c2" ""; do
  python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 >> $O/lines.jsonl
done
ls -la $O

#!/bin/bash
# Evidence for the persistent recurrence kernels (profiles/r05_chain.md): kernel traces of `bench.py --workload c2` with the
# per-step launches (default) and with every recurrence inside one persistent launch (--persist 15), their launch tables,
# and alternating graph-replayed bench lines at C2 / B = 32 / C3 (the replay sits at the device time whatever the host does).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/chain
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for tag in 0 15; do
  rocprofv3 --kernel-trace --stats -d $O/c2_p$tag -o t --output-format csv -- python3 $R/bench.py --workload c2 --no-cpu-baseline --no-alt-line --steps 5 --persist $tag > $O/c2_p$tag.log 2>&1
  python3 $R/tools/step_launches.py $O/c2_p$tag/t_kernel_trace.csv > $O/c2_p${tag}_step_launches.txt
  rm -f $O/c2_p$tag/t_kernel_trace.csv
done
cd $R
: > $O/ab.txt
for wl in "--workload c2" "--batch 32" "--batch 64" ""; do
  for a in 0 15 16 0 15 16; do
    python bench.py --no-alt-line --no-cpu-baseline $wl --graph --persist $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', 'flags', '$a', d['ms_per_step'])" >> $O/ab.txt
  done
done
cat $O/ab.txt

#!/bin/bash
# Same-box A/B of library builds (tools/build_variant.sh): alternates the product library and the named variants over
# graph-replayed small workloads.   bash tools/ab_lib.sh "<tag> <tag> ..." [rounds]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAGS=$1; N=${2:-2}
cd $R
for r in $(seq $N); do
  for t in product $TAGS; do
    if [ $t = product ]; then unset RFN_HIP_LIB; else export RFN_HIP_LIB=$R/recurrent_fusion_network_amd/librfn_hip_$t.so; fi
    for spec in "c2:--workload c2 --graph" "b32:--batch 32 --graph" "b64:--batch 64 --graph" "c3:"; do
      name=${spec%%:*}; args=${spec#*:}
      ms=$(python bench.py --no-cpu-baseline --no-alt-line --settle 1 $args 2>/dev/null | tail -1 | python3 -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')
      echo "$t $name $ms"
    done
  done
done

#!/bin/bash
# Same-box A/B of the decoder cell's two forms (round 6): hoisted z2h (default, two launches per step each way) against the
# three-launch form of rounds 3-5 (--persist 128 = RFN_PATH_OPT_DEC_UNHOISTED), alternating, graph-replayed where the
# launch path would pace the step.  Output: gpurun_out/ab_hoist/<tag>/lines.jsonl (one bench line per run, tagged).
#   bash tools/ab_hoist.sh [tag] [rounds]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r06}
N=${2:-2}
O=$R/gpurun_out/ab_hoist/$TAG
mkdir -p $O
cd $R
: > $O/lines.jsonl
run() {   # name, args...
  name=$1; shift
  line=$(python bench.py --no-cpu-baseline --no-alt-line --settle 1 "$@" 2>$O/err.log | tail -1)
  echo "{\"tag\": \"$name\", \"line\": $line}" >> $O/lines.jsonl
  echo "$name $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d.get("ms_per_step"), d.get("modes", ""))')"
}
for r in $(seq $N); do
  for form in "hoisted:0" "unhoisted:128"; do
    f=${form%%:*}; p=${form#*:}
    run c2_graph_$f --workload c2 --graph --persist $p
    run c2_eager_$f --workload c2 --persist $p
    run b32_graph_$f --batch 32 --graph --persist $p
    run b64_graph_$f --batch 64 --graph --persist $p
    run b128_$f --batch 128 --persist $p
    run c3_$f --persist $p
    run c5_$f --workload c5 --persist $p
  done
done

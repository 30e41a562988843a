#!/bin/bash
# Round evidence in one GPU-box call: the bench line, rocprofv3 kernel-trace summaries of both GEMM modes, phase tables,
# dominant launches, the secondary lines, the PMC passes of the dominant GEMM.  Outputs under gpurun_out/ev/ (copy what is
# to be judged into profiles/).   bash tools/collect_evidence.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ev
mkdir -p $O
cd $R
python bench.py --no-cpu-baseline > /dev/null 2>&1          # warm the box (first run after a fresh start reads slow)
python bench.py 2> $O/bench.err | tail -1 > $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/exact -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-alt-line > $O/exact.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/x3 -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --gemm bf16x3 > $O/x3.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/c2 -o t --output-format csv -- python3 $R/bench.py --workload c2 --no-cpu-baseline --no-alt-line > $O/c2.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/dec -o t --output-format csv -- python3 $R/tools/prof_decode.py > $O/dec.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/c3het -o t --output-format csv -- python3 $R/bench.py --workload c3het --no-cpu-baseline --no-alt-line > $O/c3het.log 2>&1
cd $R
python tools/trace_phases.py $O/exact/t_kernel_trace.csv --top 8 > $O/phases_exact.txt
python tools/trace_phases.py $O/x3/t_kernel_trace.csv --top 8 > $O/phases_x3.txt
python tools/dominant_launches.py $O/exact/t_kernel_trace.csv > $O/dominant_exact.csv
python tools/dominant_launches.py $O/x3/t_kernel_trace.csv > $O/dominant_x3.csv
python tools/step_launches.py $O/exact/t_kernel_trace.csv > $O/step_launches_exact.txt
python tools/step_launches.py $O/c2/t_kernel_trace.csv > $O/step_launches_c2.txt
: > $O/secondary.jsonl
for args in "--workload c2" "--workload c2 --graph --no-alt-line" "--workload c3het" "--recipe" "--label-smoothing" "--workload c5"; do
  python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 >> $O/secondary.jsonl
done
python bench.py --no-cpu-baseline --workload c5 --gemm bf16x3 2>/dev/null | tail -1 >> $O/secondary.jsonl
bash tools/run_gemm_pmc.sh r06 "" all > $O/pmc_gemm.txt 2>&1
: > $O/shards.jsonl      # single-GPU steps of data-parallel shards (tools/dp_predict.py)
for b in 256 128 64 32; do
  python bench.py --no-cpu-baseline --batch $b 2>/dev/null | tail -1 >> $O/shards.jsonl
done
python tools/step_launches.py $O/c3het/t_kernel_trace.csv > $O/step_launches_c3het.txt 2>/dev/null
rm -f $O/exact/t_kernel_trace.csv $O/x3/t_kernel_trace.csv $O/c2/t_kernel_trace.csv $O/dec/t_kernel_trace.csv $O/c3het/t_kernel_trace.csv    # tens of MB; the summaries stay
ls -la $O

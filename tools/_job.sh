cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final2
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/final2/r02_bench.json 2> gpurun_out/final2/bench.err; cut -c1-330 gpurun_out/final2/r02_bench.json
rocprofv3 --kernel-trace --stats -d gpurun_out/final2/prof -o r02 --output-format csv -- python3 bench.py --no-cpu-baseline > gpurun_out/final2/prof.log 2>&1
python tools/trace_phases.py gpurun_out/final2/prof/r02_kernel_trace.csv --top 6 > gpurun_out/final2/r02_phases.txt; grep "==" gpurun_out/final2/r02_phases.txt
rm -f gpurun_out/final2/secondary.jsonl
for args in "--workload c2" "--workload c3het" "--recipe" "--label-smoothing" "--workload c5 --steps 5 --warmup 2"; do
  python bench.py $args >> gpurun_out/final2/secondary.jsonl 2>> gpurun_out/final2/secondary.err
done
RFN_DIST_BACKEND=gloo RFN_DEVICE_INDEX=0 python bench.py --gpus 2 --workload c2 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline >> gpurun_out/final2/secondary.jsonl 2>> gpurun_out/final2/secondary.err
python - <<PY
import json
for l in open("gpurun_out/final2/secondary.jsonl"):
    d=json.loads(l); print(d["n_gpus"], d["ms_per_step"], d["value"], d["config"]["workload"][:50], d.get("modes"))
PY

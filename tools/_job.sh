cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
python bench.py > gpurun_out/final/r02_bench.json 2> gpurun_out/final/bench.err; cat gpurun_out/final/r02_bench.json
rocprofv3 --kernel-trace --stats -d gpurun_out/final/prof -o r02 --output-format csv -- python3 bench.py --no-cpu-baseline > gpurun_out/final/prof.log 2>&1
python tools/trace_phases.py gpurun_out/final/prof/r02_kernel_trace.csv --top 6 > gpurun_out/final/r02_phases.txt; head -3 gpurun_out/final/r02_phases.txt
bash tools/run_gemm_pmc.sh r02f "" all > gpurun_out/final/pmc.log 2>&1; cat gpurun_out/final/pmc.log
rm -f gpurun_out/final/secondary.jsonl
for args in "--workload c2" "--workload c3het" "--recipe" "--label-smoothing" "--workload c5 --steps 5 --warmup 2"; do
  python bench.py $args >> gpurun_out/final/secondary.jsonl 2>> gpurun_out/final/secondary.err
done
RFN_DIST_BACKEND=gloo RFN_DEVICE_INDEX=0 python bench.py --gpus 2 --workload c2 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline >> gpurun_out/final/secondary.jsonl 2>> gpurun_out/final/secondary.err
cut -c1-330 gpurun_out/final/secondary.jsonl

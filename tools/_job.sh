cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/run_gemm_pmc.sh r02 "" all > gpurun_out/pmc_r02.log 2>&1
mkdir -p gpurun_out/lines
for args in "--workload c2" "--workload c3het" "--recipe" "--label-smoothing" "--workload c5 --steps 5 --warmup 2"; do
  python bench.py $args --no-cpu-baseline >> gpurun_out/lines/secondary.jsonl 2>> gpurun_out/lines/secondary.err
done
cat gpurun_out/pmc_r02.log; cat gpurun_out/lines/secondary.jsonl; tail -5 gpurun_out/lines/secondary.err

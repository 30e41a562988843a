#!/bin/bash
# PMC passes over the stage-I attention kernels alone (tools/bench_attn.py, C3 shapes).  Separate passes per the
# MI355X guide (FETCH_SIZE and WRITE_SIZE do not fit the TCC slots together).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_attn
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_attn/fetch -- python3 tools/bench_attn.py --reps 4 > gpurun_out/pmc_attn/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_attn/write -- python3 tools/bench_attn.py --reps 4 > gpurun_out/pmc_attn/write.log 2>&1
find gpurun_out/pmc_attn -name "*counter_collection.csv" | head; tail -2 gpurun_out/pmc_attn/fetch.log

#!/bin/bash
# PMC passes over the standalone GEMM bench (product kernel, default knobs).  Separate passes per the
# MI355X guide: SQ counters; FETCH_SIZE; WRITE_SIZE (TCC slots do not fit both).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p /tmp/gb gpurun_out/pmc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result tools/gemm_bench.hip -o /tmp/gb/default || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc/sq -- /tmp/gb/default > gpurun_out/pmc/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc/fetch -- /tmp/gb/default > gpurun_out/pmc/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc/write -- /tmp/gb/default > gpurun_out/pmc/write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc/l2 -- /tmp/gb/default > gpurun_out/pmc/l2.log 2>&1
find gpurun_out/pmc -name "*.csv" | head -20; tail -3 gpurun_out/pmc/sq.log

#!/bin/bash
# PMC passes over the standalone GEMM bench.  Separate passes per the MI355X guide: SQ counters; FETCH_SIZE;
# WRITE_SIZE (TCC slots do not fit both).  Usage: run_gemm_pmc.sh <variant-name> "<-D flags>" [sq|all]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
V=${1:-default}; FLAGS=${2:-}; WHAT=${3:-all}
mkdir -p /tmp/gb gpurun_out/pmc/$V
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -DVARIANT="\"$V\"" $FLAGS tools/gemm_bench.hip -o /tmp/gb/pmc_$V || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc/$V/sq -- /tmp/gb/pmc_$V > gpurun_out/pmc/$V/sq.log 2>&1
if [ "$WHAT" = all ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc/$V/fetch -- /tmp/gb/pmc_$V > gpurun_out/pmc/$V/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc/$V/write -- /tmp/gb/pmc_$V > gpurun_out/pmc/$V/write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc/$V/l2 -- /tmp/gb/pmc_$V > gpurun_out/pmc/$V/l2.log 2>&1
fi
python3 tools/pmc_summary.py gpurun_out/pmc/$V

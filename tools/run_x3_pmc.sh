#!/bin/bash
# PMC passes over the standalone bf16-plane GEMM bench (tools/x3_gemm_bench.hip), separate passes per the MI355X guide.
# Usage: run_x3_pmc.sh <variant-name> "<-D flags>" [sq|all]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
V=${1:-default}; FLAGS=${2:-}; WHAT=${3:-all}
mkdir -p /tmp/x3 gpurun_out/pmc_x3/$V
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -DVARIANT="\"$V\"" $FLAGS tools/x3_gemm_bench.hip -o /tmp/x3/pmc_$V || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_x3/$V/sq -- /tmp/x3/pmc_$V > gpurun_out/pmc_x3/$V/sq.log 2>&1
if [ "$WHAT" = all ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_x3/$V/fetch -- /tmp/x3/pmc_$V > gpurun_out/pmc_x3/$V/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_x3/$V/write -- /tmp/x3/pmc_$V > gpurun_out/pmc_x3/$V/write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_x3/$V/l2 -- /tmp/x3/pmc_$V > gpurun_out/pmc_x3/$V/l2.log 2>&1
fi
python3 tools/pmc_summary.py gpurun_out/pmc_x3/$V x3_gemm_k

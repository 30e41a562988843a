#!/bin/bash
# builds + runs the GEMM A/B variants (run on the GPU box; hipcc is available there too)
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/gb
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -DVARIANT="\"$1\"" $2 tools/gemm_bench.hip -o /tmp/gb/$1 & }
build bk32_s2 ""
build bk16_s3 "-DGEMM_DMA_SLOTS_NT=3 -DGEMM_DMA_SLOTS_XX=3 -DGEMM_DMA_BK=16"
build bk16_s2_w4 "-DGEMM_DMA_BK=16 -DGEMM_DMA_MIN_WAVES=4"
build bk16_s4 "-DGEMM_DMA_SLOTS_NT=4 -DGEMM_DMA_SLOTS_XX=4 -DGEMM_DMA_BK=16"
build bk32_s2_w2 "-DGEMM_DMA_MIN_WAVES=2"
wait
for r in 1 2 3; do for v in bk32_s2 bk16_s3 bk16_s2_w4 bk16_s4 bk32_s2_w2; do /tmp/gb/$v | grep -v "reg-staged"; done; done

#!/bin/bash
# builds + runs the GEMM A/B variants (run on the GPU box; hipcc is available there too)
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/gb
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -DVARIANT="\"$1\"" $2 tools/gemm_bench.hip -o /tmp/gb/$1 & }
build product ""
build bk16_s3 "-DGEMM_DMA_SLOTS_NT=3 -DGEMM_DMA_SLOTS_XX=3 -DGEMM_DMA_BK=16"
wait
for r in 1 2 3; do for v in product bk16_s3; do /tmp/gb/$v; done; done

#!/bin/bash
# builds + runs the GEMM ablation variants (run on the GPU box; hipcc is available there too)
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/gb
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -DVARIANT="\"$1\"" $2 tools/gemm_bench.hip -o /tmp/gb/$1; }
build default ""
build t128x64_s2 "-DGEMM_BIG_BN=64 -DGEMM_NT_STAGES=2 -DGEMM_MIN_WAVES=4"
build t128x64_s1 "-DGEMM_BIG_BN=64 -DGEMM_NT_STAGES=1 -DGEMM_XX_STAGES=1 -DGEMM_MIN_WAVES=4"
build t64x128_s2 "-DGEMM_BIG_BM=64 -DGEMM_NT_STAGES=2 -DGEMM_MIN_WAVES=4"
build t256x128_s1 "-DGEMM_BIG_BM=256 -DGEMM_NT_STAGES=1 -DGEMM_XX_STAGES=1 -DGEMM_MIN_WAVES=1"
for v in default t128x64_s2 t128x64_s1 t64x128_s2 t256x128_s1; do /tmp/gb/$v; done

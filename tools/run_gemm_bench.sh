#!/bin/bash
# builds + runs the GEMM ablation variants (run on the GPU box; hipcc is available there too)
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/gb
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -DVARIANT="\"$1\"" $2 tools/gemm_bench.hip -o /tmp/gb/$1; }
build default ""
build single_w3 "-DGEMM_LDS_STAGES=1 -DGEMM_MIN_WAVES=3"
build single_w4 "-DGEMM_LDS_STAGES=1 -DGEMM_MIN_WAVES=4"
build single_w2 "-DGEMM_LDS_STAGES=1 -DGEMM_MIN_WAVES=2"
for v in default single_w3 single_w4 single_w2; do /tmp/gb/$v; done

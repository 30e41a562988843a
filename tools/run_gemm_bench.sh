#!/bin/bash
# builds + runs the GEMM ablation variants (run on the GPU box; hipcc is available there too)
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/gb
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -DVARIANT="\"$1\"" $2 tools/gemm_bench.hip -o /tmp/gb/$1; }
build default ""
build band4 "-DGEMM_BAND_ROWS=4"
build band16 "-DGEMM_BAND_ROWS=16"
build band32 "-DGEMM_BAND_ROWS=32"
build band64 "-DGEMM_BAND_ROWS=64"
for v in default band4 band16 band32 band64; do /tmp/gb/$v; done

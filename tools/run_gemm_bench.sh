#!/bin/bash
# builds + runs the GEMM ablation variants (run on the GPU box; hipcc is available there too)
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/gb
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -DVARIANT="\"$1\"" $2 tools/gemm_bench.hip -o /tmp/gb/$1; }
build default ""
build band4 "-DGEMM_BAND_ROWS=4"
build band12 "-DGEMM_BAND_ROWS=12"
build band16 "-DGEMM_BAND_ROWS=16"
build band24 "-DGEMM_BAND_ROWS=24"
for r in 1 2; do for v in default band4 band12 band16 band24; do /tmp/gb/$v | grep NT; done; done

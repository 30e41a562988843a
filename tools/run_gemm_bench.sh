#!/bin/bash
# builds + runs the GEMM ablation variants (run on the GPU box; hipcc is available there too)
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/gb
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -DVARIANT="\"$1\"" $2 tools/gemm_bench.hip -o /tmp/gb/$1; }
build default ""
build onewave_s1 "-DGEMM_ONE_WAVE=1 -DGEMM_ONE_WAVE_STAGES=1"
build onewave_s2 "-DGEMM_ONE_WAVE=1 -DGEMM_ONE_WAVE_STAGES=2"
for v in default onewave_s1 onewave_s2; do /tmp/gb/$v; done

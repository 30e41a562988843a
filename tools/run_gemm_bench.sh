#!/bin/bash
# builds + runs the GEMM ablation variants (run on the GPU box; hipcc is available there too)
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/gb
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -DVARIANT="\"$1\"" $2 tools/gemm_bench.hip -o /tmp/gb/$1; }
build default ""
build xx1 "-DGEMM_XX_STAGES=1"
build nt2 "-DGEMM_NT_STAGES=2"
build xx1_w3 "-DGEMM_XX_STAGES=1 -DGEMM_MIN_WAVES=3"
build fragpipe "-DGEMM_FRAG_PIPE=1"
for v in default xx1 nt2 xx1_w3 fragpipe; do /tmp/gb/$v; done

#!/bin/bash
# builds + runs the GEMM ablation variants (run on the GPU box; hipcc is available there too)
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/gb
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -DVARIANT="\"$1\"" $2 tools/gemm_bench.hip -o /tmp/gb/$1; }
build default ""
build fragpipe "-DGEMM_FRAG_PIPE=1"
build fragpipe_s2 "-DGEMM_FRAG_PIPE=1 -DGEMM_NT_STAGES=2"
for v in default fragpipe fragpipe_s2 default; do /tmp/gb/$v; done

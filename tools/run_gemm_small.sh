#!/bin/bash
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/gb
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -DVARIANT="\"$1\"" $2 tools/gemm_small_bench.hip -o /tmp/gb/s_$1; }
build default ""
build nomedium "-DGEMM_MEDIUM_MIN_FLOPS=1e30"
build medium2g "-DGEMM_MEDIUM_MIN_FLOPS=2e9"
for v in nomedium default medium2g; do /tmp/gb/s_$v; echo; done

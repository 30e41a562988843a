#!/bin/bash
# builds + runs the GEMM A/B variants (run on the GPU box; hipcc is available there too):  run_gemm_bench.sh "name|flags" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/gb
[ $# -eq 0 ] && set -- "product|" "bk16_s3|-DGEMM_DMA_SLOTS_NT=3 -DGEMM_DMA_SLOTS_XX=3 -DGEMM_DMA_BK=16"
for v in "$@"; do
  name="${v%%|*}"; flags="${v#*|}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -DVARIANT="\"$name\"" $flags tools/gemm_bench.hip -o /tmp/gb/$name &
done
wait
for r in 1 2; do for v in "$@"; do /tmp/gb/${v%%|*}; done; done

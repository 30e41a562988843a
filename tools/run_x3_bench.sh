#!/bin/bash
# builds + runs tools/x3_gemm_bench.hip variants on the GPU box:  run_x3_bench.sh "name|flags" ...
cd "$(dirname "$0")/.."
mkdir -p /tmp/x3
for v in "$@"; do
  name="${v%%|*}"; flags="${v#*|}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -DVARIANT="\"$name\"" $flags tools/x3_gemm_bench.hip -o /tmp/x3/$name 2>/dev/null &
done
wait
for v in "$@"; do name="${v%%|*}"; timeout 300 /tmp/x3/$name; done

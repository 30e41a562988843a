#!/bin/bash
# Out-of-tree build of librfn_hip with extra -D knobs into recurrent_fusion_network_amd/librfn_hip_<tag>.so (A/B builds; load
# it with RFN_HIP_LIB=<path>).   bash tools/build_variant.sh <tag> "<-D flags>"
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; FLAGS=$2
O=/tmp/rfn_variant_$TAG
mkdir -p $O
cd $R/recurrent_fusion_network_amd/csrc
RT=$(python3 -c "import os, torch; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
pids=()
for f in rfn_gemm rfn_gemm_x3 rfn_cellgemm rfn_chain rfn_attn rfn_deccell rfn_cell rfn_misc rfn_beam rfn_path; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result $FLAGS -c $f.hip -o $O/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -no-hip-rt -o ../librfn_hip_$TAG.so $O/*.o -L$RT -lamdhip64 -Wl,-rpath,$RT
ls -la ../librfn_hip_$TAG.so

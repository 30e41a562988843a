cd $GRAFT_REPO_ROOT
for v in "-DATT_NT_X=1" "-DATT_NT_X=1 -DATT_NT_P=1" "-DATT_NT_X=1" "-DATT_NT_X=1 -DATT_NT_P=1"; do
  touch recurrent_fusion_network_amd/csrc/rfn_attn.hip
  make -C recurrent_fusion_network_amd/csrc EXTRA="$v" -j8 > /dev/null 2>&1
  echo "== $v"; python tools/bench_attn.py --contig 2>&1 | grep -E "us "
done

"""Records a re-measured roofline.traffic figure in profiles/pmc_traffic.json together with the hash of the kernel
source it was measured at (bench.py nulls the field when the hash no longer matches).

    python tools/stamp_pmc_traffic.py c3 4.34 "note"            # exact-f32 projection GEMM (rfn_gemm.hip)
    python tools/stamp_pmc_traffic.py c3_bf16x3 6.53 "note"     # plane GEMM (rfn_gemm_x3.hip)
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
key, gb = sys.argv[1], float(sys.argv[2])
note = sys.argv[3] if len(sys.argv) > 3 else ''
src = 'recurrent_fusion_network_amd/csrc/' + ('rfn_gemm_x3.hip' if key.endswith('_bf16x3') else 'rfn_gemm.hip')
path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
data = json.load(open(path))
sha = hashlib.sha256(open(os.path.join(ROOT, src), 'rb').read()).hexdigest()[:16]
data[key] = {'gb': gb, 'src': src, 'src_sha16': sha, 'note': note or data.get(key, {}).get('note', '')}
json.dump(data, open(path, 'w'), indent=1)
print(key, data[key])

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as HB
import recurrent_fusion_network_amd as R
dev = torch.device('cuda:0')
w = dict(HB.WORKLOADS['c3']); B = 128
cfg = HB.make_cfg(w)
model = R.RecurrentFusionModel(cfg).to(dev)
HB.seeded_weights_(model, 100)
fc, att, labels, masks, top = HB.synthetic_inputs(cfg, B, 100, dev)
model.eval()
with torch.no_grad():
    for _ in range(3):
        model.sample(fc, att, {'beam_size': 5})
    torch.cuda.synchronize()
    for _ in range(3):
        model.sample(fc, att, {'sample_max': 1})
    torch.cuda.synchronize()

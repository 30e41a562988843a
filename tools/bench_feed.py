#!/usr/bin/env python3
"""PCIe-inclusive XE train step at the headline model size (M=4, L=196, D=2048): 51 images x 5 captions = 255 caption
rows per step, features start in HOST memory every step.  Three ways to feed the same step:
  replicated : the reference's way -- the host batch holds every image 5x (train.py:116-133), synchronous H2D;
  feeder     : unique images, pinned double-buffered async H2D (feeder.py), expanded to caption rows on the device;
  feeder+dedup: as above, and stages I/II run once per image (model.dedup_seq_per_img = 5).
Not the headline metric (bench.py keeps inputs resident); reported in DESIGN.md section 8."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench as HB
import recurrent_fusion_network_amd as R
from recurrent_fusion_network_amd.feeder import FeatureFeeder

dev = torch.device('cuda:0')
w = dict(HB.WORKLOADS['c3']); n_img, spi = 51, 5; B = n_img * spi
cfg = HB.make_cfg(w)
model = R.RecurrentFusionModel(cfg).to(dev); HB.seeded_weights_(model, 100); model.train()
crit = R.ReviewNetEnsembleCriterion(cfg)
opt = R.FusedClampAdam(model, lr=5e-4, weight_decay=1e-5, grad_clip=1.0)
_, _, labels, masks, top = HB.synthetic_inputs(cfg, B, 100, dev)
rng = np.random.default_rng(0)
imgs = [([rng.standard_normal(f['fc_feat_size']).astype(np.float32) for f in cfg.feat_array_info],
         [rng.standard_normal((f['att_num'], f['att_feat_size'])).astype(np.float32) for f in cfg.feat_array_info])
        for _ in range(n_img)]
# the reference's replicated host batch (pageable numpy, as DataLoader.get_batch returns it)
host_fc = [np.stack([im[0][i] for im in imgs for _ in range(spi)]) for i in range(w['M'])]
host_att = [np.stack([im[1][i] for im in imgs for _ in range(spi)]) for i in range(w['M'])]
feeder = FeatureFeeder(cfg.feat_array_info, n_img, spi, dev)
for s in range(2):
    feeder.stage(s, imgs)

def train(fc, att):
    opt.zero_grad()
    lp, reason = model(fc, att, labels)
    crit(lp, labels[:, 1:], masks[:, 1:], reason, top, 1.0).backward()
    opt.step()

def run(mode, steps=6):
    model.dedup_seq_per_img = spi if mode == 'feeder+dedup' else 0
    def one(step):
        if mode == 'replicated':
            fc = [torch.from_numpy(a).to(dev) for a in host_fc]
            att = [torch.from_numpy(a).to(dev) for a in host_att]
        else:
            slot = step & 1
            fc, att = feeder.batch(slot, expand=True)
            feeder.upload(slot ^ 1)              # next step's images fly while this step computes
        train(fc, att)
    if mode != 'replicated':
        feeder.upload(0)
    one(0); one(1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        one(s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({'mode': mode, 'captions_per_s': round(B / dt, 1), 'ms_per_step': round(dt * 1e3, 2),
                      'captions': B, 'images': n_img,
                      'pcie_MB_per_step': round((sum(a.nbytes for a in host_fc + host_att) if mode == 'replicated'
                                                 else feeder.pcie_bytes(0)) / 1e6, 1)}), flush=True)

for mode in ('replicated', 'feeder', 'feeder+dedup'):
    run(mode)

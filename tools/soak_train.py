"""Soak run: XE train steps at C3 (B=256) on one fixed synthetic batch -- the loss must fall on the printed samples,
device memory must stay flat and every parameter finite.  `--recipe` uses the published recipe (drop_prob_lm 0.3,
label smoothing, scheduled sampling 0.25: the step-wise sampled decoder pass).  Diagnostic, not a benchmark."""
import argparse
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

import bench as HB
import recurrent_fusion_network_amd as R

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=300)
ap.add_argument('--recipe', action='store_true')
ap.add_argument('--gemm', default='exact', choices=['exact', 'bf16x3'])
ap.add_argument('--workload', default='c3', choices=sorted(HB.WORKLOADS))
ap.add_argument('--batch', type=int, default=0, help='captions (default: the workload value)')
a = ap.parse_args()
dev = torch.device('cuda:0')
w = dict(HB.WORKLOADS[a.workload])
B = a.batch or w.get('B', 256)
cfg = HB.make_cfg(w)
if a.recipe:
    cfg.drop_prob_lm, cfg.use_label_smoothing = 0.3, 1
model = R.RecurrentFusionModel(cfg).to(dev)
HB.seeded_weights_(model, 100)
if a.gemm == 'bf16x3':
    import recurrent_fusion_network_amd._native as N
    model.gemm_flags |= N.GEMM_OPT_BF16X3
model.train()
model.ss_prob = 0.25 if a.recipe else 0.0
crit = R.ReviewNetEnsembleCriterion(cfg)
opt = R.FusedClampAdam(model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=1.0)
fc, att, labels, masks, top = HB.synthetic_inputs(cfg, B, 100, dev)
losses = []
for it in range(a.steps):
    opt.zero_grad()
    lp, tp = model(fc, att, labels)
    loss = crit(lp, labels[:, 1:], masks[:, 1:], tp, top, 1.0)
    loss.backward()
    opt.step()
    if it % 50 == 0 or it == a.steps - 1:
        torch.cuda.synchronize()
        losses.append(float(loss.detach()))
        print(it, 'loss %.4f' % losses[-1], 'alloc %.2f GB reserved %.2f GB peak %.2f GB' % (
            torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30,
            torch.cuda.max_memory_allocated() / 2**30), flush=True)
assert all(x == x for x in losses) and losses[-1] < losses[0]
print('finite params:', all(bool(torch.isfinite(p).all()) for p in model.parameters()))

#!/usr/bin/env python3
"""BASELINE.json configs[4]: beam-search eval (beam=5) + self-critical RL sample path, M=4, L=196, D=2048, B=128 on one
MI355X.  Prints one JSON line per mode (images/s, inputs resident in HBM).  Not the headline metric: see bench.py."""
import json, os, sys, time
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
rl_crit = R.ReviewNetRewardCriterion(cfg)
opt = R.FusedClampAdam(model, lr=5e-5, weight_decay=0.0, grad_clip=1.0)

def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out

def greedy():
    model.eval()
    with torch.no_grad():
        return model.sample(fc, att, {'sample_max': 1})
def beam():
    model.eval()
    with torch.no_grad():
        return model.sample(fc, att, {'beam_size': 5})
def rl_step():
    # train_rl.py:160-203: multinomial sample with grad, greedy baseline sample, reward criterion, backward, step
    model.train(); opt.zero_grad()
    seq, lp, lp_all, reason = model.sample(fc, att, {'sample_max': 0})
    with torch.no_grad():
        model.eval(); base = model.sample(fc, att, {'sample_max': 1})[0]; model.train()
    reward = torch.randn(B, 1, device=dev).expand(B, seq.size(1)).contiguous()   # CIDEr-D scoring is out of scope
    loss = rl_crit(lp, seq, reward, lp_all, 0.01, reason, top, 1.0, None, cfg)
    loss.backward(); opt.step()
    return loss

def rl_step_reuse():
    model.reuse_prefix = True
    try:
        return rl_step()
    finally:
        model.reuse_prefix = False

for name, fn, reps in (('greedy sample', greedy, 5), ('beam=5 sample_beam', beam, 3), ('RL step (sample+baseline+loss+bwd+Adam)', rl_step, 3),
                       ('RL step, model.reuse_prefix (baseline sample reuses stages I/II)', rl_step_reuse, 3)):
    dt, out = timed(fn, reps)
    print(json.dumps({'mode': name, 'images_per_s': round(B / dt, 1), 'ms': round(dt * 1e3, 2), 'B': B,
                      'config': 'M=4, L=196, D=2048, R=512, V+1=9488, seq=16'}), flush=True)

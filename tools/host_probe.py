"""Diagnostic: how much host time does one train step need?  The same C3 model at B = 2 (every kernel tiny, the same ~610
launches per step) runs at the speed of the launching host thread; the full-size step is device-bound as long as that
is comfortably below the device time."""
import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import bench as HB
import recurrent_fusion_network_amd as R
dev = torch.device('cuda:0')
w = HB.WORKLOADS['c3']
cfg = HB.make_cfg(w)
model = R.RecurrentFusionModel(cfg).to(dev); HB.seeded_weights_(model, 100); model.train()
crit = R.ReviewNetEnsembleCriterion(cfg)
opt = R.FusedClampAdam(model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=1.0)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fc, att, labels, masks, top = HB.synthetic_inputs(cfg, NB, 100, dev)
def step():
    opt.zero_grad()
    lp, tp = model(fc, att, labels)
    loss = crit(lp, labels[:, 1:], masks[:, 1:], tp, top, 1.0)
    loss.backward()
    opt.step(grad_scale=1.0)
for _ in range(3): step()
torch.cuda.synchronize()
for flags in (0, 4):
    model.gemm_flags = flags
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('flags %d: host enqueue %.1f ms/step, total %.1f ms/step' % (flags, (t1 - t0) * 100, (t2 - t0) * 100))
# host-only cost: same loop with the GPU idle-fast? measure CPU time per step via process_time
model.gemm_flags = 0
c0 = time.process_time(); t0 = time.perf_counter()
for _ in range(10): step()
c1 = time.process_time(); t1 = time.perf_counter()
torch.cuda.synchronize()
print('cpu time %.1f ms/step (wall enqueue %.1f)' % ((c1 - c0) * 100, (t1 - t0) * 100))

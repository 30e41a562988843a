"""Host time of one C2 train step (the model of `bench.py --workload c2` at B = 2: every kernel tiny, the same launches per step,
so the loop runs at the speed of the launching thread), split by where it is spent (cProfile, cumulative)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch  # noqa: E402

import bench as HB  # noqa: E402
import recurrent_fusion_network_amd as R  # noqa: E402

dev = torch.device('cuda:0')
cfg = HB.make_cfg(HB.WORKLOADS['c2'])
model = R.RecurrentFusionModel(cfg).to(dev)
HB.seeded_weights_(model, 100)
model.train()
crit = R.ReviewNetEnsembleCriterion(cfg)
opt = R.FusedClampAdam(model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=1.0)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 2
fc, att, labels, masks, top = HB.synthetic_inputs(cfg, NB, 100, dev)


def step():
    opt.zero_grad()
    lp, tp = model(fc, att, labels)
    loss = crit(lp, labels[:, 1:], masks[:, 1:], tp, top, 1.0)
    loss.backward()
    opt.step(grad_scale=1.0)


for _ in range(5):
    step()
torch.cuda.synchronize()
t0, c0 = time.perf_counter(), time.thread_time()
for _ in range(50):
    step()
t1, c1 = time.perf_counter(), time.thread_time()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('B = %d: host enqueue %.2f ms/step (thread CPU %.2f), with the device drained %.2f ms/step' % (
    NB, (t1 - t0) * 20, (c1 - c0) * 20, (t2 - t0) * 20))
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)

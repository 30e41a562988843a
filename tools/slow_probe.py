"""Diagnostic: GPU-side phase times (HIP events) of consecutive train steps from process start, to see what a slow episode
(74 ms steps on a fresh box, both GEMM modes alike) consists of."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as HB
import recurrent_fusion_network_amd as R
dev = torch.device('cuda:0')
cfg = HB.make_cfg(HB.WORKLOADS['c3'])
model = R.RecurrentFusionModel(cfg).to(dev); HB.seeded_weights_(model, 100); model.train()
if len(sys.argv) > 1 and sys.argv[1] == 'x3':
    model.gemm_flags |= 4
crit = R.ReviewNetEnsembleCriterion(cfg)
opt = R.FusedClampAdam(model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=1.0)
fc, att, labels, masks, top = HB.synthetic_inputs(cfg, 256, 100, dev)
N = 80
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(N)]
t0 = time.perf_counter()
for i in range(N):
    ev[i][0].record()
    opt.zero_grad()
    lp, tp = model(fc, att, labels)
    loss = crit(lp, labels[:, 1:], masks[:, 1:], tp, top, 1.0)
    ev[i][1].record()
    loss.backward()
    ev[i][2].record()
    opt.step(grad_scale=1.0)
    ev[i][3].record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3 / N
print('wall %.1f ms/step' % wall)
for i in range(0, N, 4):
    f = ev[i][0].elapsed_time(ev[i][1]); b = ev[i][1].elapsed_time(ev[i][2]); a = ev[i][2].elapsed_time(ev[i][3])
    nxt = ev[i][3].elapsed_time(ev[i + 1][0]) if i + 1 < N else 0.0
    print('step %2d  fwd %.1f  bwd %.1f  adam %.1f  gap-to-next %.2f  (sum %.1f)' % (i, f, b, a, nxt, f + b + a + nxt))

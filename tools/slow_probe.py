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
def run(n, label):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n)]
    host = []
    t0 = time.perf_counter()
    for i in range(n):
        h0 = time.perf_counter()
        ev[i][0].record()
        opt.zero_grad()
        lp, tp = model(fc, att, labels)
        loss = crit(lp, labels[:, 1:], masks[:, 1:], tp, top, 1.0)
        ev[i][1].record()
        loss.backward()
        ev[i][2].record()
        opt.step(grad_scale=1.0)
        ev[i][3].record()
        host.append((time.perf_counter() - h0) * 1e3)
    torch.cuda.synchronize()
    print('%s: wall %.1f ms/step' % (label, (time.perf_counter() - t0) * 1e3 / n))
    rows = []
    for i in range(n - 1):
        f = ev[i][0].elapsed_time(ev[i][1]); b = ev[i][1].elapsed_time(ev[i][2]); a = ev[i][2].elapsed_time(ev[i][3])
        g = ev[i][3].elapsed_time(ev[i + 1][0])
        rows.append((f, b, a, g, host[i]))
    for i in range(0, len(rows), max(1, len(rows) // 24)):
        f, b, a, g, h = rows[i]
        print('  step %3d  fwd %.1f  bwd %.1f  adam %.1f  gap %.2f  sum %.1f   host-enqueue %.1f' % (i, f, b, a, g, f + b + a + g, h))


run(30, 'exact')
model.gemm_flags |= 4
run(120, 'bf16x3 (first use of its kernels and workspaces)')
model.gemm_flags &= ~4
run(30, 'exact again')

"""A/B of the order of the deferred stage-I weight gradients in one process: every encoder's big a-bucket first (the product order since
round 4) against encoder by encoder (a0 b0 a1 b1 ...).  One MI355X: 65.32 vs 65.40 ms at B = 256, 13.86 vs 13.88 at B = 32.
    python tools/ab_wgrad_order.py [B ...]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench as HB
import recurrent_fusion_network_amd as R
dev = torch.device('cuda:0')
Bs = [int(x) for x in sys.argv[1:]] or [256]
for NB in Bs:
    w = HB.WORKLOADS['c3']; cfg = HB.make_cfg(w)
    model = R.RecurrentFusionModel(cfg).to(dev); HB.seeded_weights_(model, 100); model.train()
    crit = R.ReviewNetEnsembleCriterion(cfg)
    opt = R.FusedClampAdam(model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=1.0)
    fc, att, labels, masks, top = HB.synthetic_inputs(cfg, NB, 100, dev)
    def step():
        opt.zero_grad(); lp, tp = model(fc, att, labels)
        loss = crit(lp, labels[:, 1:], masks[:, 1:], tp, top, 1.0); loss.backward(); opt.step(); return loss
    for _ in range(30): keep = step()
    torch.cuda.synchronize()
    res = {False: [], True: []}
    for rnd in range(4):
        for inter in (False, True):
            model._wgrad_interleaved = inter
            for _ in range(5): keep = step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): keep = step()
            torch.cuda.synchronize(); res[inter].append((time.perf_counter() - t0) / 20 * 1e3)
    print('B=%d  a-buckets first: %s   interleaved: %s' % (NB, ['%.3f' % x for x in res[False]], ['%.3f' % x for x in res[True]]))
    del model, opt

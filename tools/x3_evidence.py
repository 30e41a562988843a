"""Evidence for RFN_GEMM_OPT_BF16X3 at the benchmarked shape: on the SAME weights and inputs,
max |d log-prob| and the worst relative gradient difference of bf16x3 vs the exact-f32 path at C3 with B = 256 and B = 32,
and -- at B = 32, where the CPU oracle finishes in seconds -- exact vs oracle and bf16x3 vs oracle next to them.  One JSON
line per batch size; copy into profiles/ (r03_x3_evidence.json).

    python tools/x3_evidence.py > gpurun_out/x3_evidence.jsonl
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench as HB  # noqa: E402
import recurrent_fusion_network_amd as R  # noqa: E402
import recurrent_fusion_network_amd._native as N  # noqa: E402
from oracle import rfn_oracle as O  # noqa: E402

dev = torch.device('cuda:0')
cfg = HB.make_cfg(HB.WORKLOADS['c3'])
crit = R.ReviewNetEnsembleCriterion(cfg)
for B in (32, 256):
    P = O.seeded_params(cfg, 21)
    fc, att, labels, masks, top = O.synthetic_batch(cfg, B, seed=22)
    model = R.RecurrentFusionModel(cfg)
    model.load_state_dict(P)
    model = model.to(dev).eval()
    d = lambda ts: [t.to(dev) for t in ts]  # noqa: E731
    dfc, datt, dl, dm, dt = d(fc), d(att), labels.to(dev), masks.to(dev), top.to(dev)

    def run(flags):
        model.gemm_flags = flags
        model.zero_grad(set_to_none=True)
        lp, reason = model(dfc, datt, dl)
        crit(lp, dl[:, 1:], dm[:, 1:], reason, dt, 1.0).backward()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        with torch.no_grad():
            seq, _, lp_all, _ = model.sample(dfc, datt, {'sample_max': 1})
        top2 = lp_all.topk(2, dim=2).values
        return lp.detach(), grads, seq, float((top2[:, :, 0] - top2[:, :, 1]).min())

    lp0, g0, s0, margin0 = run(0)
    lp1, g1, s1, _ = run(N.GEMM_OPT_BF16X3)
    rel = lambda a, b: max(float((a[k] - b[k]).abs().max()) / (1e-6 + float(a[k].abs().max())) for k in a  # noqa: E731
                       if not k.startswith('reason_linear'))     # heads: arg-max flips over steps (tests/test_fullsize_gpu.py)
    out = {'B': B, 'x3_vs_exact_max_dlogp': float((lp0 - lp1).abs().max()), 'x3_vs_exact_worst_rel_grad': rel(g0, g1),
           'greedy_ids_equal': bool(torch.equal(s0, s1)), 'min_greedy_margin_exact': margin0}
    if B == 32:
        o_lp = O.forward(cfg, P, fc, att, labels)[0]
        _, o_g = O.train_step_loss_and_grads(cfg, P, fc, att, labels, masks, top, 1.0)
        o_gd = {k: v.to(dev) for k, v in o_g.items()}
        out.update({'exact_vs_oracle_max_dlogp': float((lp0.cpu() - o_lp).abs().max()),
                    'x3_vs_oracle_max_dlogp': float((lp1.cpu() - o_lp).abs().max()),
                    'exact_vs_oracle_worst_rel_grad': rel(o_gd, g0), 'x3_vs_oracle_worst_rel_grad': rel(o_gd, g1),
                    'greedy_ids_equal_oracle': bool(torch.equal(s0.cpu(), O.sample_greedy(cfg, P, fc, att)[0]))})
    print(json.dumps(out), flush=True)
    del model
    torch.cuda.empty_cache()

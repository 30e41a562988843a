"""Where a phase of the persistent recurrence kernels spends its time (diagnostic build only):
    make -C recurrent_fusion_network_amd/csrc clean && make -C recurrent_fusion_network_amd/csrc -j8 EXTRA=-DRFN_CHAIN_TIMING
    python tools/chain_timing.py [c2|c3] [B]
Every block stamps the 100 MHz clock at the phase boundaries of every step (csrc/rfn_chain.hip CH_STAMP); printed per phase:
mean over blocks and steps of each segment, and the step's critical path (latest block at every boundary)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import recurrent_fusion_network_amd as R  # noqa: E402
from bench import WORKLOADS, make_cfg, synthetic_inputs, seeded_weights_  # noqa: E402

N = R._native
if not hasattr(N.lib, 'rfn_debug_chain_timing'):
    raise SystemExit('build with EXTRA=-DRFN_CHAIN_TIMING first')
wl = sys.argv[1] if len(sys.argv) > 1 else 'c2'
w = WORKLOADS[wl]
B = int(sys.argv[2]) if len(sys.argv) > 2 else w['B']
dev = torch.device('cuda:0')
cfg = make_cfg(w)
model = R.RecurrentFusionModel(cfg).to(dev)
seeded_weights_(model, 100)
model.train()
model.path_flags = N.PATH_OPT_PERSIST_ALL
crit = R.ReviewNetEnsembleCriterion(cfg)
opt = R.FusedClampAdam(model, lr=5e-4)
fc, att, labels, masks, top = synthetic_inputs(cfg, B, 100, dev)
STAMPS, MAXB, MAXS = 16, 256, 32
buf = torch.zeros(MAXB * MAXS * STAMPS, dtype=torch.int64, device=dev)
N.lib.rfn_debug_chain_timing.argtypes = [C.c_void_p, C.c_size_t, C.c_int]


def step():
    opt.zero_grad()
    lp, reason = model(fc, att, labels)
    loss = crit(lp, labels[:, 1:], masks[:, 1:], reason, top, 1.0)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
names = ['descriptor', 'phase 0 (K1 / Kb1) tiles', 'barrier 0', 'attention', 'barrier 1', 'phase 2 (K3 / Kb2) tiles', 'barrier 2']
# one chain per train step: the library stamps the pick-th persistent launch after the registration (a step launches stage II
# forward, decoder forward, decoder backward, stage II backward, in this order)
def report(tag):
    torch.cuda.synchronize()
    raw = buf.cpu()
    G, S = C.c_int(0), C.c_int(0)
    N.lib.rfn_debug_chain_last(C.byref(G), C.byref(S))
    G, S = G.value, S.value
    if G == 0 or not bool((raw[:G * S * STAMPS] != 0).any()):
        print(tag, ': no stamps (the chain ran as launches)')
        return
    x = raw[:G * S * STAMPS].view(G, S, STAMPS).double() * 0.01     # us (100 MHz)
    tile = x[:, :, 8:]                                # in-tile stamps: phase 0 (4), phase 2 (4)
    x = x[:, :, :8]
    seg = x[:, :, 1:] - x[:, :, :-1]                  # (G, S, 7)
    print('%s: %d blocks x %d steps; whole chain %.1f us = %.2f us per step' % (tag, G, S, float(x[:, :, 7].max() - x[:, :, 0].min()),
                                                                              float(x[:, :, 7].max() - x[:, :, 0].min()) / S))
    crit_path = x.max(0).values                       # latest block at every stamp
    cp = crit_path[:, 1:] - crit_path[:, :-1]
    for ph, base, start in (('phase 0', 0, 1), ('phase 2', 4, 5)):
        t = tile[:, :, base:base + 4]
        ok = t[:, :, 0] > 0
        if bool(ok.any()):
            d = lambda a_, b_: float((a_ - b_)[ok].mean())   # noqa: E731
            print('  %s tile (blocks that had one): weights requested + barrier waited %.2f us after the phase began | operands requested +%.2f | K loop done +%.2f | '
                  'epilogue done +%.2f' % (ph, d(t[:, :, 0], x[:, :, start]), d(t[:, :, 1], t[:, :, 0]), d(t[:, :, 2], t[:, :, 1]), d(t[:, :, 3], t[:, :, 2])))
    for i, n in enumerate(names):
        print('  %-28s mean over blocks %6.2f us   (busiest block %6.2f)   latest-block-to-latest-block %6.2f' % (
            n, float(seg[:, :, i].mean()), float(seg[:, :, i].mean(1).max()), float(cp[:, i].mean())))


for pick, tag in enumerate(['stage II forward', 'decoder forward', 'decoder backward (steps S-1 .. 1)', 'stage II backward (steps T2-1 .. 1)']):
    buf.zero_()
    torch.cuda.synchronize()
    N.lib.rfn_debug_chain_timing(buf.data_ptr(), buf.numel() * 8, pick)
    step()
    report(tag)

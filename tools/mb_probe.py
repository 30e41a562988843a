"""Diagnostic: what do micro-batches on separate HIP streams buy for the C3 XE train step (fwd + criterion + bwd)?

The step has ~50 ms of MFMA-bound grouped GEMMs and ~18 ms of latency-bound recurrence chains that depend on each
other; rows are independent, so two half-batches on two streams let one half's chains run under the other half's GEMMs.
Modes (all process the same 256 captions, no optimizer step):
  full      one batch, one stream (the r01 schedule)
  seq2      two half batches one after the other on one stream (cost of splitting)
  join2     two streams, both forwards -> join -> both backwards (a criterion on the whole batch joins the streams)
  free2     two streams, each half computes its own loss and runs backward without waiting for the other
  free2p    free2 with the first stream at high priority
  free4     four quarter batches, four streams
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as BN
import recurrent_fusion_network_amd as R

dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
w = dict(BN.WORKLOADS['c3'])
B = int(os.environ.get('MB_B', w['B']))
cfg = BN.make_cfg(w)
model = R.RecurrentFusionModel(cfg).to(dev)
BN.seeded_weights_(model, 100)
model.train()
PARAMS = list(model.parameters())


def reset():
    model._last_flat_grads.clear()
    for p in PARAMS:
        p.grad = None


crit = R.ReviewNetEnsembleCriterion(cfg)
fc, att, labels, masks, top = BN.synthetic_inputs(cfg, B, 100, dev)
labels[:] = labels[0]                                     # identical label rows: one cached step count for every slice
torch.cuda.synchronize()


_PARTS = {}


def parts(n):
    if n not in _PARTS:    # built once: the model caches the decoder step count per label tensor OBJECT
        lab, msk = labels[:B // n], masks[:B // n]
        _PARTS[n] = [([f[k * B // n:(k + 1) * B // n] for f in fc], [a[k * B // n:(k + 1) * B // n] for a in att],
                      lab, msk, top[k * B // n:(k + 1) * B // n]) for k in range(n)]
    return _PARTS[n]


def fwd(p):
    lp, tp = model(p[0], p[1], p[2])
    return lp, tp


def loss_of(p, lp, tp):
    return crit(lp, p[2][:, 1:], p[3][:, 1:], tp, p[4], 1.0)


def run_full():
    reset()
    p = parts(1)[0]
    lp, tp = fwd(p)
    loss_of(p, lp, tp).backward()


def run_seq(n):
    reset()
    for p in parts(n):
        lp, tp = fwd(p)
        loss_of(p, lp, tp).backward()


def run_streams(n, join, prio):
    reset()
    main = torch.cuda.current_stream()
    ps = parts(n)
    streams = STREAMS[(n, prio)]
    outs = []
    for s, p in zip(streams, ps):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            lp, tp = fwd(p)
            outs.append((lp, tp))
    if join:
        for s in streams:
            main.wait_stream(s)
        for s in streams:
            s.wait_stream(main)
    losses = []
    for s, p, (lp, tp) in zip(streams, ps, outs):
        with torch.cuda.stream(s):
            losses.append(loss_of(p, lp, tp))
    for ls in losses:
        ls.backward()          # autograd replays each half's nodes on the stream its forward ran on
    for s in streams:
        main.wait_stream(s)


STREAMS = {}
for n in (2, 4):
    STREAMS[(n, False)] = [torch.cuda.Stream() for _ in range(n)]
    STREAMS[(n, True)] = [torch.cuda.Stream(priority=-1 if k == 0 else 0) for k in range(n)]

MODES = [('full', run_full), ('seq2', lambda: run_seq(2)), ('join2', lambda: run_streams(2, True, False)),
         ('free2', lambda: run_streams(2, False, False)), ('free2p', lambda: run_streams(2, False, True)),
         ('join2p', lambda: run_streams(2, True, True)), ('free4', lambda: run_streams(4, False, False)),
         ('free4p', lambda: run_streams(4, False, True))]
only = os.environ.get('MB_MODES')
if only:
    MODES = [m for m in MODES if m[0] in only.split(',')]
res = {name: [] for name, _ in MODES}
for rnd in range(4):
    for name, fn in MODES:
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 3 * 1e3)
for name, _ in MODES:
    print('%-8s ms/step (fwd+loss+bwd, B=%d): %s   best %.2f' % (name, B, ' '.join('%.2f' % x for x in res[name]), min(res[name])))

"""Predicted data-parallel step times from single-GPU shard measurements (profiles/r03_shards.jsonl: bench.py --batch B on
one MI355X) -- so that the first real multi-GPU run has numbers to be compared with.

Model.  A rank's step = its shard's compute time (measured) + the part of the gradient exchange that is still running
when backward ends.  The exchange is one ring all-reduce per bucket, issued in the order backward finishes the buckets
(decoder, core, then enc_i a / enc_i b per encoder); bucket b of S_b bytes takes 2 (N-1)/N S_b / BW on the RCCL stream,
buckets queue behind each other, and a bucket cannot start before backward has produced it.  Ready times are taken as
fractions of the measured step (from the phase trace of the C3 step, profiles/r03_phases.txt: backward starts at 46 % of
the step, decoder bucket at 53 %, core at 60 %, the encoders' big buckets (a) at 60.5 ... 62 %, encoder i's small bucket
(b) at 62 % + (i + 1) * 8.6 %, clamp+Adam is the last 3.5 %) -- the fractions hold within a few points down to B = 32
because every phase scales with the batch except Adam.  BW: 153 GB/s = one xGMI link (a single ring, the pessimistic
end), 7 x 153 GB/s = all links of the fully connected node (the optimistic end).

Sharded update (round 5: FusedClampAdam(shard=...), bench.py --shard-optimizer): bucket k is reduce-scattered ((N-1)/N of its
bytes), the rank updates its 1/N of it (the measured 10.9 GB launch, ADAM_MS, becomes ADAM_MS / N spread over the buckets) and
the bucket's parameters are all-gathered ((N-1)/N of the bytes again) -- all three on a side stream under the rest of backward,
bucket after bucket in the order backward finishes them; what is still running when backward ends is exposed.  Printed as a
second table beside the all-reduce path.

    python tools/dp_predict.py [profiles/r03_shards.jsonl]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'profiles', 'r03_shards.jsonl')
meas = {}
for ln in open(path):
    d = json.loads(ln)
    meas[d['config']['captions_per_gpu']] = (d['ms_per_step'], d['bf16x3']['ms_per_step'])

MB = 1e6
BUCKETS = [('decoder', 54 * MB, 0.53), ('core', 262 * MB, 0.60)]
for i in range(4):       # round 4: every encoder's big a-bucket is produced before the first long att_2_att_h product
    BUCKETS.append(('enc%da' % i, 277 * MB, 0.60 + (i + 1) * 0.005))
for i in range(4):
    BUCKETS.append(('enc%db' % i, 34 * MB, 0.62 + (i + 1) * 0.08625))
ADAM = 0.035          # share of the step after the last bucket is produced
TOTAL = sum(b[1] for b in BUCKETS)


def step_ms(compute_ms, n, bw):
    """compute_ms: measured single-GPU step of the shard; returns (step, exposed) in ms."""
    if n == 1:
        return compute_ms, 0.0
    t_done = 0.0
    for _, size, frac in BUCKETS:
        ready = frac * compute_ms
        t_done = max(t_done, ready) + 2.0 * (n - 1) / n * size / bw * 1e3
    bwd_end = (1.0 - ADAM) * compute_ms
    exposed = max(0.0, t_done - bwd_end)
    return compute_ms + exposed, exposed


print('gradient exchange per step: %.2f GB in %d buckets; ring all-reduce moves 2 (N-1)/N of that per GPU' % (TOTAL / 1e9, len(BUCKETS)))
print()
MID = 350e9      # a mid estimate: the bus bandwidth RCCL typically sustains for large all-reduces on an 8-GPU xGMI node
print('| N | scaling | captions / rank | shard step, exact / bf16x3 (measured, ms) | step at 153 GB/s (exact / x3) | step at 350 GB/s (exact / x3) | step at 1071 GB/s (exact / x3) | captions/s at 350 / 1071 GB/s (exact) |')
print('|---|---|---|---|---|---|---|---|')
for n in (1, 2, 4, 8):
    for kind in ('weak', 'strong'):
        b = 256 if kind == 'weak' else 256 // n
        if b not in meas or (n == 1 and kind == 'strong'):
            continue
        ex, x3 = meas[b]
        lo = [step_ms(t, n, 153e9) for t in (ex, x3)]
        mid = [step_ms(t, n, MID) for t in (ex, x3)]
        hi = [step_ms(t, n, 7 * 153e9) for t in (ex, x3)]
        fmt = lambda r: '%.1f / %.1f (exposed %.1f / %.1f)' % (r[0][0], r[1][0], r[0][1], r[1][1])  # noqa: E731
        print('| %d | %s | %d | %.1f / %.1f | %s | %s | %s | %.0f / %.0f |' % (
            n, kind, b, ex, x3, fmt(lo), fmt(mid), fmt(hi), n * b / mid[0][0] * 1e3, n * b / hi[0][0] * 1e3))


# ---- sharded update -----------------------------------------------------------------------------------------------------------
ADAM_MS = 2.3            # the clamp+Adam launch over all 390 M parameters (2.0-2.45 ms by box, profiles/r05_pmc_step.md)
PREFIX_BYTES = sum(sz for name, sz, _ in BUCKETS if name != 'decoder')


def step_ms_sharded(compute_ms, n, bw):
    """(step, exposed) in ms with the update sharded over n ranks and pipelined under backward."""
    if n == 1:
        return compute_ms, 0.0
    body = compute_ms - ADAM_MS                       # the shard's step without the update launch
    t_done = 0.0
    for _, size, frac in BUCKETS:
        ready = frac * compute_ms
        wire = (n - 1) / n * size / bw * 1e3          # reduce-scatter, and again the all-gather
        t_done = max(t_done, ready) + wire + ADAM_MS / n * size / TOTAL + wire
    exposed = max(0.0, t_done - body)
    return body + exposed, exposed


print()
print('sharded update (reduce-scatter + 1/N update + all-gather per bucket, under backward), exact f32, predicted step in ms (all-reduce path beside it):')
print('| N | scaling | captions / rank | at 153 GB/s: all-reduce / sharded | at 350 GB/s | at 1071 GB/s |')
print('|---|---|---|---|---|---|')
for n in (2, 4, 8):
    for kind in ('weak', 'strong'):
        b = 256 if kind == 'weak' else 256 // n
        if b not in meas:
            continue
        ex = meas[b][0]
        cells = []
        for bw in (153e9, MID, 7 * 153e9):
            cells.append('%.1f / %.1f' % (step_ms(ex, n, bw)[0], step_ms_sharded(ex, n, bw)[0]))
        print('| %d | %s | %d | %s |' % (n, kind, b, ' | '.join(cells)))

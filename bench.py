#!/usr/bin/env python3
"""Headline benchmark: captions/sec of one XE train step of the recurrent-fusion decoder on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): RecurrentFusionModel,
M=4 encoders, L=196 regions, D=2048, B=256 captions per GPU, R=A=E=512, T1=T2=8, V+1=9488, seq_length=16
(labels with 18 columns -> exactly 17 decoder steps).  Synthetic N(0,1) features and seeded uniform(+-0.1)
weights, resident in HBM before the timed region.

One step = the reference's timed region train.py:143-166: zero_grad -> forward -> ReviewNetEnsembleCriterion ->
backward -> (N>1: one RCCL sum all-reduce per gradient bucket, overlapped with backward) -> clamp + Adam.  fp32 throughout.
Rank 0 prints ONE JSON line; `value` is the whole-job aggregate over all N GPUs.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (spec)

WORKLOADS = {
    # name: (M, L, D, B per GPU)
    'c3': dict(M=4, L=196, D=2048, B=256, desc='C3'),
    'c2': dict(M=2, L=49, D=512, B=64, desc='C2'),
}


def make_cfg(w):
    """The `opt` Namespace fields the model and the criteria read (reference opts.py defaults).
    Built here, not taken from oracle/: the oracle is only touched by the cpu_baseline leg."""
    from types import SimpleNamespace
    info = [dict(att_num=w['L'], att_feat_size=w['D'], fc_feat_size=w['D']) for _ in range(w['M'])]
    return SimpleNamespace(
        caption_model='recurrent_fusion_model', vocab_size=9487, input_encoding_size=512, rnn_type='lstm',
        rnn_size=512, num_layers=1, drop_prob_lm=0.0, drop_prob_reason=0.0, drop_prob_fusion=0.0, seq_length=16,
        num_review_steps=8, num_review_steps_0=8, top_words_count=1000, att_hid_size=512, review_maxout=0,
        maxout=0, fusion_maxout=0, use_cuda=1, feat_array_info=info, use_label_smoothing=0,
        label_smoothing_epsilon=0.1, use_ppo=0, ppo_clip=0.2)


def synthetic_inputs(cfg, B, seed, dev):
    """SURVEY.md 8d, generated on the device (1.64 GB of features at C3)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    fc = [torch.randn(B, f['fc_feat_size'], generator=g, device=dev) for f in cfg.feat_array_info]
    att = [torch.randn(B, f['att_num'], f['att_feat_size'], generator=g, device=dev) for f in cfg.feat_array_info]
    S = cfg.seq_length
    labels = torch.zeros(B, S + 2, dtype=torch.long, device=dev)
    labels[:, 1:S + 1] = torch.randint(1, cfg.vocab_size + 1, (B, S), generator=g, device=dev)
    masks = torch.ones(B, S + 2, device=dev)
    top = -torch.ones(B, cfg.top_words_count, dtype=torch.long, device=dev)
    for b in range(B):
        top[b, :5] = torch.randperm(cfg.top_words_count, generator=g, device=dev)[:5]
    return fc, att, labels, masks, top


def seeded_weights_(model, seed):
    """uniform(+-0.1) for every parameter from a seeded device generator (checkpoints are not available)."""
    g = torch.Generator(device=next(model.parameters()).device).manual_seed(seed)
    with torch.no_grad():
        for _, p in sorted(model.named_parameters()):
            p.uniform_(-0.1, 0.1, generator=g)


def time_dominant_kernel(model, att, reps):
    """HIP-event timing of the dominant kernel at the workload's shapes, on the stream the path launches on:
    the grouped fp32-MFMA GEMM that applies all T1 att_2_att_h step weights of one encoder to its
    (B*L, D) feature matrix (rfn_prefix_fwd's first launch per encoder).  Returns (avg seconds, flops)."""
    import recurrent_fusion_network_amd._native as N
    B, L, D = att[0].shape
    A, T1 = model.att_hid_size, model.num_review_steps_0
    out = torch.empty(B * L, T1 * A, device=att[0].device)
    probs = []
    for t in range(T1):
        cell = model.review_steps_individual[t].lstm[0].att_model.att_2_att_h
        probs.append((out[:, t * A:], T1 * A, [(att[0], D, 1, cell.weight, D, 1, D, cell.bias)]))
    flops = 2.0 * B * L * D * A * T1
    N.gemm(B * L, A, probs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        N.gemm(B * L, A, probs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps, flops


def cpu_baseline(cfg, sample_B, seed):
    """The CPU oracle (kind "port": a PyTorch-CPU restatement validated against the reference, see oracle/)
    timed on this host on a bounded sample of the same workload: one XE train step (forward + criterion +
    backward) at the full model size on a few captions; cost is linear in B.  sample_B = 0 sizes the sample from
    a B=8 probe so that the timed step is about 15 s of CPU work whatever the host."""
    from oracle import rfn_oracle as O
    P = O.seeded_params(cfg, seed)

    def run(nb):
        fc, att, labels, masks, top = O.synthetic_batch(cfg, nb, seed=seed + 1)
        t0 = time.perf_counter()
        O.train_step_loss_and_grads(cfg, P, fc, att, labels, masks, top)
        return time.perf_counter() - t0

    run(1)                                              # warm the allocator and the thread pool
    if sample_B <= 0:
        probe = run(8)
        sample_B = int(min(64, max(8, round(8 * 15.0 / max(probe, 1e-3) / 8) * 8)))
        dt = run(sample_B)
        for _ in range(2):                              # fast host: per-caption cost still falls with B, grow the sample
            if dt >= 10.0 or sample_B >= 128:
                break
            sample_B = int(min(128, max(sample_B + 8, round(sample_B * 15.0 / dt / 8) * 8)))
            dt = run(sample_B)
    else:
        dt = run(sample_B)
    return dict(value=round(sample_B / dt, 4), unit='captions/s', cores=torch.get_num_threads(), kind='port',
                sample='1 XE train step (fwd+loss+bwd) of the same model at B=%d captions, %.1f s' % (sample_B, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='c3', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=0, help='captions per GPU (default: the workload value)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--label-smoothing', action='store_true', help='XE criterion with label smoothing 0.1 (SURVEY 8d)')
    ap.add_argument('--cpu-sample', type=int, default=0,
                    help='captions in the CPU-baseline sample (0: sized for ~15 s of CPU work from a B=8 probe)')
    args = ap.parse_args()

    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        # data parallel: keep the big GEMM tiles single-buffered (<= 110 KB LDS per CU) so RCCL's kernels can run
        # beside the long weight-gradient GEMMs instead of waiting for them (read once by librfn_hip.so)
        os.environ.setdefault('RFN_GEMM_LDS_LEAN', '1')
    import recurrent_fusion_network_amd as R
    from recurrent_fusion_network_amd import parallel as DP

    rank, world, local = DP.init_from_env('nccl')
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run' % (args.gpus, world))
    dev = torch.device('cuda', int(os.environ.get('RFN_DEVICE_INDEX', local)))   # test hook: ranks sharing one GPU
    torch.cuda.set_device(dev)
    w = dict(WORKLOADS[args.workload])
    B = args.batch or w['B']
    cfg = make_cfg(w)
    cfg.use_label_smoothing = int(args.label_smoothing)
    torch.manual_seed(100 + rank)            # opts.py:178 default seed, + rank (train.py:23)

    model = R.RecurrentFusionModel(cfg).to(dev)
    seeded_weights_(model, 100)              # identical replicas on every rank
    model.train()                            # dropout probabilities are 0 (opts.py defaults)
    crit = R.ReviewNetEnsembleCriterion(cfg)
    opt = R.FusedClampAdam(model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=1.0)
    fc, att, labels, masks, top = synthetic_inputs(cfg, B, 100 + rank, dev)

    sync = DP.GradSync(model, world)     # per-bucket async all-reduce, overlapped with the rest of backward

    def step():
        opt.zero_grad()
        log_prob, top_pred = model(fc, att, labels)
        loss = crit(log_prob, labels[:, 1:], masks[:, 1:], top_pred, top, 1.0)
        loss.backward()
        scale = sync.finish()              # 1/world: applied before the clamp inside the fused update
        opt.step(grad_scale=scale)
        return loss

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = DP.max_over_ranks(time.perf_counter() - t0, world, dev)
    final_loss = float(loss.detach())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            'metric': 'captions/sec (train fwd+bwd) at B=256, M=4, L=196, D=2048, seq=16; 1/2/4/8 GPU',
            'value': round(world * B * args.steps / elapsed, 2), 'unit': 'captions/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s: RecurrentFusionModel XE train step (zero_grad+fwd+criterion+bwd+clamp+Adam), '
                                   'M=%d encoders, L=%d, D=%d, R=A=E=512, T1=T2=8, K=1000, V+1=9488, seq=16 '
                                   '(17 decoder steps)' % (w['desc'], w['M'], w['L'], w['D']),
                       'captions_per_gpu': B, 'global_batch': world * B,
                       'parallelism': 'dp%d (batch sharded, RCCL all-reduce of grads)' % world if world > 1 else 'single GPU',
                       'final_loss': round(final_loss, 4)},
        }
        # roofline of the dominant kernel, timed live with HIP events on the launch stream
        secs, flops = time_dominant_kernel(model, att, reps=5)
        achieved = flops / secs / 1e12
        traffic = None               # HBM-side bytes per launch of that kernel, from the rocprofv3 --pmc passes
        tfile = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(tfile):
            try:
                gb = json.load(open(tfile)).get(args.workload)
                traffic = None if gb is None else int(gb * 1e9)
            except Exception:
                traffic = None
        out['roofline'] = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': FP32_MFMA_PEAK_TFLOPS,
                           'unit': 'TFLOP/s', 'frac': round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), 'traffic': traffic,
                           'traffic_unit': 'bytes per launch (2*FETCH_SIZE + WRITE_SIZE)',
                           'algorithmic_bytes': int(4 * (B * w['L'] * w['D'] + 8 * 512 * w['D'] + B * w['L'] * 8 * 512)),
                           'kernel': 'rfn_gemm_kernel<128,128,kfast,kfast,vec,...,tail> (grouped att_2_att_h projection, '
                                     '%.3f TFLOP per launch, %.3f ms per launch)' % (flops / 1e12, secs * 1e3)}
        # whole-step view against the same peak (SURVEY.md 8d algorithmic FLOP of one step)
        step_flops = {'c3': 7.667e12, 'c2': 0.1706e12}[args.workload] * (B / w['B'])
        out['roofline']['step_frac'] = round(step_flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, args.cpu_sample, 100)
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

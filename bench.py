#!/usr/bin/env python3
"""Headline benchmark: captions/sec of one XE train step of the recurrent-fusion decoder on MI355X.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: before it imports torch or touches a
GPU it starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py`
as a child, relays rank 0's JSON line and exits with the child's code.  Under an external launcher (WORLD_SIZE set)
it is one rank.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): RecurrentFusionModel,
M=4 encoders, L=196 regions, D=2048, B=256 captions per GPU, R=A=E=512, T1=T2=8, V+1=9488, seq_length=16
(labels with 18 columns -> exactly 17 decoder steps).  Synthetic N(0,1) features and seeded uniform(+-0.1)
weights, resident in HBM before the timed region.

One step = the reference's timed region train.py:143-166: zero_grad -> forward -> ReviewNetEnsembleCriterion ->
backward -> (N>1: one RCCL sum all-reduce per gradient bucket, overlapped with backward) -> clamp + Adam.  fp32
throughout.  Rank 0 prints ONE JSON line; `value` is the whole-job aggregate over all N GPUs.

The default single-GPU line also carries `"secondary": {"c2", "c5_beam5", "c5_greedy", "c5_rl"}`: BASELINE configs 2 (eager and
replayed from a HIP graph) and 5 (beam = 5, greedy, the RL step at B = 128) timed in the same process after the headline and the
bf16x3 leg, each with ms_per_step, value, step_frac and the roofline it is priced against (`--no-secondary` skips them).

Other lines (same schema, not the headline): --workload c2 | c3het (the reference's shipped 5 heterogeneous encoders,
feat_array.py:240-244) | c5 (BASELINE configs[4]: greedy / beam=5 decode and the self-critical RL step at B=128);
--recipe (the published XE recipe train_recurrent_fusion_model.sh:17-27: drop_prob_lm 0.3, label smoothing, scheduled
sampling); --strong (global batch 256 sharded over the ranks instead of 256 per rank).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (spec)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (spec); --gemm bf16x3 is priced against it
METRIC = 'captions/sec (train fwd+bwd) at B=256, M=4, L=196, D=2048, seq=16; 1/2/4/8 GPU'
CPU_SAMPLE_B = 32               # captions in the CPU-baseline sample: fixed, so rounds and hosts are comparable

WORKLOADS = {
    # encoders: (att_num L, att_feat_size D, fc_feat_size F) per encoder; B = captions per GPU
    'c3': dict(desc='C3', B=256, enc=[(196, 2048, 2048)] * 4, step_tflop=7.667),
    'c2': dict(desc='C2', B=64, enc=[(49, 512, 512)] * 2, step_tflop=0.1706),
    # resnet / inception_v4 / inception_v3 / densenet / inception_resnet_v2 (feat_array.py:6-9,53-56,100-103,147-150,194-197)
    'c3het': dict(desc='shipped 5 heterogeneous encoders', B=256,
                  enc=[(196, 2048, 2048), (64, 1536, 1536), (64, 1280, 2048), (49, 2208, 2208), (64, 1536, 1536)],
                  step_tflop=None),
    # the same five encoders with the 196 x 2048 one LAST (A/B of launch-order effects on its projection: profiles/r06_c3het.md)
    'c3het_last': dict(desc='shipped 5 heterogeneous encoders, the 196 x 2048 map last', B=256,
                       enc=[(64, 1536, 1536), (64, 1280, 2048), (49, 2208, 2208), (64, 1536, 1536), (196, 2048, 2048)],
                       step_tflop=None),
    'c5': dict(desc='C5 decode', B=128, enc=[(196, 2048, 2048)] * 4, step_tflop=None),
}
for _w in WORKLOADS.values():       # uniform-encoder shorthands used by tools/
    _w['M'], _w['L'], _w['D'] = len(_w['enc']), _w['enc'][0][0], _w['enc'][0][1]


def make_cfg(w):
    """The `opt` Namespace fields the model and the criteria read (reference opts.py defaults).
    Built here, not taken from oracle/: the oracle is only touched by the cpu_baseline leg."""
    from types import SimpleNamespace
    info = [dict(att_num=L, att_feat_size=D, fc_feat_size=F) for (L, D, F) in w['enc']]
    return SimpleNamespace(
        caption_model='recurrent_fusion_model', vocab_size=9487, input_encoding_size=512, rnn_type='lstm',
        rnn_size=512, num_layers=1, drop_prob_lm=0.0, drop_prob_reason=0.0, drop_prob_fusion=0.0, seq_length=16,
        num_review_steps=8, num_review_steps_0=8, top_words_count=1000, att_hid_size=512, review_maxout=0,
        maxout=0, fusion_maxout=0, use_cuda=1, feat_array_info=info, use_label_smoothing=0,
        label_smoothing_epsilon=0.1, use_ppo=0, ppo_clip=0.2)


def train_step_flops(cfg, B, S=17):
    """Algorithmic FLOP of one XE train step (SURVEY.md 8d accounting: backward = 2x forward, except the attention
    feature projection and context, whose input needs no gradient: 1x).  Reproduces 7.667 TF at C3 and 0.1706 at C2
    to within a few tenths of a percent; used for the workloads SURVEY gives no figure for."""
    R, A, E, K, V1 = cfg.rnn_size, cfg.att_hid_size, cfg.input_encoding_size, cfg.top_words_count, cfg.vocab_size + 1
    T1, T2, M = cfg.num_review_steps_0, cfg.num_review_steps, len(cfg.feat_array_info)
    nograd = grad = 0.0
    for f in cfg.feat_array_info:
        L, D, F = f['att_num'], f['att_feat_size'], f['fc_feat_size']
        nograd += 2.0 * B * L * D * A * T1 + 2.0 * B * L * D * T1          # att_2_att_h projection + context bmm
        grad += 2.0 * B * F * R                                            # fc2h
        grad += T1 * (2.0 * B * R * A + 2.0 * B * L * A + 2.0 * B * (M * R + D) * 4 * R)
        grad += 2.0 * T1 * B * R * K
    grad += T2 * (M * (2.0 * T1 * B * R * A + 2.0 * B * R * A + 4.0 * B * T1 * A + 2.0 * B * T1 * R)
                  + 2.0 * B * R * 4 * R * (M + 1)) + 2.0 * T2 * B * R * K
    grad += 2.0 * T2 * B * R * A + S * (2.0 * B * R * A + 4.0 * B * T2 * A + 2.0 * B * T2 * R
                                        + 2.0 * B * (E + 2 * R) * 4 * R + 2.0 * B * R * V1)
    return 2.0 * nograd + 3.0 * grad


def synthetic_inputs(cfg, B, seed, dev):
    """SURVEY.md 8d, generated on the device (1.64 GB of features at C3)."""
    import torch
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
    import torch
    g = torch.Generator(device=next(model.parameters()).device).manual_seed(seed)
    with torch.no_grad():
        for _, p in sorted(model.named_parameters()):
            p.uniform_(-0.1, 0.1, generator=g)


def time_dominant_kernel(model, att, reps):
    """HIP-event timing of the dominant kernel at the workload's shapes, on the stream the path launches on:
    the grouped fp32-MFMA GEMM that applies all T1 att_2_att_h step weights of encoder 0 to its
    (B*L, D) feature matrix (rfn_prefix_fwd's first big launch).  Returns (avg seconds, flops)."""
    import torch
    import recurrent_fusion_network_amd._native as N
    B, L, D = att[0].shape
    A, T1 = model.att_hid_size, model.num_review_steps_0
    out = torch.empty(T1, B * L, A, device=att[0].device)          # step-major slabs, as rfn_prefix_fwd lays them out
    probs = []
    for t in range(T1):
        cell = model.review_steps_individual[t].lstm[0].att_model.att_2_att_h
        probs.append((out[t], A, [(att[0], D, 1, cell.weight, D, 1, D, cell.bias)]))
    flops = 2.0 * B * L * D * A * T1
    if int(getattr(model, 'gemm_flags', 0)) & N.GEMM_OPT_BF16X3:
        # the same product on the bf16 matrix cores: plane images made outside the timed region, the GEMM launch alone
        # is timed (the split passes are separate, HBM-bound kernels; they are inside the step time, not inside this)
        cells = [model.review_steps_individual[t].lstm[0].att_model.att_2_att_h for t in range(T1)]
        img_x = N.x3_image([att[0].view(B * L, D)], B * L, D)
        img_w = N.x3_image([c.weight for c in cells], A, D)
        outs, bias = [out[t] for t in range(T1)], [c.bias for c in cells]
        run = lambda: N.x3_gemm(B * L, T1 * A, D, img_x, img_w, outs, gm=B * L, gn=A, ldc=A, bias=bias)  # noqa: E731
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps, flops
    N.gemm(B * L, A, probs, flags=int(getattr(model, 'gemm_flags', 0)))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        N.gemm(B * L, A, probs, flags=int(getattr(model, 'gemm_flags', 0)))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps, flops


def source_sha16(rel):
    import hashlib
    with open(os.path.join(ROOT, rel), 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def pmc_traffic(key, same_batch=True):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json), or None.  Every entry carries the hash of the kernel source it was measured at: a
    kernel that has changed since reports null instead of a stale figure (re-run tools/run_gemm_pmc.sh / run_x3_pmc.sh
    and tools/stamp_pmc_traffic.py)."""
    if not same_batch:
        return None
    try:
        ent = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'))).get(key)
        if not isinstance(ent, dict) or ent.get('src_sha16') != source_sha16(ent['src']):
            return None
        return int(ent['gb'] * 1e9)
    except Exception:      # noqa: BLE001
        return None


def cpu_baseline(cfg, sample_B, seed, mode='train'):
    """The CPU oracle (kind "port": a PyTorch-CPU restatement validated against the reference, see oracle/)
    timed on this host on a bounded sample of the same workload at the full model size: one XE train step
    (forward + criterion + backward) or, for the decode workload, one greedy sample(); cost is linear in B."""
    import torch
    from oracle import rfn_oracle as O
    P = O.seeded_params(cfg, seed)

    def run(nb):
        fc, att, labels, masks, top = O.synthetic_batch(cfg, nb, seed=seed + 1)
        t0 = time.perf_counter()
        if mode == 'train':
            O.train_step_loss_and_grads(cfg, P, fc, att, labels, masks, top)
        else:
            with torch.no_grad():
                O.sample_greedy(cfg, P, fc, att)
        return time.perf_counter() - t0

    run(1)                                              # warm the allocator and the thread pool
    dt = run(sample_B)
    what = '1 XE train step (fwd+loss+bwd)' if mode == 'train' else '1 greedy sample() (stages I/II + 17 decoder steps)'
    unit = 'captions/s' if mode == 'train' else 'images/s'
    return dict(value=round(sample_B / dt, 4), unit=unit, cores=torch.get_num_threads(), kind='port',
                sample='%s of the same model at B=%d, %.1f s' % (what, sample_B, dt))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='c3', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=0, help='captions per GPU (default: the workload value)')
    ap.add_argument('--strong', action='store_true',
                    help='strong scaling: the workload batch is the GLOBAL batch, sharded over the ranks')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--label-smoothing', action='store_true', help='XE criterion with label smoothing 0.1 (SURVEY 8d)')
    ap.add_argument('--drop-lm', type=float, default=0.0, help='drop_prob_lm (decoder dropout)')
    ap.add_argument('--ss-prob', type=float, default=0.0, help='scheduled-sampling probability')
    ap.add_argument('--recipe', action='store_true',
                    help='published XE recipe: --drop-lm 0.3 --label-smoothing --ss-prob 0.25')
    ap.add_argument('--gemm', default='exact', choices=['exact', 'bf16x3'],
                    help="exact: every product on the exact-f32 MFMA (default, the headline).  bf16x3: the hoisted stage-I "
                         "projection and its weight gradient on the bf16 matrix cores, f32 operands as three bf16 planes, six "
                         "plane products, f32 accumulation (RFN_GEMM_OPT_BF16X3): f32-level accuracy, not bit-identical")
    ap.add_argument('--settle', type=float, default=2.0,
                    help='seconds of untimed steps BEFORE the W warm-up steps (the first second of GPU work after an idle '
                         'spell -- e.g. behind a CPU-only phase of the caller -- ran up to 13 %% slow on some boxes while '
                         'the kernels themselves timed normal); reported as settle_s')
    ap.add_argument('--trace-steps', action='store_true',
                    help='diagnostic: HIP events around the forward / backward / update of every timed step, printed to '
                         'stderr after the run (no extra synchronisation inside the timed region)')
    ap.add_argument('--no-alt-line', action='store_true',
                    help="with --gemm exact (the default): do not append the same workload re-timed with --gemm bf16x3 "
                         "(the 'bf16x3' object of the JSON line; `value` is always the exact-f32 measurement)")
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the `secondary` object of the default single-GPU line (BASELINE configs 2 and 5 timed after the headline)')
    ap.add_argument('--cpu-sample', type=int, default=CPU_SAMPLE_B, help='captions in the CPU-baseline sample')
    ap.add_argument('--micro-batches', type=int, default=-1, help='override model.micro_batches (-1: model default)')
    ap.add_argument('--lds-lean', action='store_true',
                    help='force RFN_GEMM_OPT_LDS_LEAN (GradSync sets it by itself when world > 1): the big-tile configuration '
                         'of a data-parallel run on one rank')
    ap.add_argument('--digest', action='store_true',
                    help="add config.digest: a hash of the loss bits and of every parameter bucket's float64 sum / sum of "
                         'squares after the timed steps -- two runs that agree bit for bit print the same digest')
    ap.add_argument('--graph', action='store_true',
                    help='replay the train step from a captured HIP graph (graphed.GraphedTrainStep; single GPU, dropout 0): for '
                         'configurations whose 350 launches a slow host cannot issue as fast as the device retires them')
    ap.add_argument('--persist', type=int, default=0,
                    help='A/B: RFN_PATH_OPT_PERSIST_* bits (1 decoder fwd, 2 stage II fwd, 4 decoder bwd, 8 stage II bwd; 15 = all): '
                         'those recurrences inside ONE persistent launch each (csrc/rfn_chain.hip) instead of three launches per step; '
                         '16 = RFN_PATH_OPT_DEEP_CELLS: few-tile per-step products on the deep-ring kernel instead of the 3-slot one; '
                         '32 = RFN_PATH_OPT_NO_SMALL_TILES: keep 32-row tiles where the library would take 16-row ones; '
                         '64 = RFN_PATH_OPT_SHARED_SMALL_TILES: those 16-row tiles on block-shared ring slots (variants 4 / 5); '
                         'bit-identical either way (profiles/r05_chain.md); '
                         '128 = RFN_PATH_OPT_DEC_UNHOISTED: the three-launch decoder cell of rounds 3-5 (z2h(z) as a per-step product) '
                         'instead of the hoisted two-launch form (csrc/rfn_deccell.hip); same mathematics, different rounding')
    ap.add_argument('--fused-loss', action='store_true',
                    help='forward + criterion through RecurrentFusionModel.forward_loss (the language term straight from the '
                         'logits, d logits written in place: no (B, T, V+1) log_prob / d log_prob round trip) instead of '
                         'model(...) + crit(...); same loss and gradients to rounding')
    ap.add_argument('--shard-optimizer', action='store_true',
                    help='data parallel only: every rank updates 1/N of each flat bucket (FusedClampAdam(shard=...): '
                         'reduce-scatter of the gradients, Adam on the shard, all-gather of the parameters under the next '
                         'forward) instead of all-reduce + the full update on every rank; bit-identical parameters')
    ap.add_argument('--overlap-update', action='store_true',
                    help='single GPU, opt-in A/B: each bucket\'s clamp + Adam on a side stream as soon as backward has finished the '
                         'bucket (parallel.OverlappedUpdate) instead of one launch after backward; bit-identical parameters')
    ap.add_argument('--selftest-launch', action='store_true',
                    help='launcher / rendezvous check without a GPU: ranks meet, reduce a timing, rank 0 prints the line')
    args = ap.parse_args(argv)
    if args.recipe:
        args.drop_lm, args.label_smoothing, args.ss_prob = 0.3, True, 0.25
    return args


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """Parent of an N-rank run: starts fresh children through torch.distributed.run BEFORE anything in this process
    has imported torch or touched a GPU, relays rank 0's JSON line, returns the children's exit code."""
    port = int(os.environ.get('MASTER_PORT', 0)) or _free_port()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: RCCL needs it on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
    import uuid
    env['RFN_BENCH_RUN_ID'] = uuid.uuid4().hex       # names this launch's RunGuard flag file: no earlier run can share it
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        if out.startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    try:
        os.unlink(run_guard_path(env))               # a failed run leaves its flag behind: the parent removes it
    except OSError:
        pass
    if rc != 0:
        sys.stderr.write('bench.py: a rank failed (torch.distributed.run exit code %d)\n' % rc)
        return rc
    if line is None:
        sys.stderr.write('bench.py: the ranks exited without printing a result line\n')
        return 1
    print(line, flush=True)
    return 0


def selftest_rank(args):
    """One rank of --selftest-launch: the rendezvous, barrier, max-over-ranks timing and the failure discipline (RunGuard)
    of the real run on CPU tensors; no model, no GPU.  `value` is null: this line proves the launch recipe, it is not a
    measurement."""
    import torch
    from recurrent_fusion_network_amd import parallel as DP
    rank, world, _ = DP.init_from_env(os.environ.get('RFN_DIST_BACKEND', 'gloo'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    guard = RunGuard(rank, world)
    fail = os.environ.get('RFN_BENCH_FAIL_RANK') == str(rank)     # test hooks: a rank that dies mid-run
    where = os.environ.get('RFN_BENCH_FAIL_WHERE', 'headline')
    cpu = torch.device('cpu')

    def exchange():
        flat = torch.full((1024,), float(rank + 1))
        for wk in DP.allreduce_flat([flat], world, async_op=True):
            wk.wait()
        assert float(flat[0]) == world * (world + 1) / 2.0

    try:
        if world > 1:
            torch.distributed.barrier()
        t0 = time.perf_counter()
        if fail and where == 'headline':      # the peers are inside the all-reduce below when this rank gives up
            raise RuntimeError('injected failure on rank %d in the headline leg (RFN_BENCH_FAIL_RANK)' % rank)
        exchange()
        elapsed = DP.max_over_ranks(time.perf_counter() - t0, world, cpu)
        out = None
        if rank == 0:
            B = args.batch or WORKLOADS[args.workload]['B']
            out = {'metric': METRIC, 'value': None, 'unit': 'captions/s', 'n_gpus': world, 'steps': args.steps,
                   'warmup': args.warmup, 'ms_per_step': round(elapsed * 1e3, 3), 'higher_is_better': True,
                   'scaling': 'strong' if args.strong else 'weak', 'vs_baseline': None, 'dtype': 'f32',
                   'data': 'synthetic', 'selftest': True,
                   'rccl_ranks': torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
                   'config': {'workload': 'launcher self-test (no GPU work)', 'captions_per_gpu': B}}
            guard.line = out
        if world > 1 and not args.strong:     # an optional leg, with the real run's discipline
            guard.begin_optional('strong', 60.0)
            try:
                if fail and where == 'strong':
                    raise RuntimeError('injected failure in the strong leg')
                if fail and where == 'strong-hang':
                    guard.deadline = time.monotonic() + 1.0
                    time.sleep(3600)
                exchange()
            except Exception as e:      # noqa: BLE001
                guard.optional_failed('strong', '%s: %s' % (type(e).__name__, e))
            guard.end_optional()
            if rank == 0:
                out['strong'] = {'value': None, 'scaling': 'strong'}
        if rank == 0:
            guard.emit(out)
    except BaseException as e:      # noqa: BLE001
        guard.fatal(e)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        guard.close()
        torch.distributed.destroy_process_group()


def run_decode(args, rank, world, dev):
    """--workload c5 (BASELINE configs[4]): beam=5 sample_beam, greedy sample and the self-critical RL step
    (train_rl.py:160-203) at M=4, L=196, D=2048, B=128 per GPU; replicas only (no collective on a decode path)."""
    import torch
    import recurrent_fusion_network_amd as R
    from recurrent_fusion_network_amd import parallel as DP
    w = dict(WORKLOADS['c5'])
    B = args.batch or w['B']
    cfg = make_cfg(w)
    model = R.RecurrentFusionModel(cfg).to(dev)
    seeded_weights_(model, 100)
    if args.gemm == 'bf16x3':
        import recurrent_fusion_network_amd._native as N
        model.gemm_flags |= N.GEMM_OPT_BF16X3
    import recurrent_fusion_network_amd._native as N_
    model.path_flags |= int(args.persist) & (N_.PATH_OPT_PERSIST_ALL | N_.PATH_OPT_DEEP_CELLS | N_.PATH_OPT_NO_SMALL_TILES |
                                              N_.PATH_OPT_SHARED_SMALL_TILES | N_.PATH_OPT_DEC_UNHOISTED)
    unhoisted = bool(model.path_flags & N_.PATH_OPT_DEC_UNHOISTED)    # A/B hook; the beam loop has no three-launch form
    fc, att, labels, masks, top = synthetic_inputs(cfg, B, 100 + rank, dev)
    rl_crit = R.ReviewNetRewardCriterion(cfg)
    opt = R.FusedClampAdam(model, lr=5e-5, weight_decay=0.0, grad_clip=1.0)

    t_beam, t_greedy, t_rl = decode_legs(model, opt, rl_crit, cfg, (fc, att, labels, masks, top), B, dev, world, args.steps,
                                         args.warmup, beam=not unhoisted)
    if rank != 0:
        return
    fwd_flops = (train_step_flops(cfg, B) / 7.667 * 3.680)    # forward-only share (3.680 of 7.667 TF at C3)
    out = {'metric': 'images/sec (beam=5 sample_beam eval) at B=128, M=4, L=196, D=2048, seq=16',
           'value': round(world * B / t_beam, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': round(t_beam * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
           'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
           'config': {'workload': 'C5: beam=5 sample_beam (stages I/II once per image + 16 device-resident beam steps) '
                                  'at M=4, L=196, D=2048, B=%d per GPU; also greedy sample and the self-critical RL step' % B,
                      'images_per_gpu': B, 'parallelism': 'replicas only'},
           'modes': {'beam5_images_per_s': round(world * B / t_beam, 2), 'beam5_ms': round(t_beam * 1e3, 3),
                     'greedy_images_per_s': round(world * B / t_greedy, 2), 'greedy_ms': round(t_greedy * 1e3, 3),
                     'rl_step_images_per_s': round(world * B / t_rl, 2), 'rl_step_ms': round(t_rl * 1e3, 3)}}
    secs, flops = time_dominant_kernel(model, att, reps=5)
    achieved = flops / secs / 1e12
    mult, peak = (6, BF16_MFMA_PEAK_TFLOPS) if args.gemm == 'bf16x3' else (1, FP32_MFMA_PEAK_TFLOPS)
    if args.gemm == 'bf16x3':
        out['dtype'] = 'f32 (stage-I projections as 3 bf16 planes x 6 MFMA products, f32 accumulate; the rest exact f32)'
        out['config']['gemm'] = 'bf16x3'
    out['roofline'] = {'bound': 'mfma', 'achieved': round(mult * achieved, 2), 'peak': peak, 'unit': 'TFLOP/s',
                       'frac': round(mult * achieved / peak, 4), 'traffic': None,
                       'kernel': 'grouped att_2_att_h projection (%.3f TFLOP of f32 product, %.3f ms per launch)' % (flops / 1e12, secs * 1e3),
                       'greedy_frac': round(fwd_flops / t_greedy / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                       'beam5_frac': round(fwd_flops / t_beam / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)}
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(cfg, min(args.cpu_sample, 16), 100, mode='greedy')
    flush_c_stdio()
    print(json.dumps(out), flush=True)


def decode_legs(model, opt, rl_crit, cfg, inputs, B, dev, world, steps, warmup, beam=True):
    """BASELINE configs[4] on `model`: seconds per call of beam = 5 sample_beam, greedy sample and the self-critical RL step
    (train_rl.py:160-203: multinomial sample with grad, greedy baseline, reward criterion, backward, clamp + Adam)."""
    import torch
    from recurrent_fusion_network_amd import parallel as DP
    fc, att, labels, masks, top = inputs

    def beam5():
        model.eval()
        with torch.no_grad():
            return model.sample(fc, att, {'beam_size': 5})

    def greedy():
        model.eval()
        with torch.no_grad():
            return model.sample(fc, att, {'sample_max': 1})

    def rl_step():
        model.train()
        opt.zero_grad()
        seq, lp, lp_all, reason = model.sample(fc, att, {'sample_max': 0})
        with torch.no_grad():
            model.eval()
            model.sample(fc, att, {'sample_max': 1})
            model.train()
        reward = torch.randn(B, 1, device=dev).expand(B, seq.size(1)).contiguous()   # CIDEr-D scoring is out of scope
        rl_crit(lp, seq, reward, lp_all, 0.01, reason, top, 1.0, None, cfg).backward()
        opt.step()

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(fn):
        for _ in range(warmup):
            fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        fence()
        return DP.max_over_ranks(time.perf_counter() - t0, world, dev) / steps

    was_training = model.training
    try:
        return (timed(beam5) if beam else float('nan')), timed(greedy), timed(rl_step)
    finally:
        model.train(was_training)


# forward-only algorithmic FLOP of a decode at the C3 shape = prefix (stages I/II) 3.59 TF per 256 images +
# 5.3 GF per 256 decoder rows and decoded step; the XE train step is 7.667 TF per 256 captions, of which 3.680 forward.
def decode_flops(images, rows, steps=17):
    return 3.59e12 * images / 256.0 + 5.3e9 * rows / 256.0 * steps


def secondary_legs(args, dev, model, opt, start, R, N):
    """BASELINE configs 2 and 5 beside the headline, in the SAME process and JSON line (`secondary`), so that the driver's
    run times them too: `--workload c2` (eager and replayed from a HIP graph) on a fresh C2 model, and config 5 (beam = 5
    sample_beam, greedy sample, the self-critical RL step at B = 128) on the headline's own model restored to its initial
    weights (C5 has the C3 architecture).  Each entry carries the roofline it is priced against; a failing entry reports its
    error, the headline line stands.  ~10 s."""
    import torch
    out = {}

    def guarded(name, fn):
        try:
            out[name] = fn()
        except Exception as e:      # noqa: BLE001
            out[name] = {'error': '%s: %s' % (type(e).__name__, e)}
        torch.cuda.synchronize()

    def c2():
        w = WORKLOADS['c2']
        B = w['B']
        cfg = make_cfg(w)
        m2 = R.RecurrentFusionModel(cfg).to(dev)
        seeded_weights_(m2, 100)
        m2.train()
        crit = R.ReviewNetEnsembleCriterion(cfg)
        o2 = R.FusedClampAdam(m2, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=1.0)
        fc, att, labels, masks, top = synthetic_inputs(cfg, B, 100, dev)

        def eager():
            o2.zero_grad()
            log_prob, top_pred = m2(fc, att, labels)
            loss = crit(log_prob, labels[:, 1:], masks[:, 1:], top_pred, top, 1.0)
            loss.backward()
            o2.step()
            return loss

        def timed(fn, n, settle_s):
            ts = time.perf_counter()
            while time.perf_counter() - ts < settle_s:
                fn()
                torch.cuda.synchronize()
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n

        n = 20
        t_eager = timed(eager, n, 0.5)
        launches = count_launches(eager)
        m2.grad_ready_hook = None
        from recurrent_fusion_network_amd.graphed import GraphedTrainStep
        g = GraphedTrainStep(m2, crit, o2, fc, att, labels, masks, top)
        t_graph = timed(g, n, 0.5)
        flops = w['step_tflop'] * 1e12
        return {'workload': 'C2: RecurrentFusionModel XE train step (zero_grad+fwd+criterion+bwd+clamp+Adam), M=2 encoders, '
                            'L=49, D=512, B=%d, seq=16 (17 decoder steps)' % B,
                'ms_per_step': round(t_eager * 1e3, 3), 'value': round(B / t_eager, 1), 'unit': 'captions/s', 'steps': n,
                'graph_ms_per_step': round(t_graph * 1e3, 3), 'graph_value': round(B / t_graph, 1),
                'launches': launches,
                'step_frac': round(flops / t_eager / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                'graph_step_frac': round(flops / t_graph / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                'roofline': {'bound': 'mfma', 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'step_tflop': w['step_tflop'],
                             'achieved': round(flops / t_graph / 1e12, 2), 'frac': round(flops / t_graph / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                             'note': 'whole step (SURVEY 8d: 0.1706 TF) against the f32-MFMA peak, graph-replayed; a latency-regime '
                                     'workload: ~320 dependent launches of 5-15 us'}}

    def c5():
        B = WORKLOADS['c5']['B']
        cfg = make_cfg(WORKLOADS['c5'])
        if start is not None:
            opt.restore(start)              # bench.py --workload c5's weights: the seeded initial ones
        inputs = synthetic_inputs(cfg, B, 100, dev)
        rl_crit = R.ReviewNetRewardCriterion(cfg)
        t_beam, t_greedy, t_rl = decode_legs(model, opt, rl_crit, cfg, inputs, B, dev, 1, 5, 2)
        S1 = cfg.seq_length + 1
        f_greedy, f_beam = decode_flops(B, B, S1), decode_flops(B, 5 * B, S1)
        # RL step: sampled pass with grad = the train step's forward and backward on B rows, + the greedy baseline decode
        f_rl = 7.667e12 * B / 256.0 + f_greedy

        def entry(t, flops, what):
            return {'ms_per_step': round(t * 1e3, 3), 'value': round(B / t, 1), 'unit': 'images/s', 'steps': 5,
                    'step_frac': round(flops / t / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                    'roofline': {'bound': 'mfma', 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'step_tflop': round(flops / 1e12, 4),
                                 'achieved': round(flops / t / 1e12, 2), 'frac': round(flops / t / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                                 'flop_count': what}}
        return {
            'c5_beam5': entry(t_beam, f_beam, 'SURVEY 8d forward-only: prefix 3.59 TF x B/256 + 5.3 GF x (5B)/256 x %d decoder steps' % S1),
            'c5_greedy': entry(t_greedy, f_greedy, 'SURVEY 8d forward-only: prefix 3.59 TF x B/256 + 5.3 GF x B/256 x %d decoder steps' % S1),
            'c5_rl': entry(t_rl, f_rl, 'multinomial sample with grad + backward = the XE train step 7.667 TF x B/256, + the greedy '
                                        'baseline decode (forward-only figure above); reward criterion and Adam are not FLOP-bound'),
        }

    guarded('c2', c2)
    try:
        out.update(c5())
    except Exception as e:      # noqa: BLE001
        out['c5'] = {'error': '%s: %s' % (type(e).__name__, e)}
    torch.cuda.synchronize()
    out['note'] = ('BASELINE configs 2 and 5 timed in this process after the headline (B = 64 / B = 128 images on this GPU); '
                   'c5_* run on the headline model restored to its seeded initial weights; workload-level evidence: profiles/r06_secondary.jsonl')
    return out


def count_launches(fn):
    """Kernel launches of one call of `fn`, counted by the profiler's device-activity records (None when unavailable)."""
    try:
        import torch
        from torch.profiler import ProfilerActivity, profile
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            fn()
            torch.cuda.synchronize()
        n = 0
        for ev in prof.events():
            if getattr(ev, 'device_type', None) is not None and 'cuda' in str(ev.device_type).lower():
                name = ev.name.lower()
                if 'memcpy' in name or 'memset' in name:
                    continue
                n += 1
        return n or None
    except Exception:      # noqa: BLE001
        return None


def flush_c_stdio():
    """RCCL prints its version banner through C stdio when the communicator is created; with stdout redirected that buffer
    is only flushed at exit, i.e. AFTER the result line Python printed.  A driver that reads the last line of stdout must find
    the JSON line there, so the C buffers are flushed before anything this script prints."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:      # noqa: BLE001
        pass


def _proc_start_ticks(pid):
    """Start time of process `pid` in clock ticks since boot (field 22 of /proc/<pid>/stat): tells two launchers that
    happened to get the same pid apart."""
    try:
        with open('/proc/%d/stat' % pid) as f:
            return f.read().rsplit(')', 1)[1].split()[19]
    except (OSError, IndexError):
        return '0'


def run_guard_path(env=None):
    """The flag file of THIS launch.  bench.py's own launcher hands every rank a fresh RFN_BENCH_RUN_ID; under an external
    launcher (torch.distributed.run's static rendezvous gives every job the run id 'none') the ranks' common parent -- its
    pid AND its start time -- plus the rendezvous port name the launch, so a flag left behind by an earlier failed run can
    never be read as this run's, and nobody has to delete anything at start-up (rank 0 used to unlink a leftover
    in its constructor, racing the peers' watcher threads both ways)."""
    import tempfile
    env = os.environ if env is None else env
    run_id = env.get('RFN_BENCH_RUN_ID')
    if not run_id:
        ppid = os.getppid()
        run_id = '%d_%s_%s' % (ppid, _proc_start_ticks(ppid), env.get('MASTER_PORT', '0'))
    return os.path.join(tempfile.gettempdir(), 'rfn_bench_%s.flag' % run_id)


class RunGuard:
    """What keeps the one driver-run line of an N-rank job from hanging or getting lost.

    A rank that raises must never enter a collective its peers are not in: they sit in a gradient all-reduce and everybody
    would wait for the NCCL watchdog (10 min).  So ranks agree through a flag FILE (one node; named by a per-launch nonce,
    `run_guard_path`), polled by a daemon thread on every rank -- a thread, because the main thread of a healthy rank
    may be blocked inside a collective or a device synchronisation (both release the GIL).

    * a failure before the headline is measured: `fatal(exc)` -- traceback, flag, os._exit(1).  Peers see the flag and
      exit 1 at once (torchrun tears the job down as soon as the first child is gone; the flag covers other launchers).
    * a failure or an overrun inside an OPTIONAL leg (the strong-scaling and bf16x3 re-timings): the headline is already
      in rank 0's hands (`self.line`), so rank 0 prints it with `{leg: {"error": ...}}` and every rank exits 0.
    """
    POLL_S = 0.25

    def __init__(self, rank, world):
        import threading
        self.rank, self.world = rank, world
        self.path = run_guard_path()
        self.line = None            # rank 0: the result line as far as it is known (set once the headline leg is done)
        self.leg = None             # the optional leg in progress
        self.deadline = None
        self.lock = threading.Lock()
        self.closed = False
        if world > 1:
            threading.Thread(target=self._watch, name='rfn-bench-guard', daemon=True).start()

    def _flag(self, text):
        try:
            with open(self.path, 'a') as f:
                f.write(text.replace('\n', ' ')[:400] + '\n')
        except OSError:
            pass

    def _read(self):
        try:
            with open(self.path) as f:
                return f.readline().strip() or None       # created but not yet written: nothing to act on
        except OSError:
            return None

    def _watch(self):
        while not self.closed:
            time.sleep(self.POLL_S)
            msg = self._read()
            if msg is None and self.deadline is not None and time.monotonic() > self.deadline:
                msg = 'optional\t%s\trank %d: the leg overran its deadline' % (self.leg, self.rank)
                self._flag(msg)
            if msg is None or self.closed:
                continue
            if msg.startswith('optional\t'):
                _, leg, text = (msg.split('\t', 2) + ['', ''])[:3]
                self._finish_without(leg, text)
                return
            sys.stderr.write('bench.py rank %d: leaving, a peer failed: %s\n' % (self.rank, msg))
            sys.stderr.flush()
            os._exit(1)

    def _finish_without(self, leg, text):
        """Optional leg `leg` is abandoned job-wide: rank 0 prints the line it holds, everybody exits 0."""
        waited = 0.0
        while self.rank == 0 and self.line is None and waited < 300.0:
            time.sleep(self.POLL_S)          # a peer failed while rank 0 was still pricing the headline's roofline
            waited += self.POLL_S
        with self.lock:
            if self.closed:
                return
            self.closed = True
            if self.rank == 0 and self.line is not None:
                self.line.setdefault(leg or 'optional', {'error': text})
                flush_c_stdio()
                print(json.dumps(self.line), flush=True)
            sys.stderr.write('bench.py rank %d: optional leg %s abandoned (%s)\n' % (self.rank, leg, text))
            sys.stderr.flush()
            os._exit(0 if (self.rank != 0 or self.line is not None) else 1)

    def fatal(self, exc):
        import traceback
        traceback.print_exception(type(exc), exc, exc.__traceback__)
        sys.stderr.write('bench.py rank %d: failed at t=%.3f before the result line; not entering any collective\n'
                         % (self.rank, time.time()))
        sys.stderr.flush()
        if self.world > 1:
            self._flag('fatal\trank %d: %s: %s' % (self.rank, type(exc).__name__, exc))
        os._exit(1)

    def begin_optional(self, leg, seconds):
        self.leg, self.deadline = leg, time.monotonic() + seconds

    def end_optional(self):
        self.leg, self.deadline = None, None

    def optional_failed(self, leg, err):
        """This rank raised inside optional leg `leg`.  With peers around nobody may wait for it: flag and leave."""
        text = 'rank %d: %s' % (self.rank, err)
        self._flag('optional\t%s\t%s' % (leg, text))
        first = (self._read() or '').split('\t', 2)         # the FIRST failure is the cause; later ones are its echoes
        if len(first) == 3 and first[0] == 'optional':
            leg, text = first[1], first[2]
        self._finish_without(leg, text)

    def close(self):
        self.closed = True
        if self.world > 1 and self.rank == 0:
            try:
                os.unlink(self.path)
            except OSError:
                pass

    def emit(self, line):
        with self.lock:
            if self.closed:
                return
            self.closed = True
            flush_c_stdio()
            print(json.dumps(line), flush=True)


def run_rank(args):
    import torch
    import recurrent_fusion_network_amd as R
    from recurrent_fusion_network_amd import parallel as DP

    rank, world, local = DP.init_from_env('nccl')
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    dev = torch.device('cuda', int(os.environ.get('RFN_DEVICE_INDEX', local)))   # test hook: ranks sharing one GPU
    torch.cuda.set_device(dev)
    guard = RunGuard(rank, world)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()          # creates the communicator now: its banner goes out before any result
        flush_c_stdio()
    try:
        if args.workload == 'c5':
            run_decode(args, rank, world, dev)
        else:
            run_train(args, rank, world, dev, R, DP, guard)
    except BaseException as e:      # noqa: BLE001  (SystemExit included: whatever it is, the peers must not wait for us)
        # NOT a collective: the peers are inside a gradient all-reduce this rank will never join
        guard.fatal(e)
    # every rank got here: the line is out, the collectives below are matched
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        guard.close()
        torch.distributed.destroy_process_group()
    # nothing may follow the result line on stdout (library chatter at exit, late C stdio buffers): whatever is still written
    # to file descriptor 1 goes to the null device.  The process ends normally -- a profiler attached to it (rocprofv3 writes
    # its tables from exit handlers) must get to run them.
    sys.stdout.flush()
    flush_c_stdio()
    try:
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    except OSError:
        pass


def run_train(args, rank, world, dev, R, DP, guard):
    import torch
    import recurrent_fusion_network_amd._native as N
    w = dict(WORKLOADS[args.workload])
    B = args.batch or w['B']
    global_B = B * world
    if args.strong:                      # the workload batch is the global batch: each rank takes its row shard
        global_B = B
        lo, hi = DP.shard_rows(global_B, rank, world)
        B = hi - lo
    cfg = make_cfg(w)
    cfg.use_label_smoothing = int(args.label_smoothing)
    cfg.drop_prob_lm = float(args.drop_lm)
    torch.manual_seed(100 + rank)            # opts.py:178 default seed, + rank (train.py:23)
    in_group = torch.distributed.is_initialized()

    model = R.RecurrentFusionModel(cfg).to(dev)
    seeded_weights_(model, 100)              # identical replicas on every rank
    model.train()
    model.ss_prob = float(args.ss_prob)
    x3 = args.gemm == 'bf16x3'
    if x3:
        model.gemm_flags |= N.GEMM_OPT_BF16X3
    if args.lds_lean:
        model.gemm_flags |= N.GEMM_OPT_LDS_LEAN
    model.path_flags |= int(args.persist) & (N.PATH_OPT_PERSIST_ALL | N.PATH_OPT_DEEP_CELLS | N.PATH_OPT_NO_SMALL_TILES |
                                              N.PATH_OPT_SHARED_SMALL_TILES | N.PATH_OPT_DEC_UNHOISTED)
    if args.micro_batches >= 0 and hasattr(model, 'micro_batches'):
        model.micro_batches = args.micro_batches
    crit = R.ReviewNetEnsembleCriterion(cfg)
    shard = (rank, world) if (args.shard_optimizer and (world > 1 or in_group)) else None     # a one-rank group (RFN_FORCE_DIST=1) runs the RCCL branches too
    opt = R.FusedClampAdam(model, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5, grad_clip=1.0, shard=shard)
    inputs = synthetic_inputs(cfg, B, 100 + rank, dev)
    if args.strong:
        inputs = tuple(inputs) + (DP.shard_loss_scale(B, global_B, world),)

    if args.graph and (in_group or (not x3 and not args.no_alt_line)):
        raise SystemExit('--graph: single GPU, one GEMM mode per run (add --no-alt-line)')
    sync = DP.GradSync(model, world, shard_optimizer=opt if shard else None)     # per-bucket async exchange, overlapped with the rest of backward
    if args.graph:
        model.grad_ready_hook = None     # no exchange to overlap on one GPU
    if args.overlap_update:
        if in_group or args.graph:
            raise SystemExit('--overlap-update: single GPU, eager steps')
        sync = DP.OverlappedUpdate(model, opt)
    start = opt.snapshot() if (not x3 and not args.no_alt_line) else None     # the bf16x3 leg restarts from here

    trace = [] if args.trace_steps else None
    fail_rank = os.environ.get('RFN_BENCH_FAIL_RANK') == str(rank)            # test hooks: a rank that dies mid-run
    fail_where = os.environ.get('RFN_BENCH_FAIL_WHERE', 'headline')
    counts = {}

    def mark(row):
        if trace is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            row.append(e)

    graphed = {}
    in_step = {}                # per timed step and encoder: ms of the projection / weight-gradient launch (headline leg)

    def step(inp, leg='headline'):
        if args.graph:                      # one captured graph per input set (its buffers are the graph's static inputs)
            if id(inp) not in graphed:
                from recurrent_fusion_network_amd.graphed import GraphedTrainStep
                graphed[id(inp)] = GraphedTrainStep(model, crit, opt, *inp[:5])
            counts[leg] = counts.get(leg, 0) + 1
            return graphed[id(inp)]()
        fc, att, labels, masks, top = inp[:5]
        loss_scale = inp[5] if len(inp) > 5 else 1.0      # uneven shards weigh their local mean (parallel.shard_loss_scale)
        row = []
        mark(row)
        opt.zero_grad()
        if args.fused_loss:
            loss, top_pred = model.forward_loss(fc, att, labels, masks, top, crit, 1.0)
        else:
            log_prob, top_pred = model(fc, att, labels)
            loss = crit(log_prob, labels[:, 1:], masks[:, 1:], top_pred, top, 1.0)
        if loss_scale != 1.0:
            loss = loss * loss_scale
        mark(row)
        loss.backward()
        mark(row)
        scale = sync.finish()              # 1/world: applied before the clamp inside the fused update
        opt.step(grad_scale=scale)
        mark(row)
        if trace is not None:
            trace.append((row, time.perf_counter()))
        counts[leg] = counts.get(leg, 0) + 1
        if fail_rank and leg == fail_where and counts[leg] == 2:
            torch.cuda.synchronize()
            raise RuntimeError('injected failure on rank %d in the %s leg (RFN_BENCH_FAIL_RANK)' % (rank, leg))
        return loss

    def dump_trace(label, n):
        if trace is None or rank != 0:
            return
        torch.cuda.synchronize()
        rows = trace[-n:]
        for i, (r, h) in enumerate(rows):
            nxt = r[3].elapsed_time(rows[i + 1][0][0]) if i + 1 < len(rows) else 0.0
            sys.stderr.write('%s step %2d: fwd %.1f bwd %.1f update %.1f gap %.2f | host return at +%.1f ms\n' % (
                label, i, r[0].elapsed_time(r[1]), r[1].elapsed_time(r[2]), r[2].elapsed_time(r[3]), nxt,
                (h - rows[0][1]) * 1e3))

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def allocs():
        st = torch.cuda.memory_stats(dev)
        return st.get('num_device_alloc', 0) + st.get('num_device_free', 0)

    def settle(inp, leg, seconds=0.0, n_steps=None):
        """Untimed steps before the warm-up: for `seconds` (the same number of steps on every rank: the slowest clock
        decides), or exactly `n_steps` of them."""
        n = 0
        if n_steps is not None:
            keep = None
            for _ in range(n_steps):
                keep = step(inp, leg)      # noqa: F841  held like the timed loop holds it: same allocator pattern
                n += 1
            return n
        if seconds > 0:
            ts = time.perf_counter()
            keep = None
            while True:
                keep = step(inp, leg)      # noqa: F841
                n += 1
                torch.cuda.synchronize()
                # every rank leaves after the same step: continue while ANY rank's clock is still inside the window
                if DP.max_over_ranks(1.0 if time.perf_counter() - ts < seconds else 0.0, world, dev) == 0.0:
                    break
        return n

    def time_leg(inp, leg, seconds=0.0, n_settle=None):
        """settle + W warm-up steps + EXACTLY K timed steps bracketed by barrier + synchronize.  Local times only: the
        max over ranks is taken by the caller, outside any try block."""
        n = settle(inp, leg, seconds, n_settle)
        loss = None
        for _ in range(args.warmup):
            loss = step(inp, leg)
        # the dominant launches INSIDE the timed steps: HIP events recorded by the path itself around every encoder's
        # hoisted projection and weight-gradient launch (rfn_dims.probe_events), one event set per timed step, read
        # after the region's own fence -- nothing is synchronised for them
        probes = None
        if leg == 'headline' and not args.graph:
            n_ev = 4 * len(w['enc'])
            probes = [[torch.cuda.Event(enable_timing=True) for _ in range(n_ev)] for _ in range(args.steps)]
            for evs in probes:
                for e in evs:
                    e.record()         # creates the HIP event behind it
        fence()
        sync.record = True
        a0, t0, c0 = allocs(), time.perf_counter(), time.thread_time()
        for k in range(args.steps):
            if probes is not None:
                model.set_probe_events(probes[k])
            loss = step(inp, leg)
        fence()
        local = time.perf_counter() - t0
        model.set_probe_events(None)
        if probes is not None:
            M_ = len(w['enc'])
            in_step['fwd_ms'] = [[evs[2 * i].elapsed_time(evs[2 * i + 1]) for i in range(M_)] for evs in probes]
            in_step['wgrad_ms'] = [[evs[2 * M_ + 2 * i].elapsed_time(evs[2 * M_ + 2 * i + 1]) for i in range(M_)] for evs in probes]
        host_cpu = (time.thread_time() - c0) / max(1e-9, local)      # share of the timed region the launching thread was
        sync.record = False                                          # on a CPU (1.0 = never descheduled)
        exposed, per_bucket, how = sync.exposed_ms()
        return dict(local=local, loss=float(loss.detach()), allocs=int(allocs() - a0), host_cpu=host_cpu, settle_n=n,
                    exposed=exposed, per_bucket=per_bucket, exposed_how=how)

    def exposed_fields(r):
        """exposed_ms: mean time per step the compute stream (or, on a host-blocking backend, the host) waited for the
        gradient exchange, this rank; max over ranks beside it."""
        if r['exposed'] is None:
            return {'exposed_ms': None}
        worst = DP.max_over_ranks(r['exposed'], world, dev)
        return {'exposed_ms': round(worst, 3), 'exposed_ms_rank0': round(r['exposed'], 3),
                'exposed_ms_by_bucket_rank0': {k: round(v, 3) for k, v in r['per_bucket'].items()},
                'exposed_measured_by': r['exposed_how']}

    # ---- the headline leg: nothing optional runs before this is measured and in rank 0's hands -------------------------
    head = time_leg(inputs, 'headline', seconds=args.settle)
    elapsed = DP.max_over_ranks(head['local'], world, dev)
    head_x = exposed_fields(head)
    dump_trace('exact' if not x3 else 'bf16x3', args.steps)
    final_loss = head['loss']
    digest = None
    if args.digest:
        import hashlib
        import struct
        hsh = hashlib.sha256(struct.pack('<f', final_loss))
        opt.wait_params()               # a sharded / overlapped update may still be writing parameters on another stream
        # over the PARAMETERS, in name order (the flat buckets also hold the 16-B padding between them, which is no state)
        ps = [p_.detach().double() for _, p_ in sorted(model.named_parameters())]
        # ... and the Adam moments, bucket by bucket (gathered from the ranks when the update is sharded: a collective)
        # -- parameter by parameter: the 16-B alignment gaps between parameters in a flat bucket hold whatever the gradient
        # buffer's allocation held, which is no state either
        sd = opt.state_dict()['buckets']
        for k in sorted(sd):
            params_k, offs_k, _ = model.bucket_layout(k)
            ps += [sd[k][mv][o:o + p_.numel()].double() for p_, o in zip(params_k, offs_k) for mv in ('m', 'v')]
        sums = torch.stack([q.sum() for q in ps] + [(q * q).sum() for q in ps]).cpu().tolist()
        hsh.update(struct.pack('<%dd' % len(sums), *sums))
        digest = hsh.hexdigest()[:16]
    ms = elapsed / args.steps * 1e3
    out = None
    if rank == 0:
        M = len(w['enc'])
        uniform = all(e == w['enc'][0] for e in w['enc'])
        shape = ('M=%d encoders, L=%d, D=%d' % (M, w['enc'][0][0], w['enc'][0][1]) if uniform else
                 'M=%d encoders (L,D,fc)=%s' % (M, ','.join('(%d,%d,%d)' % e for e in w['enc'])))
        extras = ''.join([', label smoothing 0.1' if args.label_smoothing else '',
                          ', drop_prob_lm %.2g' % args.drop_lm if args.drop_lm else '',
                          ', ss_prob %.2g' % args.ss_prob if args.ss_prob else ''])
        out = {
            'metric': METRIC, 'value': round(global_B * args.steps / elapsed, 2), 'unit': 'captions/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True,
            'scaling': 'strong' if args.strong else 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'settle_s': args.settle, 'settle_steps': head['settle_n'], 'host_thread_cpu_share': round(head['host_cpu'], 3),
            'device_mallocs_frees_in_timed_region': head['allocs'],
            'rccl_ranks': torch.distributed.get_world_size() if in_group else 1,
            'dist_backend': torch.distributed.get_backend() if in_group else None,
            'config': {'workload': '%s: RecurrentFusionModel XE train step (zero_grad+fwd+criterion+bwd+clamp+Adam), '
                                   '%s, R=A=E=512, T1=T2=8, K=1000, V+1=9488, seq=16 (17 decoder steps)%s'
                                   % (w['desc'], shape, extras),
                       'captions_per_gpu': B, 'global_batch': global_B,
                       'parallelism': 'dp%d (batch sharded, RCCL all-reduce of grads)' % world if world > 1 else 'single GPU',
                       'micro_batches': int(getattr(model, 'micro_batches', 1) or 1),
                       'final_loss': round(final_loss, 4), 'updates': head['settle_n'] + args.warmup + args.steps,
                       'hip_graph': bool(args.graph), 'shard_optimizer': bool(shard), 'overlap_update': bool(args.overlap_update),
                       'fused_loss': bool(args.fused_loss),
                       'gemm_flags': int(model.gemm_flags), 'digest': digest},
        }
        if in_group:
            out.update(head_x)
        if x3:
            out['dtype'] = 'f32 (the two long products as 3 bf16 planes x 6 MFMA products, f32 accumulate; the rest exact f32)'
            out['config']['gemm'] = 'bf16x3'
        # roofline of the dominant kernel, timed live with HIP events on the launch stream
        att = inputs[1]
        secs_alone, flops = time_dominant_kernel(model, att, reps=5)
        standalone = flops / secs_alone / 1e12
        # `achieved` = the projection launches INSIDE the timed steps (all encoders, all K steps): sum of their algorithmic
        # FLOP / sum of their durations.  The stand-alone figure (5 back-to-back launches of encoder 0's projection, the
        # best case) stays beside it as frac_standalone.
        A_, T1_ = model.att_hid_size, model.num_review_steps_0
        enc_flops = [2.0 * B * e[0] * e[1] * A_ * T1_ for e in w['enc']]
        secs, achieved, in_step_note = secs_alone, standalone, 'stand-alone launches (no in-step events: --graph)'
        wgrad_tf = None
        if in_step.get('fwd_ms') and min(min(r) for r in in_step['fwd_ms']) > 1e-3:
            tot_ms = sum(sum(r) for r in in_step['fwd_ms'])
            n_l = sum(len(r) for r in in_step['fwd_ms'])
            achieved = sum(enc_flops) * len(in_step['fwd_ms']) / (tot_ms * 1e-3) / 1e12
            secs = tot_ms * 1e-3 / n_l
            flops = sum(enc_flops) / len(enc_flops)
            in_step_note = 'mean of the %d projection launches inside the %d timed steps' % (n_l, len(in_step['fwd_ms']))
            wg_ms = sum(sum(r) for r in in_step['wgrad_ms'])
            wgrad_tf = sum(enc_flops) * len(in_step['wgrad_ms']) / (wg_ms * 1e-3) / 1e12
        # HBM-side bytes per launch of that kernel, from the rocprofv3 --pmc passes (null when the kernel source has
        # changed since they were taken)
        traffic = pmc_traffic(args.workload, B == w['B'])
        traffic_x3 = pmc_traffic(args.workload + '_bf16x3', B == w['B'])
        L0, D0 = w['enc'][0][0], w['enc'][0][1]
        if x3:      # priced in bf16 MFMA FLOP (6 plane products per f32 product) against the bf16 peak
            roof = {'bound': 'mfma', 'achieved': round(6 * achieved, 2), 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(6 * achieved / BF16_MFMA_PEAK_TFLOPS, 4), 'traffic': traffic_x3,
                    'traffic_unit': 'bytes per launch (2*FETCH_SIZE + WRITE_SIZE)',
                    'f32_equivalent_tflops': round(achieved, 2),
                    'algorithmic_bytes': int(6 * (B * L0 * D0 + 8 * 512 * D0) + 4 * B * L0 * 8 * 512),
                    'kernel': 'x3_gemm_k (grouped att_2_att_h projection of encoder 0 on plane images, %.3f TFLOP of f32 '
                              'product = %.3f TFLOP of bf16 MFMA per launch, %.3f ms per launch)'
                              % (flops / 1e12, 6 * flops / 1e12, secs * 1e3)}
        else:
            roof = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), 'traffic': traffic,
                    'traffic_unit': 'bytes per launch (2*FETCH_SIZE + WRITE_SIZE)',
                    'algorithmic_bytes': int(4 * (B * L0 * D0 + 8 * 512 * D0 + B * L0 * 8 * 512)),
                    'kernel': 'rfn_gemm_kernel NT big tile (grouped att_2_att_h projection of one encoder, '
                              '%.3f TFLOP per launch, %.3f ms per launch: %s)' % (flops / 1e12, secs * 1e3, in_step_note),
                    'frac_standalone': round(standalone / FP32_MFMA_PEAK_TFLOPS, 4),
                    'standalone_ms': round(secs_alone * 1e3, 3)}
            if wgrad_tf is not None:
                roof['wgrad_frac'] = round(wgrad_tf / FP32_MFMA_PEAK_TFLOPS, 4)     # the TN twin (dW = dP^T X), in step
                roof['in_step_ms'] = {'projection': [round(sum(r[i] for r in in_step['fwd_ms']) / len(in_step['fwd_ms']), 3)
                                                     for i in range(len(w['enc']))],
                                      'weight_gradient': [round(sum(r[i] for r in in_step['wgrad_ms']) / len(in_step['wgrad_ms']), 3)
                                                          for i in range(len(w['enc']))]}
        out['roofline'] = roof
        # whole-step view against the same peak (SURVEY.md 8d algorithmic FLOP of one step)
        step_flops = (w['step_tflop'] * 1e12 * (B / w['B'])) if w['step_tflop'] else train_step_flops(cfg, B)
        out['roofline']['step_frac'] = round(step_flops / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)
        out['roofline']['step_tflop'] = round(step_flops / 1e12, 4)
        guard.line = out                   # from here on the headline cannot be lost to an optional leg

    # ---- optional legs: re-timings reported beside the headline, never as `value` ---------------------------------------
    # Each runs under try / except; with peers around, a rank that fails (or a leg that overruns) ends the job through the
    # guard -- rank 0 prints the line it holds -- because a failing rank cannot tell its peers through a collective.
    budget = 60.0 + 4.0 * (args.settle + (head['settle_n'] + 2 * args.warmup + 2 * args.steps) * ms * 1e-3)

    def optional(leg, fn):
        guard.begin_optional(leg, budget)
        err, res = None, None
        try:
            res = fn()
        except Exception as e:      # noqa: BLE001  (reported in the line, the headline stands)
            err = '%s: %s' % (type(e).__name__, e)
            sync.abandon()
            sync.record = False
            if world > 1:
                guard.optional_failed(leg, err)          # does not return
        guard.end_optional()
        return err, res

    # (1) the other reading of the metric: the SAME global batch one GPU runs (B captions), sharded over the ranks
    strong = None
    if in_group and not args.strong:
        lo, hi = DP.shard_rows(B, rank, world)
        if DP.max_over_ranks(1.0 if hi - lo < 1 else 0.0, world, dev) > 0.0:
            strong = {'error': 'a global batch of %d rows cannot be sharded over %d ranks' % (B, world)}
        else:
            shard_inputs = synthetic_inputs(cfg, hi - lo, 1100 + rank, dev) if world > 1 else inputs
            shard_inputs = tuple(shard_inputs) + (DP.shard_loss_scale(hi - lo, B, world),)
            err, r = optional('strong', lambda: time_leg(shard_inputs, 'strong', seconds=min(args.settle, 1.0)))
            if err:
                strong = {'error': err}
            else:
                s_elapsed = DP.max_over_ranks(r['local'], world, dev)
                s_ms = s_elapsed / args.steps * 1e3
                s_flops = (w['step_tflop'] * 1e12 * ((hi - lo) / w['B'])) if w['step_tflop'] else train_step_flops(cfg, hi - lo)
                strong = {'value': round(B * args.steps / s_elapsed, 2), 'unit': 'captions/s', 'ms_per_step': round(s_ms, 3),
                          'global_batch': B, 'captions_per_gpu': hi - lo, 'scaling': 'strong',
                          'device_mallocs_frees_in_timed_region': r['allocs'],
                          'step_frac': round(s_flops / (s_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)}
                strong.update(exposed_fields(r))
            del shard_inputs
        if rank == 0:
            out['strong'] = strong

    # (2) the same W + K steps with the two long products on the bf16 matrix cores, from the SAME
    # starting weights and optimizer state and after the same number of updates as the headline leg, so that the two
    # `final_loss` values are comparable
    alt = None
    if not x3 and not args.no_alt_line:
        def x3_leg():
            opt.restore(start)
            model.gemm_flags |= N.GEMM_OPT_BF16X3
            try:
                return time_leg(inputs, 'bf16x3', n_settle=head['settle_n'])
            finally:
                model.gemm_flags &= ~N.GEMM_OPT_BF16X3
        err, r = optional('bf16x3', x3_leg)
        if err:
            alt = {'error': err}
        else:
            alt_elapsed = DP.max_over_ranks(r['local'], world, dev)
            dump_trace('bf16x3', args.steps)
            alt = {'value': round(global_B * args.steps / alt_elapsed, 2), 'unit': 'captions/s',
                   'ms_per_step': round(alt_elapsed / args.steps * 1e3, 3), 'final_loss': round(r['loss'], 4),
                   'updates': r['settle_n'] + args.warmup + args.steps,
                   'final_loss_note': 'same start, same number of updates as config.final_loss',
                   'device_mallocs_frees_in_timed_region': r['allocs'],
                   'dtype': 'f32 (the two long products as 3 bf16 planes x 6 MFMA products, f32 accumulate; the rest exact f32)'}
            if in_group:
                alt.update(exposed_fields(r))
            if rank == 0:
                model.gemm_flags |= N.GEMM_OPT_BF16X3
                try:
                    secs_a, flops_a = time_dominant_kernel(model, inputs[1], reps=5)
                    alt['roofline'] = {'bound': 'mfma', 'achieved': round(6 * flops_a / secs_a / 1e12, 2),
                                       'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                       'frac': round(6 * flops_a / secs_a / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                                       'f32_equivalent_tflops': round(flops_a / secs_a / 1e12, 2),
                                       'kernel': 'x3_gemm_k, same projection on plane images, %.3f ms per launch' % (secs_a * 1e3)}
                    alt['roofline']['traffic'] = pmc_traffic(args.workload + '_bf16x3', B == w['B'])
                except Exception as e:      # noqa: BLE001
                    alt['roofline'] = {'error': '%s: %s' % (type(e).__name__, e)}
                finally:
                    model.gemm_flags &= ~N.GEMM_OPT_BF16X3
        if rank == 0:
            out['bf16x3'] = alt
    # (3) BASELINE configs 2 and 5 beside the headline (single GPU, the default run only)
    if (world == 1 and args.workload == 'c3' and not args.batch and not args.no_alt_line and not args.no_secondary and not x3
            and not args.graph and not args.persist):
        err, sec = optional('secondary', lambda: secondary_legs(args, dev, model, opt, start, R, N))
        out['secondary'] = {'error': err} if err else sec
    if rank != 0:
        return
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(cfg, args.cpu_sample, 100)
    guard.emit(out)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args, argv))     # nothing has touched torch or the GPU in this process
    if args.selftest_launch:
        selftest_rank(args)
    else:
        run_rank(args)


if __name__ == '__main__':
    main()

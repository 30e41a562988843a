"""bench.py as the driver runs it: `python bench.py --gpus N ...` with no launcher around it must start the N ranks
itself (fresh child processes, decided before torch or the GPU is touched), relay ONE JSON line and propagate a
failing rank as a non-zero exit code."""
import json
import os
import re
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(extra)
    return env


def _json_lines(stdout):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.startswith('{')]


def test_parent_decides_before_importing_torch():
    code = 'import sys; sys.path.insert(0, %r); import bench; assert "torch" not in sys.modules, "bench imports torch at module level"' % ROOT
    subprocess.run([sys.executable, '-c', code], check=True, env=_env(), timeout=120)


def test_two_ranks_self_launch_one_line_over_gloo():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--batch', '4', '--steps', '1', '--selftest-launch'],
                       env=_env(RFN_DIST_BACKEND='gloo'), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    out = lines[0]
    assert out['n_gpus'] == 2 and out['rccl_ranks'] == 2 and out['selftest'] is True
    assert out['metric'].startswith('captions/sec') and out['scaling'] == 'weak' and out['value'] is None


def _failed_at(stderr):
    """Wall-clock time at which the injected failure happened (RunGuard.fatal prints it)."""
    m = re.search(r'failed at t=([0-9.]+)', stderr)
    assert m, stderr[-2000:]
    return float(m.group(1))


def test_strong_flag_and_failing_rank_propagates():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '1', '--selftest-launch', '--strong'],
                       env=_env(RFN_DIST_BACKEND='gloo'), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and _json_lines(r.stdout)[0]['scaling'] == 'strong'
    # a rank that dies before the headline exists: no line, non-zero exit, and QUICKLY -- the failing rank must not enter a
    # collective (its peers are inside an all-reduce it will never join) and the peers must not wait for a watchdog
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '1', '--selftest-launch'],
                       env=_env(RFN_DIST_BACKEND='gloo', RFN_BENCH_FAIL_RANK='1'), capture_output=True, text=True,
                       timeout=300)
    done = time.time()
    assert r.returncode != 0 and not _json_lines(r.stdout)
    assert done - _failed_at(r.stderr) < 60.0


def test_eight_ranks_self_launch_over_gloo():
    """The N = 8 launch the driver will make, on CPU: 8 fresh ranks, one line, weak + strong legs."""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '8', '--steps', '1', '--selftest-launch'],
                       env=_env(RFN_DIST_BACKEND='gloo', OMP_NUM_THREADS='1'), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0]['n_gpus'] == 8 and lines[0]['rccl_ranks'] == 8 and lines[0]['strong'] == {'value': None, 'scaling': 'strong'}


@pytest.mark.parametrize('where', ['strong', 'strong-hang'])
def test_a_failing_optional_leg_cannot_lose_the_headline(where):
    """Once the headline is measured, a rank that raises (or hangs past the deadline) inside an optional leg ends the job
    through the flag file: rank 0 prints the line it holds with the leg's error, exit code 0."""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '4', '--steps', '1', '--selftest-launch'],
                       env=_env(RFN_DIST_BACKEND='gloo', RFN_BENCH_FAIL_RANK='2', RFN_BENCH_FAIL_WHERE=where, OMP_NUM_THREADS='1'),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0]['n_gpus'] == 4 and 'rank 2' in lines[0]['strong']['error']


@pytest.mark.gpu
def test_two_rank_train_step_line_on_one_gpu():
    """The real N=2 path end to end (model, GradSync buckets, fused Adam, max-over-ranks timing); the two ranks share
    cuda:0 and exchange the buckets over gloo because a one-GPU box cannot host two RCCL ranks."""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--workload', 'c2', '--batch', '4', '--steps', '1',
                        '--warmup', '1', '--no-cpu-baseline'],
                       env=_env(RFN_DIST_BACKEND='gloo', RFN_DEVICE_INDEX='0', HSA_ENABLE_IPC_MODE_LEGACY='0'),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1
    out = lines[0]
    assert out['n_gpus'] == 2 and out['rccl_ranks'] == 2 and out['dist_backend'] == 'gloo'
    assert out['value'] > 0 and out['config']['global_batch'] == 8 and 'roofline' in out


@pytest.mark.gpu
def test_single_gpu_line_has_roofline_and_cpu_baseline():
    r = subprocess.run([sys.executable, BENCH, '--workload', 'c2', '--batch', '8', '--steps', '2', '--warmup', '1',
                        '--cpu-sample', '2'], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_lines(r.stdout)[0]
    assert out['n_gpus'] == 1 and out['rccl_ranks'] == 1
    assert set(out['roofline']) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'}
    assert out['cpu_baseline']['kind'] == 'port' and out['cpu_baseline']['value'] > 0


@pytest.mark.gpu
def test_one_rank_rccl_group_with_lean_tiles_is_bit_equal_to_the_plain_step():
    """DP readiness on one GPU: the full train step of bench.py inside a 1-rank RCCL process group
    (RFN_FORCE_DIST=1: every bucket goes through an asynchronous nccl all-reduce) with the big-tile GEMMs in the lean LDS
    configuration a data-parallel run uses (RFN_GEMM_OPT_LDS_LEAN) ends in the same bits as the plain single-process
    step: same loss, same parameters after two optimizer steps (config.digest)."""
    base = [sys.executable, os.path.join(ROOT, 'bench.py'), '--batch', '32', '--steps', '2', '--warmup', '1', '--settle', '0',
            '--no-cpu-baseline', '--no-alt-line', '--digest']
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('RFN_FORCE_DIST', None)
    plain = subprocess.run(base, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert plain.returncode == 0, plain.stderr[-2000:]
    a = json.loads([l for l in plain.stdout.splitlines() if l.startswith('{"metric"')][-1])
    env_d = dict(env, RFN_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29500 + os.getpid() % 1000))
    dist = subprocess.run(base + ['--lds-lean'], env=env_d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=900)
    assert dist.returncode == 0, dist.stderr[-2000:]
    b = json.loads([l for l in dist.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert a['rccl_ranks'] == 1 and a['dist_backend'] is None and a['config']['gemm_flags'] == 0
    assert b['rccl_ranks'] == 1 and b['dist_backend'] == 'nccl' and b['config']['gemm_flags'] & 1
    assert a['config']['digest'] and a['config']['digest'] == b['config']['digest'], (a['config'], b['config'])


@pytest.mark.gpu
def test_one_rank_rccl_group_runs_the_sharded_updates_rccl_branches_bit_equal_and_without_allocating():
    """the code the first multi-GPU `--shard-optimizer` run takes -- per bucket `reduce_scatter_tensor` on a
    side stream, clamp + Adam on the shard (update_bucket_early), `all_gather_into_tensor` of the parameters, the padded flat
    buckets, `param_wait_hook` before the next forward -- executed on RCCL by a ONE-rank process group (RFN_FORCE_DIST=1;
    FusedClampAdam(shard=(0, 1)) takes the sharded path whenever a group exists).  After 1 warm-up + 2 timed steps the loss,
    every parameter and both Adam moments (config.digest) equal the plain single-process optimizer's bit for bit, and the
    timed region performed no device allocation (the reduce-scatter outputs are allocated once)."""
    # 3 warm-up steps: a gradient bucket handed to the side stream (record_stream) is recycled one step later than on one
    # stream, so the allocator reaches its steady state -- two generations of flat buffers -- after the second step
    base = [sys.executable, os.path.join(ROOT, 'bench.py'), '--batch', '32', '--steps', '2', '--warmup', '3', '--settle', '0',
            '--no-cpu-baseline', '--no-alt-line', '--digest']
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('RFN_FORCE_DIST', None)
    plain = subprocess.run(base, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert plain.returncode == 0, plain.stderr[-2000:]
    a = json.loads([l for l in plain.stdout.splitlines() if l.startswith('{"metric"')][-1])
    env_d = dict(env, RFN_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29500 + os.getpid() % 1000))
    dist = subprocess.run(base + ['--shard-optimizer'], env=env_d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=900)
    assert dist.returncode == 0, dist.stderr[-2000:]
    b = json.loads([l for l in dist.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert a['config']['shard_optimizer'] is False and a['dist_backend'] is None
    assert b['config']['shard_optimizer'] is True and b['dist_backend'] == 'nccl' and b['rccl_ranks'] == 1
    assert a['config']['digest'] and a['config']['digest'] == b['config']['digest'], (a['config'], b['config'])
    assert b['device_mallocs_frees_in_timed_region'] == 0, b['device_mallocs_frees_in_timed_region']
    assert b['strong']['device_mallocs_frees_in_timed_region'] == 0


@pytest.mark.gpu
def test_failing_rank_inside_the_train_step_exits_fast():
    """rank 1 raises inside run_train after its first step while rank 0 is inside the bucket all-reduce of
    the next one.  The parent must return non-zero within 60 s of the failure (no finally-barrier, no watchdog wait)."""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--workload', 'c2', '--batch', '4', '--steps', '2',
                        '--warmup', '1', '--settle', '0.5', '--no-cpu-baseline', '--no-alt-line'],
                       env=_env(RFN_DIST_BACKEND='gloo', RFN_DEVICE_INDEX='0', HSA_ENABLE_IPC_MODE_LEGACY='0',
                                RFN_BENCH_FAIL_RANK='1'), capture_output=True, text=True, timeout=900)
    done = time.time()
    assert r.returncode != 0 and not _json_lines(r.stdout), r.stdout
    assert 'injected failure on rank 1 in the headline leg' in r.stderr
    assert done - _failed_at(r.stderr) < 60.0


@pytest.mark.gpu
def test_failing_rank_inside_the_strong_leg_keeps_the_headline():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--workload', 'c2', '--batch', '8', '--steps', '2',
                        '--warmup', '1', '--settle', '0.5', '--no-cpu-baseline', '--no-alt-line'],
                       env=_env(RFN_DIST_BACKEND='gloo', RFN_DEVICE_INDEX='0', HSA_ENABLE_IPC_MODE_LEGACY='0',
                                RFN_BENCH_FAIL_RANK='1', RFN_BENCH_FAIL_WHERE='strong'),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_lines(r.stdout)[0]
    assert out['value'] > 0 and 'roofline' in out and 'injected failure' in out['strong']['error']


@pytest.mark.gpu
def test_two_rank_line_carries_weak_and_strong_legs_with_exposed_wait():
    """one run answers both readings of the metric -- the weak headline (B captions per rank) and the
    SAME global batch sharded (`strong`) -- each with the time its step waited for the gradient exchange."""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--workload', 'c2', '--batch', '8', '--steps', '2',
                        '--warmup', '1', '--settle', '0.5', '--no-cpu-baseline'],
                       env=_env(RFN_DIST_BACKEND='gloo', RFN_DEVICE_INDEX='0', HSA_ENABLE_IPC_MODE_LEGACY='0'),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_lines(r.stdout)[0]
    assert out['scaling'] == 'weak' and out['config']['global_batch'] == 16 and out['exposed_ms'] >= 0.0
    st = out['strong']
    assert st['scaling'] == 'strong' and st['global_batch'] == 8 and st['captions_per_gpu'] == 4
    assert st['value'] > 0 and st['exposed_ms'] >= 0.0 and set(st['exposed_ms_by_bucket_rank0']) >= {'decoder', 'core', 'enc0a', 'enc1b'}
    x3 = out['bf16x3']
    assert x3['updates'] == out['config']['updates'] and abs(x3['final_loss'] - out['config']['final_loss']) < 0.05 * abs(out['config']['final_loss'])


@pytest.mark.gpu
def test_one_rank_rccl_line_carries_strong_and_exposed_ms():
    env = _env(RFN_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29500 + os.getpid() % 1000),
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, BENCH, '--workload', 'c2', '--batch', '16', '--steps', '3', '--warmup', '1',
                        '--settle', '0.5', '--no-cpu-baseline', '--no-alt-line'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = _json_lines(r.stdout)[0]
    assert out['dist_backend'] == 'nccl' and out['exposed_measured_by'].startswith('hip events')
    assert out['exposed_ms'] >= 0.0 and out['strong']['exposed_ms'] >= 0.0 and out['strong']['captions_per_gpu'] == 16

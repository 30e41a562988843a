"""bench.py as the driver runs it: `python bench.py --gpus N ...` with no launcher around it must start the N ranks
itself (fresh child processes, decided before torch or the GPU is touched), relay ONE JSON line and propagate a
failing rank as a non-zero exit code."""
import json
import os
import subprocess
import sys

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


def test_strong_flag_and_failing_rank_propagates():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '1', '--selftest-launch', '--strong'],
                       env=_env(RFN_DIST_BACKEND='gloo'), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and _json_lines(r.stdout)[0]['scaling'] == 'strong'
    # ranks that disagree with --gpus die: the parent must not print a line and must exit non-zero
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '1', '--selftest-launch'],
                       env=_env(RFN_DIST_BACKEND='gloo', RFN_BENCH_FAIL_RANK='1'), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and not _json_lines(r.stdout)


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

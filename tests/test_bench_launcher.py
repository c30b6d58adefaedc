"""bench.py --gpus N must run N ranks (it silently ran one).  Driven here with the CPU stand-in
engine under gloo: the launcher, the rank / device checks and the max-over-ranks timing are what is tested."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')
STANDIN = os.path.join(ROOT, 'tests', 'bench_standin.py')


def run(args, env_extra=None, drop=()):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT') + tuple(drop)}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=300)


def last_json(stdout):
    lines = [l for l in stdout.strip().splitlines() if l.startswith('{')]
    assert lines, stdout
    return json.loads(lines[-1])


def test_plain_command_launches_n_ranks():
    """`python bench.py --gpus 2` (the driver's form, no torchrun around it) yields a 2-rank line."""
    r = run(['--gpus', '2', '--steps', '2', '--warmup', '1', '--frames', '4', '--standin', STANDIN])
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d['n_gpus'] == 2 and d['rccl_ranks'] == 2
    assert d['launch'].startswith('self-launched')
    assert d['config']['parallelism'] == 'dp2' and d['config']['global_frames_per_step'] == 8
    assert d['config']['all_reduce'] and d['scaling'] == 'weak'
    assert d['value'] > 0 and abs(d['value'] - 2 * 4 * 2 / (d['ms_per_step'] * 2e-3)) / d['value'] < 1e-6
    # exactly one JSON line: only rank 0 prints
    assert len([l for l in r.stdout.splitlines() if l.startswith('{"metric"')]) == 1


def test_single_rank_line():
    r = run(['--steps', '2', '--warmup', '1', '--frames', '4', '--standin', STANDIN])
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d['n_gpus'] == 1 and d['rccl_ranks'] == 1 and d['launch'] == 'single process'


def test_refuses_world_size_mismatch():
    """N ranks asked for, another number running: non-zero exit status and no JSON line."""
    r = run(['--gpus', '2', '--steps', '1', '--warmup', '0', '--frames', '4', '--standin', STANDIN],
            env_extra={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'refusing' in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


def test_refuses_without_enough_devices():
    """No GPU in this container: the real (non-stand-in) path must refuse --gpus 2 instead of printing a line."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip('two devices visible')
    r = run(['--gpus', '2', '--steps', '1', '--warmup', '0'])
    assert r.returncode != 0 and 'refusing' in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]

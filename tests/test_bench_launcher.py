"""CPU: `python bench.py --gpus N` starts its own ranks (no torchrun), and also runs as one rank under torchrun's env.

The GPU step is replaced by `--dry-run` (gloo, a sleep per step); what is exercised is exactly what the driver's multi-GPU
command depends on: the launcher, the 127.0.0.1 rendezvous, the barriers, the MAX-over-ranks timing and the single JSON
line from rank 0.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _clean_env():
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR', 'LOCAL_WORLD_SIZE'):
        env.pop(k, None)
    return env


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith('{')]


def test_self_launch_world_2():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '3', '--warmup', '0', '--dry-run'],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                               # ONE line, from rank 0
    d = lines[0]
    assert d['n_gpus'] == 2 and d['config']['world_size'] == 2 and d['steps'] == 3 and d['dry_run'] is True
    # MAX over ranks: rank 1 sleeps 20 ms per step, rank 0 only 10 ms
    assert d['ms_per_step'] >= 19.0
    assert d['scaling'] == 'weak' and d['higher_is_better'] is True


def test_runs_as_a_rank_under_torchrun_env():
    """The driver's form: torch.distributed.run sets WORLD_SIZE etc. and starts bench.py once per rank."""
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29611', BENCH, '--gpus', '2', '--steps', '2',
                        '--warmup', '0', '--dry-run'], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]['n_gpus'] == 2


def test_mismatched_world_is_an_error_not_a_hang():
    env = _clean_env()
    env.update(WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '4', '--dry-run'], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and 'does not match WORLD_SIZE' in r.stderr


def test_failing_rank_fails_the_launcher():
    """Without a GPU the real (non dry-run) ranks exit with an error; the launcher must report it, not hang."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip('GPU present')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '1', '--warmup', '0'], env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not _json_lines(r.stdout)

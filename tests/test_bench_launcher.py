"""bench.py's own multi-rank launcher on CPU (gloo): `--gpus 2` without a launcher must start two ranks, rendezvous,
give each its contiguous block of the pair list, take the MAX of the per-rank timings and report n_gpus == 2; a
`--gpus` that disagrees with WORLD_SIZE must fail instead of silently measuring one GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], env=e, capture_output=True, text=True, timeout=300)


def test_gpus_2_starts_two_ranks():
    r = _run(['--gpus', '2', '--dry-run', '--steps', '3', '--batch', '4'])
    assert r.returncode == 0, r.stderr
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['dry_run'] is True and line['steps'] == 3
    assert sorted(map(tuple, line['shard_plan'])) == [(0, 0, 12), (1, 12, 24)]       # 2 x 3 x 4 pairs, contiguous blocks
    assert line['elapsed_max_s'] >= 0.02                                             # rank 1's (longer) time won the MAX


def test_single_rank_dry_run():
    r = _run(['--dry-run', '--steps', '2', '--batch', '8'])
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and line['shard_plan'] == [[0, 0, 16]]


def test_gpus_must_match_world_size():
    r = _run(['--gpus', '4', '--dry-run'], env={'RANK': '0', 'LOCAL_RANK': '0', 'WORLD_SIZE': '2'})
    assert r.returncode != 0 and 'WORLD_SIZE=2' in (r.stderr + r.stdout)
    r = _run(['--gpus', '1', '--dry-run'], env={'RANK': '0', 'LOCAL_RANK': '0', 'WORLD_SIZE': '2'})
    assert r.returncode != 0


def test_gpus_8_starts_eight_ranks():
    """The 8-GPU node's launch shape (BASELINE configs[4]: 1024 pairs, 128 per GPU), without GPUs: eight ranks over gloo."""
    r = _run(['--gpus', '8', '--dry-run', '--steps', '16', '--batch', '8'])
    assert r.returncode == 0, r.stderr
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 8 and line['scaling'] == 'weak'
    plan = sorted(map(tuple, line['shard_plan']))
    assert plan == [(k, 128 * k, 128 * (k + 1)) for k in range(8)]                  # 1024 pairs, contiguous blocks of 128
    assert line['elapsed_max_s'] >= 0.08                                             # rank 7's time won the MAX


def test_a_dead_rank_ends_the_launch_quickly():
    """one rank dying before the rendezvous must not leave its siblings waiting for the collective timeout -
    the launcher polls every child, terminates the rest and reports which rank failed."""
    import time
    t = time.time()
    r = _run(['--gpus', '4', '--dry-run', '--steps', '2'], env={'GEOFORMER_BENCH_FAIL_RANK': '2'})
    assert r.returncode != 0 and 'first failure: rank 2' in r.stderr, r.stderr
    assert time.time() - t < 60

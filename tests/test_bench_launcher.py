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


def test_ranks_are_pinned_to_disjoint_core_sets():
    """every rank pins itself (os.sched_setaffinity inside the rank's process, before any GPU call) to its own
    block of host cores; the dry run reports the sets."""
    r = _run(['--gpus', '2', '--dry-run', '--steps', '2'])
    assert r.returncode == 0, r.stderr
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    a, b = line['rank_cpu_sets']
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 2:
        assert a and b and not set(a) & set(b) and len(a) == len(b) == ncpu // 2
    r = _run(['--gpus', '2', '--dry-run', '--steps', '2'], env={'GEOFORMER_BENCH_NO_PIN': '1'})
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['rank_cpu_sets'] == [None, None]


def test_rank_cpu_set_follows_the_gpus_numa_nodes(tmp_path):
    """The cores of the GPU's NUMA node (sysfs: amdgpu PCI functions in bus order), shared evenly by the ranks on that node; an even
    split of the allowed cores where sysfs says nothing."""
    sys.path.insert(0, ROOT)
    import bench
    root = tmp_path / 'sys'
    for k, node in enumerate([0, 0, 1, 1]):
        d = root / 'bus/pci/drivers/amdgpu' / f'0000:{10 * (k + 1):02x}:00.0'
        d.mkdir(parents=True)
        (d / 'numa_node').write_text(f'{node}\n')
    for node, cl in ((0, '0-7,16-23'), (1, '8-15,24-31')):
        d = root / f'devices/system/node/node{node}'
        d.mkdir(parents=True)
        (d / 'cpulist').write_text(cl + '\n')
    allowed = range(32)
    sets = [bench.rank_cpu_set(r, 4, allowed, str(root)) for r in range(4)]
    assert sets[0] == [0, 1, 2, 3, 4, 5, 6, 7] and sets[1] == [16, 17, 18, 19, 20, 21, 22, 23]
    assert sets[2] == [8, 9, 10, 11, 12, 13, 14, 15] and sets[3] == [24, 25, 26, 27, 28, 29, 30, 31]
    # no sysfs information: contiguous even split
    assert bench.rank_cpu_set(1, 4, allowed, str(tmp_path / 'none')) == list(range(8, 16))
    assert bench.rank_cpu_set(0, 1, [3, 4], str(tmp_path / 'none')) == [3, 4]
    assert bench.PIN_RULE.startswith('even split')
    # a partial lease: the local rank goes through HIP_VISIBLE_DEVICES - ranks 0, 1 on GPUs 2, 3 = NUMA node 1
    env = {'HIP_VISIBLE_DEVICES': '2,3'}
    assert bench.rank_cpu_set(0, 2, allowed, str(root), env) == [8, 9, 10, 11, 12, 13, 14, 15]
    assert bench.rank_cpu_set(1, 2, allowed, str(root), env) == [24, 25, 26, 27, 28, 29, 30, 31]
    assert bench.PIN_RULE == 'numa via HIP_VISIBLE_DEVICES'
    # HIP's list indexes into what ROCr leaves visible
    env = {'ROCR_VISIBLE_DEVICES': '1,2,3', 'HIP_VISIBLE_DEVICES': '0,2'}
    assert bench.rank_cpu_set(0, 2, allowed, str(root), env) == list(range(0, 8)) + list(range(16, 24))     # GPU 1: alone on node 0
    assert bench.rank_cpu_set(1, 2, allowed, str(root), env) == list(range(8, 16)) + list(range(24, 32))    # GPU 3: alone on node 1
    # a list that cannot be read as indices (UUIDs), or one shorter than the ranks: even split, and the rule says why
    assert bench.rank_cpu_set(1, 2, allowed, str(root), {'HIP_VISIBLE_DEVICES': 'GPU-abc,GPU-def'}) == list(range(16, 32))
    assert bench.PIN_RULE.startswith('even split: HIP_VISIBLE_DEVICES')
    assert bench.rank_cpu_set(1, 2, allowed, str(root), {'HIP_VISIBLE_DEVICES': '3'}) == list(range(16, 32))
    # functions bound to amdgpu that are not devices (class != display / processing accelerator) do not shift the order
    d = root / 'bus/pci/drivers/amdgpu' / '0000:05:00.1'
    d.mkdir(parents=True); (d / 'numa_node').write_text('1\n'); (d / 'class').write_text('0x040300\n')
    assert bench.rank_cpu_set(0, 4, allowed, str(root), {}) == [0, 1, 2, 3, 4, 5, 6, 7]

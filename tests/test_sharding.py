"""Multi-GPU layout on CPU: world_size-2 gloo run of the pair sharder (no data-path collective) with the
oracle as the matcher; the sharded result must equal the single-process result pair for pair."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_and_balance():
    from geoformer_amd.shard import shard_bounds
    for n in (0, 1, 7, 8, 1024, 1031):
        for w in (1, 2, 3, 8):
            blocks = [shard_bounds(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(1024, 8, 3) == (384, 512)            # BASELINE configs[4]: 128 pairs per GPU
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _match_fn():
    for p in (ROOT, os.path.join(ROOT, 'oracle')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import geoformer_oracle as O
    import golden_inputs as GI
    W = O.make_weights()
    cfg = O.default_geo_config(); cfg.update(coarse_thr=0.2, fine_thr=0.1)

    def fn(seeds):
        out = []
        for s in seeds:
            feats = GI.planted_features(1, 6, 8, 6, 8, 500 + s)
            d = O.geoformer_forward(W, {'image0': torch.zeros(1, 1, 48, 64), 'image1': torch.zeros(1, 1, 48, 64)}, None, cfg,
                                    lambda a, b: (None, None), None, feats)
            out.append((s, len(d['b_ids']), float(d['mkpts0_f'].sum()), float(d['mconf'].sum())))
        return out
    return fn


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from geoformer_amd.shard import run_sharded
    res = run_sharded(list(range(5)), _match_fn(), batch=2)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo_matches_single_process():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from geoformer_amd.shard import run_sharded
    torch.set_num_threads(1)
    single = run_sharded(list(range(5)), _match_fn(), batch=2)
    assert [r[0] for r in single] == [0, 1, 2, 3, 4] and all(r[1] > 0 for r in single)
    assert got[0] == single and got[1] == single

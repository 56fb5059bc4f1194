#!/usr/bin/env python3
"""Throughput bench of the GeoFormer matching path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--precision fp16|bf16|fp32]

Workload (BASELINE.json configs[4], "batched inference, 1024 synthetic 640x640 pairs, fp16 features, 8 GPUs
embarrassingly-parallel shard"): the pair list (seeds 0 .. N*K*B-1) is cut into contiguous per-rank blocks
(`geoformer_amd.shard.shard_bounds`, the reference's own habit: homodataset/HomoDataset.py:40-45) and every rank
runs `steps` batches of `batch` pairs of its block through the FULL forward (ResNet-FPN backbone on PyTorch-ROCm +
the HIP matching path).  There is no data-path collective (pairs are independent), so scaling is weak: per-GPU
work is fixed as N grows.  Inputs are resident in HBM before the timed region.  Weights: deterministic
closed-form fill of the reference architecture (no checkpoint offline).  Random-init weights cannot produce the load
SURVEY section 8 sizes the path for from images (M ~ 2000 coarse matches, K ~ 1000 inlier cells per pair at the
reference's thresholds 0.2 / 0.1), so the default workload (`--pairs planted`) runs the backbone on the images and feeds
the matching path planted-correspondence feature maps: `value` is the throughput at that NOMINAL load (M ~ 2300,
K ~ 1200, reported in `config` because cost scales with them).  `--pairs homography` is the light-load run of rounds 1-2
(thresholds 0, M ~ 360, K ~ 60), kept as a side measurement.

Launching: `--gpus N` (N > 1) without a launcher starts the N ranks ITSELF (fresh child processes, before anything
in the parent touches the GPU), waits for them and relays rank 0's line; under `python -m torch.distributed.run`
(RANK / LOCAL_RANK / WORLD_SIZE in the environment) it is one of the ranks.  `--gpus` must equal the world size.

One JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import ctypes
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
MFMA_PEAK_TFLOPS = {'fp16': 2500.0, 'bf16': 2500.0, 'fp32': 157.3}     # dense, same table

# kernel families that carry HIP events in libgeoformer_hip.so (gf_prof_begin tags): what the declared work unit
# is, which roofline bounds the family, and whether it belongs to the matching path (the north_star's hot path)
ROOFLINE_TAGS = [
    # tag,                   bound,  hot path, description
    ('enc_layer',            'mfma', True,  'fused encoder layer (projections + attention apply + merge/LN + MLP/LN + residual; since round 4 most launches also '
                                            "leave the linear-attention state of their output rows for the next layer call: the state tail's flops are declared with them)"),
    ('enc_kv_state',         'mfma', True,  'fused k/v projection + linear-attention state as a pass of its own (the sources no earlier layer call of the transformer produced)'),
    ('k3_linear',            'hbm',  True,  'K3 linear_kernel family (Geo-layer projections, FinePreprocess and fine-level GEMMs with fused epilogues; K <= 512: below the machine balance, HBM is the roof)'),
    ('k1_stats',             'mfma', True,  'K1 pass A (similarity tile statistics)'),
    ('k1_conf',              'hbm',  True,  'K1 pass B (dual-softmax correlation sweep, conf_matrix write)'),
    ('k1_unit',              'hbm',  True,  'K1 as a unit: one CoarseMatching call (pass A + reduction + pass B + selection + compaction) against its algorithmic bytes'),
    ('fine_layer',           'mfma', True,  'fused fine-level encoder layer (25-token windows: projections + window attention + merge/LN + MLP/LN + residual)'),
    ('k2_linear_attention',  'hbm',  True,  'K2 linear attention (state + apply)'),
    ('k5_window_attention',  'hbm',  True,  'K5 windowed cross attention (L2 gather)'),
    ('k4_self_attention',    'mfma', True,  'K4 inlier-key self attention (flash form)'),
    ('bias_act',             'hbm',  False, 'backbone glue: shift + shortcut + activation stream'),
    ('k3_upadd',             'hbm',  False, 'backbone: 1x1 lateral convolution with the FPN upsample + add as its epilogue (K3 tile engine; 67 flop/byte: HBM is the roof)'),
    ('conv1x1',              'hbm',  False, 'backbone: 1x1 convolutions on the K3 tile engine (layer3_outconv, the stride-2 downsample shortcuts)'),
    ('conv3x3',              'mfma', False, 'K10 3x3 convolution of the backbone (BN shift + shortcut + activation in the epilogue; SURVEY 8f rank 4)'),
]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8, help='pairs per GPU per step')
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--precision', default='fp16', choices=['fp16', 'bf16', 'fp32'])
    ap.add_argument('--pairs', default='planted', choices=['planted', 'homography', 'shift'],
                    help="'planted' (default): the nominal-load workload - backbone on the images, matching path on "
                         "planted-correspondence feature maps, the reference's thresholds 0.2 / 0.1; 'homography': image1 = image0 "
                         "under a random homography, thresholds 0 (light load: random-init weights give few matches); 'shift': "
                         "shifted by one coarse cell (the pair of the reference's CPU measurement, BASELINE.md section 2)")
    ap.add_argument('--coarse-thr', type=float, default=None, help='default: 0.2 with --pairs planted (geo_config.py:13), else 0')
    ap.add_argument('--fine-thr', type=float, default=None, help='default: 0.1 with --pairs planted (geo_config.py:15), else 0')
    ap.add_argument('--pmc-file', default=None,
                    help='per-tag PMC summary (tools/pmc_roofline.py --tags) whose HBM traffic / MFMA utilisation are attached to the '
                         'roofline entries; default: the newest profiles/r*_pmc_per_tag.json recorded for this configuration')
    ap.add_argument('--streams', type=int, default=2, help='concurrent forward pipelines (host threads, one HIP stream each)')
    ap.add_argument('--tune', action='store_true', help='let MIOpen search its convolution algorithms (minutes)')
    ap.add_argument('--save-db', action='store_true', help='with --tune: copy the searched find-db over geoformer_amd/miopen_db')
    ap.add_argument('--graphs', action='store_true', help='replay the static part of the forward from a captured hipGraph (GeoFormer.enable_graphs)')
    ap.add_argument('--repeats', type=int, default=3,
                    help='timed regions of K steps each: the FIRST is `value` (W warm-up steps, then exactly K timed steps); the others '
                         'follow it back to back and only feed the min / median / max of `repeats` in the line')
    ap.add_argument('--independent', action='store_true',
                    help="with --pairs planted: feed the matching path the planted maps WITHOUT the data dependency on that step's backbone "
                         'output (rounds 3\'s headline; now the side measurement nominal_independent)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the nominal-load and fp32 parity-mode side measurements')
    ap.add_argument('--no-train', action='store_true', help='skip the training-step side measurement (BASELINE configs[2] / [3])')
    ap.add_argument('--dry-run', action='store_true',
                    help='launcher / rendezvous / shard plan only, on CPU over gloo (what tests/test_bench_launcher.py runs)')
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# launcher: --gpus N without RANK/WORLD_SIZE in the environment
# ---------------------------------------------------------------------------------------------------------------
def spawn_ranks(args, argv):
    """Starts N fresh child processes of this script (one per GPU) and relays rank 0's JSON line.  Nothing in this
    parent initialises the GPU.  All children are polled: the first non-zero exit terminates the others (a dead rank
    would otherwise leave its siblings in the rendezvous or a barrier until the collective timeout).  The rendezvous
    port is probed free right before the ranks start and the launch is retried once on another port if rank 0 reports
    it taken.  Returns the exit code: 0 only if every rank exited 0 and the line says n_gpus == N."""
    import tempfile
    for attempt in range(2):
        with socket.socket() as s:
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        procs, errs = [], []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
                       MASTER_PORT=str(port), GEOFORMER_BENCH_CHILD='1')
            env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            err = tempfile.TemporaryFile() if r else None                # ranks > 0: stdout dropped, stderr kept for the report
            errs.append(err)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=err))
        import threading
        out0 = []
        reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        failed = None
        while any(p.poll() is None for p in procs):
            for r, p in enumerate(procs):
                if p.poll() not in (None, 0) and failed is None:
                    failed = r
                    for q in procs:
                        if q.poll() is None:
                            q.terminate()
            time.sleep(0.05)
        reader.join(timeout=10)
        text = (out0[0] if out0 else b'').decode()
        codes = [p.returncode for p in procs]
        tails = ''
        for r, err in enumerate(errs):
            if err is not None:
                err.seek(0)
                t = err.read().decode(errors='replace')
                if codes[r]:
                    tails += f'--- rank {r} stderr tail ---\n' + t[-1500:] + '\n'
                err.close()
        if failed is not None and attempt == 0 and ('EADDRINUSE' in tails or 'address already in use' in tails.lower()):
            continue                                                      # someone took the port between probe and bind
        break
    sys.stdout.write(text)
    sys.stdout.flush()
    if any(codes):
        print(f'bench.py: ranks exited with {codes} (first failure: rank {failed})\n{tails}', file=sys.stderr)
        return 1
    try:
        line = json.loads([ln for ln in text.splitlines() if ln.startswith('{')][-1])
    except (IndexError, ValueError):
        print('bench.py: rank 0 printed no JSON line', file=sys.stderr)
        return 1
    if line.get('n_gpus') != args.gpus:
        print(f"bench.py: {line.get('n_gpus')} ranks joined, {args.gpus} requested", file=sys.stderr)
        return 1
    return 0


def _cpulist(text):
    out = []
    for part in text.strip().split(','):
        if part:
            a, _, b = part.partition('-')
            out += list(range(int(a), int(b or a) + 1))
    return out


PIN_RULE = None          # which rule rank_cpu_set used in this process ('numa', 'numa via HIP_VISIBLE_DEVICES', 'even split: <why>'): in the JSON line


def _visible_device_index(local, env=None):
    """The physical device behind local rank `local`: HIP_VISIBLE_DEVICES (applied to what ROCR_VISIBLE_DEVICES leaves visible) as lists of
    indices.  (index, variable name) - (local, None) when neither is set; (None, name) when a list cannot be read as indices (UUID form) or
    is too short: the caller then falls back to the even split instead of guessing a NUMA node."""
    env = os.environ if env is None else env
    idx, used = local, None
    for name in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES'):           # HIP's list indexes into ROCr's
        v = env.get(name)
        if v is None or v.strip() == '':
            continue
        try:
            ids = [int(t) for t in v.split(',')]
        except ValueError:
            return None, name
        if idx >= len(ids) or ids[idx] < 0:
            return None, name
        idx, used = ids[idx], (name if used is None else used + '+' + name)
    return idx, used


def rank_cpu_set(local, nlocal, allowed=None, sysfs='/sys', env=None):
    """The host cores rank `local` of `nlocal` on this node should run on: the cores of its GPU's NUMA node, shared evenly by the ranks on
    that node.  The GPU of a rank = the amdgpu PCI functions of DISPLAY / PROCESSING-ACCELERATOR class in bus order (= HIP device order
    without remapping), indexed by the local rank translated through HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when those are set (a
    partial lease such as GPUs 4-7).  Whenever that chain has a gap (no sysfs, an unreadable device list, a node without cores) the rule is an
    even contiguous split of the allowed cores; PIN_RULE records which rule was used.  Reads sysfs only - nothing here touches the GPU,
    so it can run before the first HIP call of the process."""
    global PIN_RULE
    allowed = sorted(os.sched_getaffinity(0)) if allowed is None else sorted(allowed)
    nodes, why = [], 'no amdgpu devices in sysfs'
    try:
        drv = os.path.join(sysfs, 'bus/pci/drivers/amdgpu')
        for d in sorted(x for x in os.listdir(drv) if ':' in x):
            try:                                                   # 0x03xxxx display, 0x12xxxx processing accelerator; other functions
                cls = int(open(os.path.join(drv, d, 'class')).read(), 16) >> 16    # (audio, USB ...) bound to amdgpu are not devices
            except (OSError, ValueError):
                cls = 0x03                                         # no class file (the unit test's tree): count it
            if cls in (0x03, 0x12):
                nodes.append(int(open(os.path.join(drv, d, 'numa_node')).read()))
    except (OSError, ValueError):
        nodes = []
    dev = [_visible_device_index(r, env) for r in range(nlocal)]
    via = dev[local][1]
    if any(i is None for i, _ in dev):
        why = f'{via} is not a list of indices covering {nlocal} ranks'
    elif nodes and all(i < len(nodes) and nodes[i] >= 0 for i, _ in dev):
        mine = [nodes[i] for i, _ in dev]
        node = mine[local]
        try:
            cores = [c for c in _cpulist(open(os.path.join(sysfs, f'devices/system/node/node{node}/cpulist')).read()) if c in set(allowed)]
        except OSError:
            cores = []
        sharers = [r for r in range(nlocal) if mine[r] == node]
        if len(cores) >= len(sharers):
            k, per = sharers.index(local), len(cores) // len(sharers)
            PIN_RULE = 'numa' + (f' via {via}' if via else '')
            return cores[k * per:(k + 1) * per]
        why = f'NUMA node {node} has fewer allowed cores than ranks'
    elif nodes:
        why = 'a device index beyond the amdgpu list, or a device without a NUMA node'
    PIN_RULE = 'even split: ' + why
    per = max(1, len(allowed) // max(nlocal, 1))
    return allowed[local * per:(local + 1) * per] or allowed


def pin_rank(local, nlocal):
    """Per-rank CPU affinity: set inside the rank's own process BEFORE anything initialises the GPU (never through
    taskset / numactl in front of a profiled program: that is an exec after the profiler's preload has touched the GPU).  The two
    host pipelines, the caching allocator's and RCCL's helper threads inherit it."""
    if nlocal <= 1 or os.environ.get('GEOFORMER_BENCH_NO_PIN') == '1':
        return None
    cores = rank_cpu_set(local, nlocal)
    try:
        os.sched_setaffinity(0, cores)
    except OSError:
        return None
    return cores


def dist_env(args):
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (('RANK', 0), ('LOCAL_RANK', 0), ('WORLD_SIZE', 1)))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: pass --gpus equal to the number of ranks '
                         f'(or drop the launcher and let bench.py start them)')
    return rank, local, world


def dry_run(args):
    """The distributed skeleton without a GPU: rendezvous over gloo, the shard plan, the barrier-bracketed timed
    region and the MAX all-reduce of its duration - everything bench.py does around the forward passes."""
    import torch
    import torch.distributed as dist
    from geoformer_amd.shard import shard_bounds
    rank, local, world = dist_env(args)
    pinned = pin_rank(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    if os.environ.get('GEOFORMER_BENCH_FAIL_RANK') == str(rank):          # launcher test: a rank that dies before the rendezvous
        raise SystemExit(3)
    if world > 1:
        import datetime
        dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    total = args.steps * args.batch * world
    lo, hi = shard_bounds(total, world, rank)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (1 + rank))                         # rank-dependent 'work': the MAX must come from the last rank
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    plan = [None] * world
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_gather_object(plan, (rank, lo, hi))
        pins = [None] * world
        dist.all_gather_object(pins, sorted(pinned) if pinned else None)
    else:
        plan = [(rank, lo, hi)]
        pins = [None]
    if rank == 0:
        print(json.dumps({'metric': 'image-pairs/sec (640x640)', 'value': 0.0, 'unit': 'image-pairs/s', 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'dry_run': True, 'elapsed_max_s': float(t[0]),
                          'shard_plan': plan, 'scaling': 'weak', 'rank_cpu_sets': pins, 'rank_pin_rule': PIN_RULE}), flush=True)
    if world > 1:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------
# workload
# ---------------------------------------------------------------------------------------------------------------
def synth_pairs(batch, seed, size=640, device='cpu', kind='homography'):
    """bench-pair(seed): low-frequency texture + pixel noise; image1 = image0 under a random homography (corner
    perturbation <= 32 px) or, kind='shift', shifted by one coarse cell (8 px) in x and y (the
    pair of the reference's own CPU timing)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(batch, 1, size // 8 + 2, size // 8 + 2, generator=g)
    img = torch.nn.functional.interpolate(base, size=(size + 16, size + 16), mode='bicubic', align_corners=True)
    img = (img + 0.15 * torch.rand(batch, 1, size + 16, size + 16, generator=g)).clamp(0, 1)
    image0 = img[:, :, 8:8 + size, 8:8 + size].contiguous()
    if kind == 'shift':
        return image0.to(device), img[:, :, 16:16 + size, 16:16 + size].contiguous().to(device)
    # projective warp through a sampling grid: corners move by up to +-32 px
    src = torch.tensor([[-1., -1.], [1., -1.], [-1., 1.], [1., 1.]])
    dst = src[None] + (torch.rand(batch, 4, 2, generator=g) - 0.5) * (2 * 64.0 / size)
    A = torch.zeros(batch, 8, 8)
    for k in range(4):
        x, y = src[k]
        u, v = dst[:, k, 0], dst[:, k, 1]
        A[:, 2 * k, 0], A[:, 2 * k, 1], A[:, 2 * k, 2] = x, y, 1
        A[:, 2 * k, 6], A[:, 2 * k, 7] = -u * x, -u * y
        A[:, 2 * k + 1, 3], A[:, 2 * k + 1, 4], A[:, 2 * k + 1, 5] = x, y, 1
        A[:, 2 * k + 1, 6], A[:, 2 * k + 1, 7] = -v * x, -v * y
    h = torch.linalg.solve(A, dst.reshape(batch, 8))
    Hm = torch.cat([h, torch.ones(batch, 1)], 1).view(batch, 3, 3)
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, size), torch.linspace(-1, 1, size), indexing='ij')
    pts = torch.stack([xs, ys, torch.ones_like(xs)], -1).view(1, -1, 3) @ Hm.transpose(1, 2)
    grid = (pts[..., :2] / pts[..., 2:]).view(batch, size, size, 2) * (size / (size + 16.0))
    image1 = torch.nn.functional.grid_sample(img, grid, mode='bilinear', padding_mode='border', align_corners=True)
    return image0.to(device), image1.contiguous().to(device)


def build_model(precision, coarse_thr, fine_thr, device):
    from geoformer_amd.model.full_model import GeoFormer
    from geoformer_amd.model.cvpr_ds_config import get_default_cfg
    from geoformer_amd.model.geo_config import get_cfg_model
    from geoformer_amd.weights import deterministic_init_
    gc = get_cfg_model()
    gc.update(coarse_thr=coarse_thr, fine_thr=fine_thr, precision='fp32')
    m = deterministic_init_(GeoFormer(get_default_cfg(), gc).eval())
    W = {k: v.detach().clone() for k, v in m.state_dict().items()}      # fp32 copy for the CPU baseline leg
    m.set_precision(precision)
    return m.to(device), W


def cpu_baseline(W, coarse_thr, fine_thr, size, kind, seconds_budget=25.0):
    """The oracle (a PyTorch-CPU port of the reference's forward) on the host cores, same workload,
    a bounded sample of pairs.  torch's intra-op pool is capped at 32 threads: on the 256-thread hosts
    of the GPU boxes more threads make these medium-sized ops slower, not faster (measured 164 s/pair
    with all 256)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import geoformer_oracle as O
    import ransac_oracle as RO
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    cfg = O.default_geo_config()
    cfg.update(coarse_thr=coarse_thr, fine_thr=fine_thr)

    def one(seed, sz):
        planted = kind == 'planted'
        i0, i1 = synth_pairs(1, seed, sz, kind='shift' if planted else kind)
        feats = None
        if planted:          # the same workload as the GPU step: backbone on the images, matching path on planted maps
            c0, f0, c1, f1 = (t.float().contiguous() for t in planted_features(1, 60000 + seed, sz // 8))
            feats = ((c0, f0), (c1, f1))
        t = time.perf_counter()
        with torch.no_grad():
            if planted:
                O.backbone(W, torch.cat([i0, i1], 0))
            out = O.geoformer_forward(W, {'image0': i0, 'image1': i1}, None, cfg, RO.make_homography_fn(), None, feats)
        return time.perf_counter() - t, len(out['b_ids'])
    one(999, 160)                                    # warm the thread pool / allocator on a small pair
    t0, ts, M = time.perf_counter(), [], 0
    while len(ts) < 4 and (not ts or time.perf_counter() - t0 + ts[-1] < seconds_budget):
        dt, M = one(1000 + len(ts), size)
        ts.append(dt)
    per_pair = sum(ts) / len(ts)
    return {'value': 1.0 / per_pair, 'unit': 'image-pairs/s', 'cores': cores, 'kind': 'port',
            'sample': f'{len(ts)} synthetic {size}x{size} pairs of the same workload ({kind}), batch 1, fp32, oracle/geoformer_oracle.py '
                      f'(PyTorch-CPU port of the reference forward incl. backbone and RANSAC), {per_pair:.2f} s/pair, M={M} on the last pair'}


class Pipelines:
    """`nstreams` persistent host threads, each with its own HIP stream, taking the steps round-robin: while one
    forward waits on its two host syncs (match counts) or runs small latency-bound kernels (RANSAC, compaction) the
    other keeps the GPU fed.  All K timed steps are executed inside the timed region."""

    def __init__(self, step_fn, nstreams, dev):
        import queue
        import threading
        import torch
        self.torch, self.step_fn, self.n, self.dev = torch, step_fn, nstreams, dev
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
        self.jobs = [queue.Queue() for _ in range(nstreams)]
        self.done = queue.Queue()
        self.results = {}
        self.serial = False
        self.pool = [threading.Thread(target=self._worker, args=(w,), daemon=True) for w in range(nstreams)]
        for t in self.pool:
            t.start()

    def _record(self, i, out):
        # keep the counts and the small geometry summary only: holding every step's output (two 1.3 GB confidence
        # matrices each) made the timed steps hipMalloc fresh memory - up to 3.5x the step time on a fresh box
        self.results[i] = (len(out['b_ids']), len(out['mkpts0_f']), out.get('_geo_dev', {}).get('nidx'))

    def _worker(self, w):
        torch = self.torch
        torch.cuda.set_device(self.dev)
        with torch.cuda.stream(self.streams[w]):
            while True:
                job = self.jobs[w].get()
                if job is None:
                    return
                first, count = job[:2]
                try:
                    if len(job) > 2 and job[2] > 0 and w > 0:
                        time.sleep(w * job[2])      # staggered start (see run)
                    for i in (range(first, first + 1) if count == 0 else range(first + w, first + count, self.n)):
                        out = self.step_fn(i)
                        self._record(i, out)
                        del out
                    self.streams[w].synchronize()
                    self.done.put(w)
                except BaseException as e:          # surface a failed step instead of hanging the barrier
                    self.done.put(e)

    def run_single(self, w, i):
        self.jobs[w].put((i, 0))
        r = self.done.get()
        if isinstance(r, BaseException):
            raise r

    def run(self, first, count, stagger=0.0):
        """`stagger` (seconds): pipeline w starts w * stagger late.  Two pipelines that start a region together at batch 1 often stay IN PHASE for the
        whole region - both enqueue at the same time (sharing the interpreter), then both wait for the device: ~230 pairs/s where the out-of-phase
        mode (one enqueues while the other's kernels run) gives ~530 and one stream alone 318 (tools/b1_pipeline_probe.py, round 6); half a step of
        stagger starts them out of phase.  The headline region does not use it (8-pair steps are device-bound in either phase)."""
        if self.serial:
            for i in range(first, first + count):
                out = self.step_fn(i)
                self._record(i, out)
                del out
            self.torch.cuda.synchronize()
            return
        for w in range(self.n):
            self.jobs[w].put((first, count, stagger))
        for _ in range(self.n):
            r = self.done.get()
            if isinstance(r, BaseException):
                raise r

    def close(self):
        for q in self.jobs:
            q.put(None)


def planted_features(batch, seed, grid=80, noise=0.35, device='cpu', dtype=None):
    """Stand-in backbone outputs with planted correspondences (the construction of the parity fixtures, restated): the
    coarse and fine maps of image 1 are those of image 0 shifted by one coarse cell plus noise, so that the matching
    path runs at its nominal load (thousands of coarse matches at the reference's coarse_thr = 0.2,
    as many inlier cells) whatever the weights are."""
    import torch
    g = torch.Generator().manual_seed(seed)
    h = grid + 1
    big = torch.randn(batch, 256, h, h, generator=g) * 0.5
    bigf = torch.randn(batch, 128, 4 * h, 4 * h, generator=g)
    c0, f0 = big[:, :, :grid, :grid], bigf[:, :, :4 * grid, :4 * grid]
    c1 = big[:, :, 1:, 1:] + noise * torch.randn(batch, 256, grid, grid, generator=g)
    f1 = bigf[:, :, 4:, 4:] + noise * torch.randn(batch, 128, 4 * grid, 4 * grid, generator=g)
    # channels_last, like the maps the backbone emits (the fine-window gather reads 256-byte channel rows)
    return tuple(t.to(device=device, dtype=dtype).contiguous(memory_format=torch.channels_last) for t in (c0, f0, c1, f1))


def planted_maps(batch, seed, grid, device, dtype):
    """planted_features as the backbone hands its output over: ONE coarse and ONE fine tensor holding the image-0 maps, then the
    image-1 maps ([2N, 256, h, w], [2N, 128, 4h, 4w], channels_last) - what `planted_step` adds the backbone's output to."""
    import torch
    c0, f0, c1, f1 = planted_features(batch, seed, grid)
    cl = torch.channels_last
    return {'c': torch.cat([c0, c1], 0).to(device=device, dtype=dtype).contiguous(memory_format=cl),
            'f': torch.cat([f0, f1], 0).to(device=device, dtype=dtype).contiguous(memory_format=cl)}


def planted_step(model, i0, i1, pl, dependent):
    """One nominal-load step.  dependent (the default workload): the matching path consumes THAT STEP's backbone output - every
    feature map it reads is `planted + 0 * backbone_output`, one fused elementwise pass per map on the device (the values are the
    planted ones bit for bit, so the matches are those of the planted run, but no kernel of the matching path can start before
    the backbone's last kernel has written its maps: the reference's data dependency, full_model.py:55-61,83-101).  Not dependent:
    the backbone output is discarded and the resident planted maps are read directly (round 3's headline)."""
    import torch
    n = i0.shape[0]
    feats_c, feats_f = model._backbone(torch.cat([i0, i1], dim=0))
    if not dependent:
        return model.forward_features({'image0': i0, 'image1': i1}, pl['c'][:n], pl['f'][:n], pl['c'][n:], pl['f'][n:])
    c = torch.add(pl['c'], feats_c, alpha=0.0)
    f = torch.add(pl['f'], feats_f, alpha=0.0)
    return model.forward_features({'image0': i0, 'image1': i1}, c[:n], f[:n], c[n:], f[n:])


def measure(model, batches, steps, warmup, nstreams, dev, dist, log, profile_tag=None, L=None, planted=None, dependent=True,
            repeats=1, pipes=None):
    """W untimed + K timed steps of `model` over the resident `batches`; returns (elapsed_s, pipelines, step_fn, repeat_times).
    planted = per-batch {'c': [2N,256,h,w], 'f': [2N,128,4h,4w]} (image-0 maps, then image-1 maps): the backbone runs on the
    images and the matching path is fed the planted feature maps, through the backbone's output when `dependent`.
    repeats > 1: further timed regions of K steps each behind the first (which alone is `elapsed_s`).
    pipes: host pipelines of an earlier measurement to reuse (same threads, streams, per-stream workspaces, MIOpen handles and
    allocator pools: a SECOND set of pipelines measured ~10 % slower than the first in the same process, whatever it ran)."""
    import torch
    nres = len(batches)

    def step(i):
        i0, i1 = batches[i % nres]
        with torch.no_grad():
            if planted is None:
                return model({'image0': i0, 'image1': i1})
            return planted_step(model, i0, i1, planted[i % nres], dependent)
    if pipes is None:
        pipes = Pipelines(step, max(1, min(nstreams, steps)), dev)
    else:
        pipes.step_fn, pipes.results = step, {}
    step(0)                          # single-threaded first pass: fills the weight / table caches
    torch.cuda.synchronize()
    # each pipeline thread owns its MIOpen handle: let every one of them look its convolution algorithms up ALONE
    # first (concurrent first lookups of the user find-db were seen to leave a handle on slow fallback picks for
    # the whole process: 3.7x slower steps in ~1 of 10 fresh-box runs)
    for w in range(pipes.n):
        pipes.run_single(w, w)
    t = time.perf_counter()
    step(0)
    torch.cuda.synchronize()
    t_serial = time.perf_counter() - t
    t = time.perf_counter()
    pipes.run(0, warmup)
    torch.cuda.synchronize()
    t_threads = (time.perf_counter() - t) / max(warmup, 1)
    # the first pipelined steps of a process are sometimes slow by themselves (the caching allocator fills the second stream's
    # pool: seen as 84 ms against 15 ms once in ~10 fresh-box runs): give the pipelines two more (untimed) rounds before the verdict
    for _ in range(2):
        if not (pipes.n > 1 and warmup >= 2 and t_threads > 1.5 * t_serial):
            break
        t = time.perf_counter()
        pipes.run(0, warmup)
        torch.cuda.synchronize()
        t_threads = (time.perf_counter() - t) / max(warmup, 1)
    if pipes.n > 1 and warmup >= 2 and t_threads > 1.5 * t_serial:
        log(f'pipelined steps run at {t_threads * 1e3:.1f} ms against {t_serial * 1e3:.1f} ms single-threaded: '
            f'falling back to one host pipeline')
        pipes.serial = True
    if dist is not None:
        dist.barrier()
    if profile_tag is not None:
        # inside the timed region only ONE kernel family carries HIP events (two launches per step); event pairs
        # around all ~110 profiled launches per step were seen to slow a whole run 3.5x on some boxes
        L.gf_profile_filter(profile_tag)
        L.gf_profile_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipes.run(warmup, steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if profile_tag is not None:
        L.gf_profile_enable(0)                       # the in-region k1_conf figure belongs to the headline region only
    times = [elapsed]
    for r in range(1, repeats):                      # the same K steps again, bracketed the same way; not part of `value`
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        pipes.run(warmup + r * steps, steps)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t)
    if profile_tag is not None:
        L.gf_profile_enable(1)
    return elapsed, pipes, step, times


def load_pmc(path, run_cfg):
    """Per-tag PMC figures (HBM bytes per launch, MFMA utilisation) from a COMMITTED rocprofv3 --pmc run
    (tools/gpu_profile.sh + tools/pmc_roofline.py): they are not measured by the run that prints the line, so every entry
    that carries them names the file in `traffic_source`, and a file recorded for another configuration (precision,
    batch, size, pairs, thresholds: its `_config` block) is not attached at all."""
    import glob
    import re
    cands = [path] if path else sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_per_tag.json')),
                                       key=lambda f: [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', os.path.basename(f))],
                                       reverse=True)
    for f in cands:
        try:
            d = json.load(open(f))
        except Exception:
            continue
        cfg = d.get('_config')
        if cfg is None or any(cfg.get(k) != v for k, v in run_cfg.items()):
            continue
        src = f'{os.path.relpath(f, ROOT)} (separate rocprofv3 --pmc passes of this configuration, recorded {d.get("_recorded", "?")}; not this run)'
        return {k: v for k, v in d.items() if not k.startswith('_')}, src
    return {}, None


def host_launch_us(model, batches, planted):
    """Host time to ENQUEUE the data-independent part of one step (backbone .. second coarse matching; launch-only, no
    host synchronisation inside) on an idle stream: what one host thread spends per step before the GPU has anything to
    wait for.  With --graphs it is the cost of one replay."""
    import torch
    i0, i1 = batches[0]
    ts = []
    with torch.no_grad():
        for _ in range(4):
            torch.cuda.synchronize()
            t = time.perf_counter()
            if getattr(model, '_graphs', None) is not None and planted is None:
                model({'image0': i0, 'image1': i1})              # replay + the dynamic tail (two host syncs)
            elif planted is None:
                model.forward_static({'image0': i0, 'image1': i1})
            else:
                n = i0.shape[0]
                fc, ff = model._backbone(torch.cat([i0, i1], dim=0))
                c, f = torch.add(planted[0]['c'], fc, alpha=0.0), torch.add(planted[0]['f'], ff, alpha=0.0)
                model.forward_features({'image0': i0, 'image1': i1}, c[:n], f[:n], c[n:], f[n:], static_only=True)
            ts.append(time.perf_counter() - t)
    torch.cuda.synchronize()
    return 1e6 * min(ts[1:])


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args, argv))
    if args.dry_run:
        return dry_run(args)
    rank, local, world = dist_env(args)
    pinned = pin_rank(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)))          # before the first GPU call of this process

    # MIOpen's convolution search results for the bench shapes are shipped with the repo (plain-text user find-db
    # for gfx950), so warm-up looks the algorithms up instead of re-running a multi-minute search; every process
    # works on its own copy (geoformer_amd/miopen.py) and the shipped files are only rewritten by --tune --save-db
    from geoformer_amd import miopen as gf_miopen
    gf_miopen.use_shipped_find_db()
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        import datetime
        dist.init_process_group('nccl', device_id=torch.device('cuda', local), timeout=datetime.timedelta(seconds=600))
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    torch.backends.cudnn.benchmark = args.tune
    from geoformer_amd import _lib
    from geoformer_amd.shard import shard_bounds
    L = _lib.lib()
    tlog = time.perf_counter()

    def log(msg):
        if rank == 0:
            print(f'[bench +{time.perf_counter() - tlog:6.1f}s] {msg}', file=sys.stderr, flush=True)
    if args.coarse_thr is None:
        args.coarse_thr = 0.2 if args.pairs == 'planted' else 0.0          # the reference's threshold (geo_config.py:13)
    if args.fine_thr is None:
        args.fine_thr = 0.1 if args.pairs == 'planted' else 0.0            # geo_config.py:15
    model, W = build_model(args.precision, args.coarse_thr, args.fine_thr, dev)
    if args.graphs:
        model.enable_graphs()
    # static shard: the job is the pair list 0 .. world*steps*batch-1; this rank owns the contiguous block
    # [lo, hi).  A few distinct batches of the block are kept resident and cycled so that HBM holds the inputs
    # before the timed region starts.
    lo, hi = shard_bounds(world * args.steps * args.batch, world, rank)
    assert hi - lo == args.steps * args.batch
    nres = min(args.steps, 4)

    def resident(kind):
        return [synth_pairs(args.batch, seed=lo + i * args.batch, size=args.size, device=dev, kind=kind) for i in range(nres)]
    planted = None
    if args.pairs == 'planted':
        batches = resident('shift')
        planted = [planted_maps(args.batch, 60000 + lo + i, args.size // 8, dev, model.compute_dtype) for i in range(nres)]
    else:
        batches = resident(args.pairs)
    log('model + inputs ready')

    def backbone_ms_per_pair():
        x = torch.cat(batches[0], 0)
        with torch.no_grad():
            model._backbone(x)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                model._backbone(x)
            torch.cuda.synchronize()
        return (time.perf_counter() - t) / 3 / args.batch * 1e3
    bb_ms = backbone_ms_per_pair()
    log(f'backbone {bb_ms:.2f} ms/pair with the shipped MIOpen picks')
    if not args.tune and args.precision != 'fp32' and bb_ms > 2.2 * (args.size / 640.0) ** 2:
        # the shipped find-db did not apply (other batch size / MIOpen build): let MIOpen search once; the result
        # stays in this process's private db copy
        log('slower than the tuned reference (1.7 ms/pair): running the MIOpen search (minutes) ...')
        torch.backends.cudnn.benchmark = True
        bb_ms = backbone_ms_per_pair()
        log(f'backbone {bb_ms:.2f} ms/pair after the search')

    dependent = not args.independent
    elapsed, pipes, step, rep_times = measure(model, batches, args.steps, args.warmup, args.streams, dev, dist, log, b'k1_conf', L, planted,
                                              dependent, max(1, args.repeats))
    log('timed region done')
    nstreams = 1 if pipes.serial else pipes.n
    res_rows = [pipes.results[i] for i in range(args.warmup, args.warmup + args.steps)]
    Ms, Mfs = [r[0] for r in res_rows], [r[1] for r in res_rows]
    nidx = res_rows[-1][2]

    def collect(tag):
        ms, cnt, work = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0)
        L.gf_profile_collect(tag.encode(), ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(work))
        return ms.value, cnt.value, work.value
    timed_conf = collect('k1_conf')
    L.gf_profile_filter(None)
    # Per-kernel durations for the roofline entries: with several host pipelines the HIP-event span of a launch
    # also covers kernels of the other streams that share the GPU, so the same steps are replayed on ONE stream
    # right after the timed region (same inputs, same code, profiling events on every tagged launch) and those
    # uncontended durations are reported - they are what `rocprofv3 --kernel-trace -- python3 bench.py --streams 1`
    # (profiles/) shows per kernel.  k1_conf's in-region average is kept next to it.
    replay_steps = min(2, args.steps)
    for i in range(replay_steps):
        step(i)
    torch.cuda.synchronize()
    solo = {tag: collect(tag) for tag, _, _, _ in ROOFLINE_TAGS}
    L.gf_profile_enable(0)
    host_us = host_launch_us(model, batches, planted)
    own_times = [float(v) for v in rep_times]                   # this rank's own regions (before the MAX over ranks)
    if dist is not None:
        t = torch.tensor(rep_times, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                # every region: the slowest rank's time
        rep_times = [float(v) for v in t]
        elapsed = rep_times[0]
        rows = [None] * world
        dist.all_gather_object(rows, (sum(Ms), sum(Mfs), len(Ms), rank, lo, hi, own_times, host_us, len(pinned) if pinned else 0))
    else:
        rows = [(sum(Ms), sum(Mfs), len(Ms), rank, lo, hi, own_times, host_us, 0)]
    if rank != 0:
        pipes.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    pairs = args.batch * args.steps * world
    Lc = (args.size // 8) ** 2
    e = 4 if args.precision == 'fp32' else 2
    algo_bytes = args.batch * (2 * Lc * 256 * e + Lc * Lc * 4)     # per k1_conf launch (170.4 MB/pair-call at e=2)
    if solo['k1_conf'][1]:
        assert abs(solo['k1_conf'][2] / solo['k1_conf'][1] - algo_bytes) < 1.0
    run_cfg = {'precision': args.precision, 'batch': args.batch, 'size': args.size, 'pairs': args.pairs,
               'coarse_thr': args.coarse_thr, 'fine_thr': args.fine_thr}
    pmc, pmc_src = load_pmc(args.pmc_file, run_cfg)
    entries = []
    K = int(nidx[:, 0].float().mean()) if nidx is not None else None
    for tag, bound, hot, what in ROOFLINE_TAGS:
        tot_ms, cnt, work = solo[tag]
        if cnt == 0 or tot_ms <= 0:
            continue
        if tag == 'k4_self_attention':            # key counts are device-side: 4 L K C flops per image, 2 images per pair
            work = cnt * 4.0 * (2 * args.batch) * Lc * (K or 0) * 256
        peak = HBM_PEAK_GBPS if bound == 'hbm' else MFMA_PEAK_TFLOPS[args.precision]
        ach = work / (tot_ms * 1e-3) / (1e9 if bound == 'hbm' else 1e12)
        p = pmc.get(tag, {})
        entries.append({'kernel': f'{tag}: {what}', 'tag': tag, 'bound': bound, 'hot_path': hot, 'achieved': ach, 'peak': peak,
                        'unit': 'GB/s' if bound == 'hbm' else 'TFLOP/s', 'frac': ach / peak,
                        'traffic': p.get('hbm_bytes_per_launch'), 'mfma_util_pmc': p.get('mfma_util'),
                        'traffic_source': pmc_src if p else None,
                        'launches': cnt, 'launches_per_step': cnt / replay_steps, 'avg_launch_ms': tot_ms / cnt,
                        'total_ms_per_step': tot_ms / replay_steps,
                        ('algorithmic_bytes_per_launch' if bound == 'hbm' else 'algorithmic_flops_per_launch'): work / cnt})
    # the dominant kernel = the hot-path tag with the largest time per step (k1_unit spans k1_stats + k1_conf: not a kernel)
    hot_entries = [x for x in entries if x['hot_path'] and x['tag'] != 'k1_unit']
    dominant = dict(max(hot_entries, key=lambda x: x['total_ms_per_step'])) if hot_entries else {}
    dominant['measured'] = ('HIP events on the launch stream; single-stream replay of the timed steps (equals the kernel '
                            'durations of a --streams 1 rocprofv3 trace, profiles/)')
    if timed_conf[1]:
        for x in entries:
            if x['tag'] == 'k1_conf':
                x['avg_launch_ms_in_timed_region'] = timed_conf[0] / timed_conf[1]
    tot_M, tot_Mf, tot_steps = (sum(r[k] for r in rows) for k in range(3))
    res = {
        'metric': 'image-pairs/sec (640x640)', 'value': pairs / elapsed, 'unit': 'image-pairs/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {'fp16': 'f16', 'bf16': 'bf16', 'fp32': 'f32'}[args.precision], 'data': 'synthetic',
        # the spread of the measurement: `repeats` timed regions of K steps each, back to back; [0] is `value`
        'repeats': {'n': len(rep_times), 'pairs_per_s': [pairs / t for t in rep_times], 'min': pairs / max(rep_times),
                    'median': pairs / sorted(rep_times)[len(rep_times) // 2], 'max': pairs / min(rep_times)},
        'repeats_min': pairs / max(rep_times), 'repeats_median': pairs / sorted(rep_times)[len(rep_times) // 2],
        'repeats_max': pairs / min(rep_times),     # top-level scalars (a parser that keeps only scalar keys sees the spread too)
        'shard_plan': sorted([list(r[3:6]) for r in rows]),
        # rank skew: every rank's OWN time for the region `value` is quoted on (the line's ms_per_step is their MAX),
        # its host launch time per step and the number of host cores it is pinned to (0 = not pinned: a single rank)
        'per_rank': [{'rank': r[3], 'ms_per_step': 1e3 * r[6][0] / args.steps, 'host_launch_us_per_step': r[7], 'pinned_cores': r[8]}
                     for r in sorted(rows, key=lambda r: r[3])],
        'rank_pin_rule': PIN_RULE,                  # rank 0's rule for its host cores (rank_cpu_set); None = not pinned
        'rank_ms_per_step_min': 1e3 * min(r[6][0] for r in rows) / args.steps,
        'rank_ms_per_step_max': 1e3 * max(r[6][0] for r in rows) / args.steps,     # [rank, first pair, one past the last pair] of the job's pair list
        'config': {'workload': f'batched inference, synthetic {args.size}x{args.size} pairs (BASELINE configs[4]: static shard of the '
                               f'pair list, {args.steps * args.batch} pairs per GPU), ' +
                               ('NOMINAL LOAD: ResNet-FPN backbone on the images, matching path on planted-correspondence feature maps '
                                '(image-1 maps = image-0 maps shifted by one coarse cell + noise)' +
                                (" fed THROUGH that step's backbone output (every map the matching path reads = planted + 0 x backbone output, "
                                 'one fused pass per map: the forward\'s data dependency is kept, the values are the planted ones)'
                                 if dependent else ', backbone output discarded (no data dependency between the two halves of the step)')
                                if args.pairs == 'planted' else
                                f'image1 = {args.pairs} of image0, full forward incl. ResNet-FPN backbone') +
                               f'; closed-form random-init weights; coarse_thr={args.coarse_thr} fine_thr={args.fine_thr}',
                   'pairs_per_gpu_per_step': args.batch, 'global_pairs_per_step': args.batch * world,
                   'coarse_matches_per_pair': tot_M / tot_steps / args.batch, 'fine_matches_per_pair': tot_Mf / tot_steps / args.batch,
                   'inlier_cells_per_pair': K, 'parallelism': f'pair-shard x{world} (no collective)',
                   'host_pipelines_per_gpu': nstreams, 'hip_graphs': bool(args.graphs),
                   'backbone_dependency': bool(dependent) if args.pairs == 'planted' else True, **run_cfg},
        'host_launch_us_per_step': host_us,
        'roofline': dominant,
        'roofline_kernels': entries,
    }
    # the correlation sweep's figures as top-level scalars (north_star: achieved HBM GB/s on the correlation sweep): a parser that keeps
    # only scalar keys sees them too
    for x in entries:
        if x['tag'] in ('k1_conf', 'k1_unit', 'k1_stats', 'enc_layer', 'fine_layer', 'k4_self_attention', 'conv3x3'):
            res[f"{x['tag']}_frac"] = x['frac']
            res[f"{x['tag']}_us_per_launch"] = 1e3 * x['avg_launch_ms']
            res[f"{x['tag']}_achieved_{'GBps' if x['bound'] == 'hbm' else 'TFLOPs'}"] = x['achieved']
    if world == 1 and not args.no_extras:
        res['side_measurements'] = side_measurements(args, model, dev, log, L, batches, planted, pipes)
        ind = res['side_measurements'].get('nominal_independent')
        if ind:                                   # what the two extra elementwise passes of the dependency cost the headline
            res['side_measurements']['dependency_cost_pct'] = 100.0 * (ind['value'] - res['value']) / ind['value']
    if world == 1 and not args.no_extras:
        del model, batches, planted
        torch.cuda.empty_cache()
        res['side_measurements']['hpatches_b1'] = hpatches_b1_measurements(dev, log, L, pipes if pipes.n == 2 and not pipes.serial else None)
        hb = res['side_measurements']['hpatches_b1']
        for k in ('nominal_pairs_per_s', 'nominal_latency_ms_p50', 'light_pairs_per_s', 'light_graphs_pairs_per_s'):
            if k in hb:
                res[f'hpatches_b1_{k}'] = hb[k]
        model = batches = planted = None
    pipes.close()
    if world == 1 and not args.no_extras and not args.no_train:
        torch.cuda.empty_cache()
        res['side_measurements']['train_step'] = train_measurements(dev, log)
        for k, v in res['side_measurements']['train_step'].items():      # top-level scalars as well
            res[f'train_{k}_pairs_per_s'], res[f'train_{k}_ms_per_step'] = v['value'], v['ms_per_step']
    if not args.no_cpu_baseline and world == 1:          # reported on rank 0 at N = 1 only
        res['cpu_baseline'] = cpu_baseline(W, args.coarse_thr, args.fine_thr, args.size, args.pairs)
    print(json.dumps(res), flush=True)
    if args.tune and args.save_db:
        gf_miopen.save_find_db()             # an explicit search: keep its picks for the next process
    if dist is not None:
        dist.destroy_process_group()


def side_measurements(args, model, dev, log, L, batches=None, planted=None, pipes=None):
    """More throughput figures of the same job, N = 1 only (they are not `value`):
      nominal_independent  (when `value` is the nominal load) the headline step WITHOUT the data dependency on the backbone
                          output (the backbone's maps are discarded, the planted maps are read in place: round 3's headline);
                          `value` pays two extra elementwise passes for the dependency (`dependency_cost_pct`);
      matching_path_only  the hot path alone: `forward_features` on resident planted-correspondence feature maps (no
                          backbone in the step) at the nominal load - what the HIP kernels of SURVEY section 8(a) take;
      matching_path_match_only  the same with the opt-in match-only K1 (conf matrices not materialised);
      light_load          (when `value` is the nominal load) the full forward on homography image pairs with thresholds 0:
                          random-init weights give M ~ 360 matches and K ~ 60 inlier cells per pair (rounds 1-2's headline);
      nominal_load        (when `value` is NOT the nominal load) backbone on the images + matching path on planted maps,
                          thresholds 0.2 / 0.1;
      parity_mode         the fp32 mode in which coarse indices are bit-exact against the reference's golden vectors."""
    import torch
    out = {}
    steps = max(4, min(args.steps, 100))
    dt = {'fp16': 'f16', 'bf16': 'bf16', 'fp32': 'f32'}[args.precision]

    def summary(el, p, nsteps, nwarm, what):
        rr = [p.results[i] for i in range(nwarm, nwarm + nsteps)]
        nidx = rr[-1][2]
        return {'value': nsteps * args.batch / el, 'unit': 'image-pairs/s', 'steps': nsteps, 'ms_per_step': 1e3 * el / nsteps,
                'pairs': what, 'dtype': dt, 'coarse_matches_per_pair': sum(r[0] for r in rr) / len(rr) / args.batch,
                'fine_matches_per_pair': sum(r[1] for r in rr) / len(rr) / args.batch,
                'inlier_cells_per_pair': int(nidx[:, 0].float().mean()) if nidx is not None else None}
    nominal = args.pairs == 'planted'
    mn = model if nominal else build_model(args.precision, 0.2, 0.1, dev)[0]       # the reference's thresholds (geo_config.py:13,15)
    if nominal and planted is not None:
        el, p, _, _ = measure(model, batches, steps, 3, args.streams, dev, None, log, planted=planted, dependent=bool(args.independent), pipes=pipes)
        key = 'nominal_dependent' if args.independent else 'nominal_independent'
        out[key] = summary(el, p, steps, 3, 'the headline step ' + ('WITH' if args.independent else 'WITHOUT') + " the data dependency on that "
                           "step's backbone output")
        log(f"{key}: {out[key]['value']:.1f} pairs/s")
    feats = [planted_features(args.batch, 60000 + i, args.size // 8, device=dev, dtype=mn.compute_dtype) for i in range(2)]
    zero = torch.zeros(args.batch, 1, args.size, args.size, device=dev)

    def feat_step(i):
        with torch.no_grad():
            return mn.forward_features({'image0': zero, 'image1': zero}, *feats[i % 2])
    el, p = measure_fn(feat_step, steps, 3, args.streams, dev, pipes)
    out['matching_path_only'] = summary(el, p, steps, 3, 'matching path only (forward_features on resident planted-correspondence '
                                        'feature maps: no backbone in the step), coarse_thr 0.2, fine_thr 0.1')
    log(f"matching path only: {out['matching_path_only']['value']:.1f} pairs/s ({out['matching_path_only']['ms_per_step']:.2f} ms per "
        f"{args.batch} pairs) at M = {out['matching_path_only']['coarse_matches_per_pair']:.0f}")
    if args.precision != 'fp32':
        # match-only K1 (opt-in, geoformer_cfg['materialize_conf'] = False): conf_matrix / dect_conf_matrix are not written
        # (2 x 164 MB per pair), matches bit-identical; a side measurement only - the contract mode is what `value` runs
        mn.coarse_matching.materialize_conf = False
        el, p = measure_fn(feat_step, steps, 3, args.streams, dev, pipes)
        mn.coarse_matching.materialize_conf = True
        out['matching_path_match_only'] = summary(el, p, steps, 3, 'matching path only, match-only K1 (conf matrices not materialised), '
                                                  'coarse_thr 0.2, fine_thr 0.1')
        log(f"matching path, match-only K1: {out['matching_path_match_only']['value']:.1f} pairs/s")
    if nominal:
        ml, _ = build_model(args.precision, 0.0, 0.0, dev)
        if args.graphs:
            ml.enable_graphs()
        homo = [synth_pairs(args.batch, seed=i * args.batch, size=args.size, device=dev, kind='homography') for i in range(2)]
        el, p, _, _ = measure(ml, homo, steps, 3, args.streams, dev, None, log, pipes=pipes)
        out['light_load'] = summary(el, p, steps, 3, 'full forward incl. backbone on homography image pairs, thresholds 0 (random-init '
                                    'weights: few matches; the headline workload of rounds 1-2)')
        del ml, homo
        log(f"light load: {out['light_load']['value']:.1f} pairs/s at M = {out['light_load']['coarse_matches_per_pair']:.0f}")
    else:
        imgs = [synth_pairs(args.batch, seed=50000 + i * args.batch, size=args.size, device=dev, kind='shift') for i in range(2)]
        pm = [planted_maps(args.batch, 60000 + i, args.size // 8, dev, mn.compute_dtype) for i in range(2)]
        el, p, _, _ = measure(mn, imgs, steps, 3, args.streams, dev, None, log, planted=pm, pipes=pipes)
        del pm
        out['nominal_load'] = summary(el, p, steps, 3, 'backbone on the images + matching path on planted-correspondence feature maps '
                                      '(shift by one coarse cell + noise), coarse_thr 0.2, fine_thr 0.1')
        log(f"nominal load: {out['nominal_load']['value']:.1f} pairs/s at M = {out['nominal_load']['coarse_matches_per_pair']:.0f}, "
            f"K = {out['nominal_load']['inlier_cells_per_pair']}")
        del imgs
    del feats
    if not nominal:
        del mn
    torch.cuda.empty_cache()
    if args.precision != 'fp32':
        m32, _ = build_model('fp32', args.coarse_thr, args.fine_thr, dev)
        s32 = max(4, min(args.steps, 10))
        if nominal:
            imgs = [synth_pairs(args.batch, seed=i * args.batch, size=args.size, device=dev, kind='shift') for i in range(2)]
            f32 = [planted_maps(args.batch, 60000 + i, args.size // 8, dev, torch.float32) for i in range(2)]
            el, p, _, _ = measure(m32, imgs, s32, 2, 1, dev, None, log, planted=f32)
        else:
            imgs = [synth_pairs(args.batch, seed=i * args.batch, size=args.size, device=dev, kind=args.pairs) for i in range(2)]
            el, p, _, _ = measure(m32, imgs, s32, 2, 1, dev, None, log)
        p.close()
        out['parity_mode'] = {'value': s32 * args.batch / el, 'unit': 'image-pairs/s', 'steps': s32, 'ms_per_step': 1e3 * el / s32,
                              'dtype': 'f32', 'note': 'the same workload in fp32 storage with exact-fp32 MFMA (v_mfma_f32_32x32x2_f32); '
                                                      'unfused fp32 backbone, MIOpen immediate mode'}
        del m32
        torch.cuda.empty_cache()
        log(f"parity mode (fp32): {out['parity_mode']['value']:.1f} pairs/s")
    return out


def hpatches_b1_measurements(dev, log, L, pipes=None, steps=60, warmup=6):
    """BASELINE configs[1] (the HPatches loop: `eval_Hpatches.py` -> `hpatches_helper.py:167-182`, one pair per matcher call, `imsize: 480` =
    shorter side 480 and both sides floored to a multiple of 8, `data_io.py:16-26`): batch 1, UNEQUAL shapes 480x640 against 480x608 (coarse
    grids 60x80 = 4800 and 60x76 = 4560 tokens), bf16 features, the reference's thresholds 0.2 / 0.1.  Synthetic inputs, closed-form weights:
      nominal  backbone on the two images (one call each: the shapes differ, full_model.py:58-59) + matching path on planted-correspondence
               maps fed through that backbone output (the headline's construction at batch 1): thousands of matches per pair;
      light    the full forward on a textured image pair under a homography (thresholds 0: random-init weights give few matches), eager and
               with hipGraph replay of the static part (`GeoFormer.enable_graphs`: the loop repeats its shapes).
    Per workload: pairs/s over two host pipelines (the loop's throughput when pairs are independent), per-pair latency on ONE stream with a
    host synchronisation per pair (what `match_time` of hpatches_helper.py:174-182 records), host time to enqueue the static part, and the
    per-launch time of the K9 / K1 kernels at these shapes (4800 x 4560 tiles on 256 CUs).  Side measurements, never `value`."""
    import torch
    out = {'shapes': [[480, 640], [480, 608]], 'dtype': 'bf16', 'batch': 1}
    box = [pipes]
    hw0, hw1, g0, g1 = (480, 640), (480, 608), (60, 80), (60, 76)
    gen = torch.Generator().manual_seed(4242)

    def textured(hw, shift):
        base = torch.rand(1, 1, hw[0] // 8 + 4, hw[1] // 8 + 4, generator=gen)
        img = torch.nn.functional.interpolate(base, size=(hw[0] + 32, hw[1] + 32), mode='bicubic', align_corners=True)
        return img[:, :, shift:shift + hw[0], shift:shift + hw[1]].clamp(0, 1).contiguous().to(dev)

    def planted(seed):
        g = torch.Generator().manual_seed(seed)
        big = torch.randn(1, 256, g0[0] + 1, g0[1] + 1, generator=g) * 0.5
        bigf = torch.randn(1, 128, 4 * (g0[0] + 1), 4 * (g0[1] + 1), generator=g)
        c0, f0 = big[:, :, :g0[0], :g0[1]], bigf[:, :, :4 * g0[0], :4 * g0[1]]
        c1 = big[:, :, 1:1 + g1[0], 1:1 + g1[1]] + 0.35 * torch.randn(1, 256, *g1, generator=g)
        f1 = bigf[:, :, 4:4 + 4 * g1[0], 4:4 + 4 * g1[1]] + 0.35 * torch.randn(1, 128, 4 * g1[0], 4 * g1[1], generator=g)
        return [t.to(device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last) for t in (c0, f0, c1, f1)]

    def run(tag, step, what, model=None):
        with torch.no_grad():
            if model is not None:
                model.concurrent_backbones = True      # latency: the two backbone calls of the unequal-shape pair on two streams
            r = step(0)
            torch.cuda.synchronize()
            M = len(r['b_ids'])
            # latency: one stream, one host synchronisation per pair
            lat = []
            for i in range(warmup + steps):
                torch.cuda.synchronize()
                t = time.perf_counter()
                step(i)
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t)
            lat = sorted(lat[warmup:])
            if model is not None:
                model.concurrent_backbones = False     # throughput on two pipelines: one stream per pipeline gives more pairs per second
            # throughput: two host pipelines (the headline's own threads and streams when it ran on two)
            rates = []
            for _ in range(3):            # three back-to-back regions (the second pipeline half a step late), the median reported (a region
                el, p = measure_fn(step, steps, warmup, 2, dev, box[0], stagger=0.5 * lat[len(lat) // 2])   # in which they stay in phase runs at half the rate: Pipelines.run)
                box[0] = p
                rates.append(steps / el)
        out[f'{tag}_pairs_per_s'] = sorted(rates)[1]
        out[f'{tag}_pairs_per_s_repeats'] = rates
        out[f'{tag}_latency_ms_p50'] = 1e3 * lat[len(lat) // 2]
        out[f'{tag}_latency_ms_p99'] = 1e3 * lat[min(len(lat) - 1, int(0.99 * len(lat)))]
        out[f'{tag}_coarse_matches_per_pair'] = M
        out[f'{tag}_what'] = what
        log(f"hpatches_b1 {tag}: {out[f'{tag}_pairs_per_s']:.1f} pairs/s on two pipelines, latency p50 {out[f'{tag}_latency_ms_p50']:.2f} ms "
            f"p99 {out[f'{tag}_latency_ms_p99']:.2f} ms, M = {M}")

    i0, i1 = textured(hw0, 8), textured(hw1, 16)
    mn, _ = build_model('bf16', 0.2, 0.1, dev)
    pl = [planted(70000 + i) for i in range(2)]

    def nominal_step(i):
        (fc0, ff0), (fc1, ff1) = mn._backbone_unequal(i0, i1)             # what GeoFormer.forward does with an unequal-shape pair
        c0, f0, c1, f1 = pl[i % 2]
        return mn.forward_features({'image0': i0, 'image1': i1}, torch.add(c0, fc0, alpha=0.0), torch.add(f0, ff0, alpha=0.0),
                                   torch.add(c1, fc1, alpha=0.0), torch.add(f1, ff1, alpha=0.0))
    run('nominal', nominal_step, 'backbone on 480x640 / 480x608 images + matching path on planted maps through the backbone output, thresholds 0.2 / 0.1; '
        'latency with the two backbone calls on two streams, throughput with one stream per pipeline', model=mn)
    # per-launch times of the hot-path kernels at these shapes (one stream, HIP events on every tagged launch)
    L.gf_profile_filter(None)
    L.gf_profile_enable(1)
    with torch.no_grad():
        for i in range(4):
            nominal_step(i)
    torch.cuda.synchronize()
    kern = {}
    for tag in ('enc_layer', 'enc_kv_state', 'k1_stats', 'k1_conf', 'k1_unit', 'fine_layer', 'k4_self_attention', 'k5_window_attention', 'conv3x3'):
        ms, cnt, work = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0)
        L.gf_profile_collect(tag.encode(), ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(work))
        if cnt.value:
            kern[tag] = {'us_per_launch': 1e3 * ms.value / cnt.value, 'launches_per_pair': cnt.value / 4.0, 'ms_per_pair': ms.value / 4.0}
            if tag in ('enc_layer', 'enc_kv_state', 'k1_stats', 'fine_layer', 'conv3x3') and work.value > 0:
                kern[tag]['frac_of_mfma_peak'] = work.value / (ms.value * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS['bf16']
            if tag in ('k1_conf', 'k1_unit', 'k5_window_attention') and work.value > 0:
                kern[tag]['frac_of_hbm_peak'] = work.value / (ms.value * 1e-3) / 1e9 / HBM_PEAK_GBPS
    L.gf_profile_enable(0)
    out['kernels_nominal'] = kern
    ts = []
    with torch.no_grad():
        for _ in range(4):
            torch.cuda.synchronize()
            t = time.perf_counter()
            (fc0, ff0), (fc1, ff1) = mn._backbone_unequal(i0, i1)
            c0, f0, c1, f1 = pl[0]
            mn.forward_features({'image0': i0, 'image1': i1}, torch.add(c0, fc0, alpha=0.0), torch.add(f0, ff0, alpha=0.0),
                                torch.add(c1, fc1, alpha=0.0), torch.add(f1, ff1, alpha=0.0), static_only=True)
            ts.append(time.perf_counter() - t)
    torch.cuda.synchronize()
    out['nominal_host_launch_us_per_pair'] = 1e6 * min(ts[1:])
    del mn, pl
    torch.cuda.empty_cache()
    ml, _ = build_model('bf16', 0.0, 0.0, dev)
    j1 = synth_rect_pair(hw0, hw1, 77, dev)
    run('light', lambda i: ml({'image0': j1[0], 'image1': j1[1]}),
        'full forward on a textured 480x640 image and its homography warp at 480x608, thresholds 0; latency with the two backbone calls on two streams, '
        'throughput with one stream per pipeline', model=ml)
    ml.enable_graphs()
    run('light_graphs', lambda i: ml({'image0': j1[0], 'image1': j1[1]}), 'the same with hipGraph replay of the static part')
    del ml
    torch.cuda.empty_cache()
    if pipes is None and box[0] is not None:
        box[0].close()
    return out


def synth_rect_pair(hw0, hw1, seed, device):
    """One textured image of hw0 and its warp under a small random homography, resampled at hw1 (the HPatches loop's unequal shapes)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    H0, W0 = hw0
    base = torch.rand(1, 1, H0 // 8 + 2, W0 // 8 + 2, generator=g)
    img = torch.nn.functional.interpolate(base, size=(H0 + 16, W0 + 16), mode='bicubic', align_corners=True)
    img = (img + 0.15 * torch.rand(1, 1, H0 + 16, W0 + 16, generator=g)).clamp(0, 1)
    image0 = img[:, :, 8:8 + H0, 8:8 + W0].contiguous()
    th = (torch.rand(1, generator=g) - 0.5) * 0.1
    theta = torch.tensor([[[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]]) + torch.cat([torch.stack([torch.cos(th) - 1, -torch.sin(th), th]),
                                                                          torch.stack([torch.sin(th), torch.cos(th) - 1, -th])]).view(1, 2, 3) * 0.5
    grid = torch.nn.functional.affine_grid(theta, (1, 1, hw1[0], hw1[1]), align_corners=True) * 0.92
    image1 = torch.nn.functional.grid_sample(img, grid, mode='bilinear', padding_mode='border', align_corners=True)
    return image0.to(device), image1.contiguous().to(device)


def train_measurements(dev, log, steps=6, warmup=4):
    """The training step of BASELINE configs[2] / [3] at their per-GPU sizes (VERDICT r04 #6a), N = 1: mixed bf16 (fp32 master weights,
    `train_depth_geoformer.py:117-119`) with the fused HIP coarse loss and the HIP forward + backward Functions
    and, since the end of round 5, the backbone's stride-1 3x3 convolutions forward + backward-data on K10
    (`TrainStep(precision='bf16', fused_coarse_loss=True, hip_backward=True, hip_conv=True)`): supervision -> forward -> loss -> backward -> clipped
    AdamW step (`lightning_homo_geoformer.py:69-107`), batches made outside the timed steps.
      configs2_homo:      640 x 480 synthetic homography pairs, batch 4 per GPU (batch 32 over 8 GPUs; `homo_trainval_640.py:5`)
      configs3_megadepth: 640 x 640 MegaDepth-style pairs (depth + pose supervision, padding masks, per-image scales), batch 8 per GPU
    Closed-form random-init weights, thresholds 0 / 0 (untrained weights give no match above 0.2); a side measurement, never `value`.
    Four warm-up steps: steps 2 and 3 of a fresh model still run 40-50 % slow (the caching allocator and the workspaces growing to the step's
    peak; 0.32 / 0.31 s against 0.19 in `tools/train_profile.py`), which with two warm-up steps put 232 ms into a line whose steady state is 190."""
    import torch
    from geoformer_amd.model.cvpr_ds_config import get_default_cfg
    from geoformer_amd.model.full_model import GeoFormer
    from geoformer_amd.model.geo_config import get_cfg_model
    from geoformer_amd.weights import deterministic_init_
    from geoformer_amd.train import TrainStep, synthetic_homography_batch, synthetic_megadepth_batch
    out = {}
    for key, make, hw, batch in (('configs2_homo', synthetic_homography_batch, (480, 640), 4),
                                 ('configs3_megadepth', synthetic_megadepth_batch, (640, 640), 8)):
        g = get_cfg_model()
        g.update(coarse_thr=0.0, fine_thr=0.0, precision='fp32')
        model = deterministic_init_(GeoFormer(get_default_cfg(), g)).to(dev)
        step = TrainStep(model, batch_size=batch, fused_coarse_loss=True, precision='bf16', hip_backward=True, hip_conv=True)
        base = [make(batch, hw, seed=900 + i, device=dev) for i in range(2)]         # two resident batches, taken in turn (the step writes its
        data = [dict(base[i % 2]) for i in range(warmup + steps)]                    # supervision and outputs into the dict it is given: a fresh dict per step)
        losses = []
        for i in range(warmup):
            losses.append(float(step(data[i])))
        torch.cuda.synchronize()
        allocs0 = torch.cuda.memory_stats(dev).get('num_device_alloc', 0)
        t0 = time.perf_counter()
        host = []
        for i in range(warmup, warmup + steps):
            th = time.perf_counter()
            losses.append(step(data[i]))
            host.append(1e3 * (time.perf_counter() - th))          # host time of the step's enqueue (the device runs behind)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        allocs = torch.cuda.memory_stats(dev).get('num_device_alloc', 0) - allocs0
        losses = [float(v) for v in losses]
        out[key] = {'value': steps * batch / el, 'unit': 'image-pairs/s', 'ms_per_step': 1e3 * el / steps, 'steps': steps, 'warmup': warmup,
                    'batch_per_gpu': batch, 'image_hw': list(hw), 'precision': 'mixed bf16 (fp32 master weights)', 'hip_backward': True, 'hip_conv': True,
                    'fused_coarse_loss': True, 'losses': losses, 'finite': all(math.isfinite(v) for v in losses),
                    'host_enqueue_ms_per_step': host, 'device_allocations_in_timed_steps': allocs}
        log(f"train step {key}: {out[key]['ms_per_step']:.1f} ms at batch {batch} = {out[key]['value']:.1f} pairs/s, losses {losses[0]:.3f} -> {losses[-1]:.3f}; "
            f"host enqueue {' '.join(f'{h:.0f}' for h in host)} ms, {allocs} device allocations inside the timed steps")
        del step, model, data
        torch.cuda.empty_cache()
    return out


def measure_fn(step, steps, warmup, nstreams, dev, pipes=None, stagger=0.0):
    """`measure` for an arbitrary step function (side measurements): warm-up, then `steps` timed steps on the pipelines (`stagger`: Pipelines.run)."""
    import torch
    if pipes is None:
        pipes = Pipelines(step, max(1, min(nstreams, steps)), dev)
    else:
        pipes.step_fn, pipes.results = step, {}
    step(0)
    torch.cuda.synchronize()
    for w in range(pipes.n):
        pipes.run_single(w, w)
    pipes.run(0, warmup, stagger)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipes.run(warmup, steps, stagger)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, pipes


if __name__ == '__main__':
    main()

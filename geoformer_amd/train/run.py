"""Training entry point, the counterpart of lightning/train_homo_geoformer.py:61-130 without Lightning:

    python -m geoformer_amd.train.run --steps 20 --batch 4 --size 480 640
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m geoformer_amd.train.run ...

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the environment), DDP over RCCL with SyncBatchNorm and
find_unused_parameters=True, AdamW + linear warm-up + MultiStepLR with the canonical lr/batch scaling.  The
Oxford-Paris / MegaDepth images are not available offline: batches are synthetic homography pairs (texture +
random 4-corner warp) carrying the same keys the reference's HomoDataset provides (`H_0to1`, `H_1to0`,
`dataset_name`); a real dataset only has to yield those keys.
"""
import argparse
import os
import time

import torch


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--batch', type=int, default=4, help='pairs per GPU (reference: 32 over 8 GPUs)')
    ap.add_argument('--size', type=int, nargs=2, default=(480, 640), metavar=('H', 'W'))
    ap.add_argument('--steps-per-epoch', type=int, default=1000)
    ap.add_argument('--ckpt', default=None, help='initial state dict (geoformer.ckpt)')
    ap.add_argument('--save', default=None)
    ap.add_argument('--coarse-thr', type=float, default=0.2)
    ap.add_argument('--fused-coarse-loss', action='store_true',
                    help='coarse focal losses through the HIP kernels (fp16 operands; tolerances in TrainStep.__doc__)')
    ap.add_argument('--force-ddp', action='store_true', help='wrap in DDP/SyncBatchNorm even at world size 1')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'], help="'bf16': mixed precision (TrainStep(precision='bf16'))")
    ap.add_argument('--hip-backward', action='store_true', help='K2 / K3 / K8 forward and backward in HIP (needs --precision bf16)')
    ap.add_argument('--global-batch-seed', type=int, default=None,
                    help='data-parallel check mode: step `it` draws ONE global batch (seed + it) of world x batch pairs and every rank '
                         'takes its own slice of it - the same global batches whatever the world size')
    ap.add_argument('--dup', type=int, default=1,
                    help='with --global-batch-seed: the global batch is a base batch of (world x batch / dup) pairs repeated dup times')
    ap.add_argument('--lr', type=float, default=None, help='canonical lr override (check mode)')
    ap.add_argument('--no-clip', action='store_true', help='no gradient clipping (check mode)')
    ap.add_argument('--report', default=None,
                    help='rank 0 writes a JSON report: per-step losses averaged over the ranks, whether the parameters of all ranks '
                         'are bit-identical after the last step, and a float64 checksum per parameter')
    args = ap.parse_args(argv)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank, local = int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    ddp = world > 1 or args.force_ddp
    if ddp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        torch.distributed.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from ..model.cvpr_ds_config import get_default_cfg
    from ..model.full_model import GeoFormer
    from ..model.geo_config import get_cfg_model
    from ..weights import deterministic_init_
    from .trainer import TrainStep, synthetic_homography_batch
    gcfg = get_cfg_model()
    gcfg.update(coarse_thr=args.coarse_thr, precision='fp32')
    model = GeoFormer(get_default_cfg(), gcfg)
    if args.ckpt:
        sd = torch.load(args.ckpt, map_location='cpu')
        model.load_state_dict(sd.get('state_dict', sd), strict=False)
    else:
        deterministic_init_(model)
    model.to(dev)
    tcfg = {}
    if args.lr is not None:
        tcfg.update(canonical_lr=args.lr, warmup_step=0)
    if args.no_clip:
        tcfg.update(gradient_clipping=0.0)
    step = TrainStep(model, trainer_cfg=tcfg or None, batch_size=args.batch, distributed=ddp, fused_coarse_loss=args.fused_coarse_loss,
                     precision=args.precision, hip_backward=args.hip_backward)
    t0 = t1 = time.perf_counter()
    losses = []
    for it in range(args.steps):
        if it == 1:                               # the first step carries MIOpen's algorithm search
            torch.cuda.synchronize()
            t1 = time.perf_counter()
        if args.global_batch_seed is None:
            batch = synthetic_homography_batch(args.batch, tuple(args.size), seed=1000 * rank + it, device=dev)
        else:
            batch = rank_slice_of_global_batch(args.batch, world, rank, tuple(args.size), args.global_batch_seed + it, args.dup, dev)
        loss = step(batch)
        losses.append(loss.detach().double().reshape(1))
        if (it + 1) % args.steps_per_epoch == 0:
            step.epoch_end()
        if rank == 0:
            s = batch['loss_scalars']
            print(f'step {it:4d} loss {float(loss):.4f} (c {float(s["loss_c"]):.4f} d {float(s["loss_d"]):.4f} f {float(s["loss_f"]):.4f}) '
                  f'lr {step.optimizer.param_groups[0]["lr"]:.2e} matches {len(batch["b_ids"])} gt {int(batch["conf_matrix_gt"].sum())}',
                  flush=True)
    torch.cuda.synchronize()
    if rank == 0:
        now = time.perf_counter()
        rate = (args.steps - 1) * args.batch * world / (now - t1) if args.steps > 1 else 0.0
        print(f'{args.steps} steps in {now - t0:.1f} s; {rate:.2f} pairs/s over {world} GPU(s) after the first step')
        if args.save:
            torch.save({'state_dict': step.model.state_dict()}, args.save)
    if args.report:
        write_report(args.report, step.model, losses, rank, world, ddp)
    if ddp:
        torch.distributed.destroy_process_group()
    return float(loss)


def rank_slice_of_global_batch(batch, world, rank, hw, seed, dup, device):
    """Rank `rank`'s `batch` pairs of the global batch `seed`: world x batch pairs = a base batch of (world x batch / dup) pairs
    repeated `dup` times.  The same global batch for every world size that divides it: what makes a 2-rank run comparable with
    the 1-rank run on the concatenated batch (tests/test_multi_gpu.py, tests/test_train.py)."""
    from .trainer import synthetic_homography_batch
    total = batch * world
    if total % dup:
        raise ValueError('--dup must divide world x batch')
    base = synthetic_homography_batch(total // dup, hw, seed=seed, device=device)
    out = {}
    lo, hi = rank * batch, (rank + 1) * batch
    for k, v in base.items():
        if torch.is_tensor(v):
            out[k] = torch.cat([v] * dup, 0)[lo:hi].contiguous()
        else:
            out[k] = (list(v) * dup)[lo:hi]
    return out


def write_report(path, model, losses, rank, world, ddp):
    """Rank 0 writes {losses (mean over ranks per step), in_sync (parameters AND buffers of all ranks bit-identical), params
    (float64 sum per parameter)}.  Bit-identity is checked on the raw bytes: every rank's fp32 parameters are gathered as int32."""
    import json
    flat = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()] +
                     [b.detach().float().reshape(-1) for b in model.buffers() if b.dtype.is_floating_point])
    lo = torch.cat(losses)
    in_sync = True
    if ddp and world > 1:
        bits = flat.view(torch.int32)
        ref = bits.clone()
        torch.distributed.broadcast(ref, 0)
        same = torch.tensor([int(torch.equal(ref, bits))], device=flat.device)
        torch.distributed.all_reduce(same, op=torch.distributed.ReduceOp.MIN)
        in_sync = bool(same.item())
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.SUM)
        lo = lo / world
    if rank == 0:
        rep = {'world': world, 'losses': [float(v) for v in lo.cpu()], 'in_sync': in_sync,
               'params': {n: float(p.detach().double().sum()) for n, p in model.named_parameters()},
               'param_l2': float(flat.double().norm())}
        with open(path, 'w') as f:
            json.dump(rep, f)


if __name__ == '__main__':
    main()

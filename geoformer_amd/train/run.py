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
    step = TrainStep(model, batch_size=args.batch, distributed=ddp, fused_coarse_loss=args.fused_coarse_loss)
    t0 = t1 = time.perf_counter()
    for it in range(args.steps):
        if it == 1:                               # the first step carries MIOpen's algorithm search
            torch.cuda.synchronize()
            t1 = time.perf_counter()
        batch = synthetic_homography_batch(args.batch, tuple(args.size), seed=1000 * rank + it, device=dev)
        loss = step(batch)
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
    if ddp:
        torch.distributed.destroy_process_group()
    return float(loss)


if __name__ == '__main__':
    main()

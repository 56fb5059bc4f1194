"""The training step and its schedule, one process per GPU.

  TrainStep.__call__          lightning/lightning_homo_geoformer.py:64-77 (`_trainval_inference`) + backward + step
  build_optimizer/scheduler   model/loftr_src/optimizers/__init__.py:5-42
  warm-up rule                lightning_homo_geoformer.py:45-62 (`optimizer_step`)
  lr / warm-up scaling        lightning/train_homo_geoformer.py:80-86
  DDP                         train_homo_geoformer.py:117-125: DDP(find_unused_parameters=True) + SyncBatchNorm,
                              gradient clipping 0.5 (`TRAINER.GRADIENT_CLIPPING`)

The gradient all-reduce is torch DDP over `torch.distributed` (RCCL on MI355X, gloo in the CPU tests): the
14.19 M parameters are 56.75 MB of fp32 gradients per step, bucketed at DDP's 25 MB default - two or three
ring all-reduces per step, far below the xGMI per-link bandwidth, overlapped with the backward.
"""
import math
from typing import Callable, Optional

import torch
import torch.nn as nn

from .functional import forward_train, fused_coarse_loss_applicable
from .loss import GeoLoss
from .supervision import spvs_coarse, spvs_fine2

DEFAULT_TRAINER_CFG = {       # model/loftr_src/config/default.py:103-165 with the overrides of train_config/loftr_ds_dense.py
    'canonical_bs': 64, 'canonical_lr': 8e-3, 'optimizer': 'adamw', 'adam_decay': 0., 'adamw_decay': 0.1,
    'warmup_type': 'linear', 'warmup_ratio': 0.1, 'warmup_step': 1875, 'scheduler': 'MultiStepLR',
    'scheduler_interval': 'epoch', 'mslr_milestones': [8, 12, 16, 20, 24], 'mslr_gamma': 0.5, 'cosa_tmax': 30,
    'elr_gamma': 0.999992, 'gradient_clipping': 0.5,
}


def scale_trainer_cfg(cfg, world_size, batch_size):
    """TRUE_LR / WARMUP_STEP from the canonical values (train_homo_geoformer.py:80-86)."""
    cfg = dict(DEFAULT_TRAINER_CFG, **(cfg or {}))
    cfg['world_size'] = world_size
    cfg['true_batch_size'] = world_size * batch_size
    cfg['scaling'] = cfg['true_batch_size'] / cfg['canonical_bs']
    cfg['true_lr'] = cfg['canonical_lr'] * cfg['scaling']
    cfg['warmup_step'] = math.floor(cfg['warmup_step'] / cfg['scaling'])
    return cfg


def build_optimizer(model, cfg):
    if cfg['optimizer'] == 'adam':
        return torch.optim.Adam(model.parameters(), lr=cfg['true_lr'], weight_decay=cfg['adam_decay'])
    if cfg['optimizer'] == 'adamw':
        return torch.optim.AdamW(model.parameters(), lr=cfg['true_lr'], weight_decay=cfg['adamw_decay'])
    raise ValueError(f"TRAINER.OPTIMIZER = {cfg['optimizer']} is not a valid optimizer!")


def build_scheduler(cfg, optimizer):
    name = cfg['scheduler']
    if name == 'MultiStepLR':
        return torch.optim.lr_scheduler.MultiStepLR(optimizer, cfg['mslr_milestones'], gamma=cfg['mslr_gamma'])
    if name == 'CosineAnnealing':
        return torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, cfg['cosa_tmax'])
    if name == 'ExponentialLR':
        return torch.optim.lr_scheduler.ExponentialLR(optimizer, cfg['elr_gamma'])
    raise NotImplementedError(name)


def warmup_lr(cfg, global_step) -> Optional[float]:
    """lr to force while global_step < warmup_step (linear warm-up), else None."""
    if global_step >= cfg['warmup_step'] or cfg['warmup_type'] == 'constant':
        return None
    if cfg['warmup_type'] != 'linear':
        raise ValueError(f"Unknown lr warm-up strategy: {cfg['warmup_type']}")
    base = cfg['warmup_ratio'] * cfg['true_lr']
    return base + (global_step / cfg['warmup_step']) * abs(cfg['true_lr'] - base)


class _Core(nn.Module):
    """What DDP wraps: supervision -> forward -> fine supervision -> loss, returning the scalar loss."""

    def __init__(self, model, loss, homography_fn=None, fused=True, amp_dtype=None, channels_last=False):
        super().__init__()
        self.model, self.loss, self.homography_fn, self.fused, self.amp_dtype = model, loss, homography_fn, fused, amp_dtype
        self.channels_last = bool(channels_last)

    def forward(self, batch, backbone_features=None):
        res = tuple(self.model.config['resolution'])
        spvs_coarse(batch, res)
        fused = self.loss.fused_params() if (self.fused and fused_coarse_loss_applicable(self.model, batch)) else None
        dev = batch['image0'].device.type
        # mixed precision = Lightning's precision='bf16' (BASELINE configs[3]): fp32 master parameters, convolutions and GEMMs in
        # bf16 with fp32 accumulation, softmax / LayerNorm / losses in fp32 (torch.autocast's op lists)
        with torch.autocast(device_type=dev, dtype=self.amp_dtype or torch.bfloat16, enabled=self.amp_dtype is not None):
            forward_train(self.model, batch, self.homography_fn, fused_coarse_loss=fused, backbone_features=backbone_features,
                          channels_last=self.channels_last)
        spvs_fine2(batch, res)
        return self.loss(batch)


class TrainStep:
    """One optimisation step per call; `epoch_end()` advances an epoch-interval scheduler.

    `step(batch)` mutates `batch` exactly like the reference's `_trainval_inference` (supervision, forward outputs,
    `loss`, `loss_scalars`): the DDP wrapper is built WITHOUT `device_ids` - the module lives on one device and the
    inputs are created there - because with `device_ids` DDP rebuilds every input dict on the way in and the caller's
    `batch` would never see what the forward wrote.

    `fused_coarse_loss=False` (default) is the parity configuration: both coarse losses are differentiated by autograd
    on materialised fp32 confidence matrices, like the reference.  `True` switches the two coarse losses to the HIP
    kernels `gf_coarse_loss_forward/backward`, which compute in fp16 operands / fp32 accumulation: loss value and `p`
    agree with fp32 autograd on the SAME (fp16-rounded) features to 2e-3, the feature gradients to 2e-2 in norm
    (tests/test_train_gpu.py); against un-rounded fp32 features the loss agrees to 2e-3 and the gradients to 1e-2 in
    norm (measured 3e-4 / 2e-3), also for confident matches (p > 0.95), where the gradient itself is small.

    `precision='bf16'` is the mixed-precision step of BASELINE configs[3] (`--precision bf16` of the reference's Lightning
    trainer, lightning/train_depth_geoformer.py:117-119): fp32 master parameters and optimizer state, the forward under
    torch.autocast(bfloat16); confidence matrices, softmax, LayerNorm and the losses stay fp32.  Use it with
    `fused_coarse_loss=True`, which keeps the two L x S confidence matrices out of the autograd graph altogether.

    `hip_backward=True` (with precision='bf16'): the linears, LayerNorms and activations of every encoder layer (LoFTR coarse and
    fine, Geo) run the HIP kernels forward AND backward (train/hip_autograd.py: gf_linear, gf_linear_wgrad, gf_layernorm_*,
    gf_activation_backward) instead of torch's GEMM / autograd; the coarse layers' linear attention runs K2 forward and
    gf_linear_attention_backward, the fine level's 25-token windows K2's window form and gf_window_linear_attention_backward,
    GeoTransformer's cross attention K5 and gf_window_cross_attention_backward, FineMatching2 K8 and gf_fine_match_backward; the Geo
    SELF-attention core (K4: gf_full_attention_train_forward / _backward since round 6; the Geo layers batched over the images of the step),
    the backbone and the losses other than the fused coarse loss stay on autograd.  Step time at batch 2, 640x640: 0.128 s -> 0.075 s.
    `hip_conv=True` (with precision='bf16'; round 5): the backbone's 3x3 / stride-1 convolutions through `hip_autograd.HipConv3x3` - forward and
    backward-data on K10 (`gf_conv3x3_nhwc`; backward-data = the forward kernel on dY with the transposed, flipped weights), backward-weights on
    K10's weight-gradient kernel (round 6), the stride-1 1x1 convolutions on the K3 engine and the FPN upsampling's backward as a gather
    (round 6); the backbone runs in NHWC (implies channels_last).  MegaDepth-style step at batch 8, 640x640: 0.284 s -> 0.245 s (round 5)
    -> 0.18-0.19 s (round 6).
    Every HIP backward sums its partials in a fixed order since round 6 (K5's backward gathers along the inverse window table; its
    fp32-atomics form is kept behind hip_autograd._GATHER_K5_BACKWARD for comparison); what is NOT bit-reproducible run to run is the
    library's part of the step (train-mode convolutions / BatchNorm of the backbone: loss terms move by ~1 % between identical runs)."""

    def __init__(self, model, trainer_cfg=None, loss_cfg=None, batch_size=1, distributed=False, device_ids=None,
                 homography_fn: Optional[Callable] = None, sparse_spvs=True, fused_coarse_loss=False, precision='fp32',
                 hip_backward=False, channels_last=False, hip_conv=False):
        world = torch.distributed.get_world_size() if distributed else 1
        self.cfg = scale_trainer_cfg(trainer_cfg, world, batch_size)
        if model.precision != 'fp32':
            raise ValueError("training runs the fp32 parameters: GeoFormer.set_precision('fp32')")
        model.train()
        # channels_last: the backbone's weights and its input in NHWC memory format (the arithmetic is the same; MIOpen then runs
        # its NHWC convolution kernels forward and backward); an option, off by default - measured per shape, tools/train_profile.py --cl
        if channels_last:
            model.backbone.to(memory_format=torch.channels_last)
        self.model = model
        if precision not in ('fp32', 'bf16'):
            raise ValueError("TrainStep precision: 'fp32' or 'bf16' (mixed: fp32 master weights, bf16 GEMMs / convolutions)")
        self.precision = precision
        if hip_backward and precision != 'bf16':
            raise ValueError("hip_backward=True (K3 chain forward + backward in HIP) is built for the mixed-16-bit step: precision='bf16'")
        self.hip_backward = bool(hip_backward)
        # hip_conv (round 5; with precision='bf16'): the backbone's 3x3 / stride-1 convolutions - forward and backward-data on K10
        # (hip_autograd.HipConv3x3), backward-weights on the library; implies the NHWC backbone
        if hip_conv and precision != 'bf16':
            raise ValueError("hip_conv=True is built for the mixed-16-bit step: precision='bf16'")
        self.hip_conv = bool(hip_conv)
        if hip_conv and not channels_last:
            channels_last = True
            model.backbone.to(memory_format=torch.channels_last)
        core = _Core(model, GeoLoss(loss_cfg, model.config['match_coarse'].get('match_type', 'dual_softmax'), sparse_spvs),
                     homography_fn, fused_coarse_loss, torch.bfloat16 if precision == 'bf16' else None, channels_last)
        if distributed:
            if next(model.parameters()).is_cuda:      # torch's SyncBatchNorm is device-only; the gloo/CPU tests keep local BN
                core = nn.SyncBatchNorm.convert_sync_batchnorm(core)
            self.model = core.model
            if device_ids is not None:
                raise ValueError('TrainStep builds DDP without device_ids (see the class docstring): call '
                                 'torch.cuda.set_device(local_rank) and move the model there instead')
            core = nn.parallel.DistributedDataParallel(core, find_unused_parameters=True)
        self.core = core
        self.optimizer = build_optimizer(self.model, self.cfg)
        self.scheduler = build_scheduler(self.cfg, self.optimizer)
        self.global_step = 0

    def __call__(self, batch):
        from .functional import set_hip_backward
        from ..model import backbone as _bb
        set_hip_backward(self.hip_backward)
        _bb.TRAIN_HIP_CONV = self.hip_conv
        try:
            loss = self.core(batch)
        finally:
            set_hip_backward(False)
            _bb.TRAIN_HIP_CONV = False
        hip_weights = None
        if self.hip_backward or self.hip_conv:
            from .hip_autograd import WEIGHTS as hip_weights
        if 'loss_scalars' not in batch:          # the wrapper handed the forward a copy of the batch
            raise RuntimeError('the training forward did not write into the caller\'s batch')
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        if self.cfg['gradient_clipping']:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.cfg['gradient_clipping'])
        lr = warmup_lr(self.cfg, self.global_step)
        if lr is not None:
            for pg in self.optimizer.param_groups:
                pg['lr'] = lr
        self.optimizer.step()
        if hip_weights is not None:
            hip_weights.clear()                  # the 16-bit copies belong to the weights of the step that made them
        if self.cfg['scheduler_interval'] == 'step':
            self.scheduler.step()
        self.global_step += 1
        self.model._invalidate()                 # the inference path caches packed weights
        return loss.detach()

    def epoch_end(self):
        if self.cfg['scheduler_interval'] == 'epoch':
            self.scheduler.step()


# ---------------------------------------------------------------------------------------------
# synthetic homography pairs (the Oxford-Paris images are not available offline): a smooth random texture and
# its warp under a random corner-perturbation homography, with the GT matrices the supervision needs
# ---------------------------------------------------------------------------------------------
def synthetic_homography_batch(n, hw, seed, device='cpu', max_shift=0.12):
    H, W = hw
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(n, 1, H // 8 + 2, W // 8 + 2, generator=g)
    img0 = torch.nn.functional.interpolate(base, size=(H, W), mode='bicubic', align_corners=True)
    img0 = (img0 + 0.15 * torch.rand(n, 1, H, W, generator=g)).clamp(0, 1)
    src = torch.tensor([[0., 0.], [W - 1., 0.], [W - 1., H - 1.], [0., H - 1.]]).expand(n, 4, 2)
    dst = src + (torch.rand(n, 4, 2, generator=g) * 2 - 1) * torch.tensor([W, H]) * max_shift
    A = torch.zeros(n, 8, 8, dtype=torch.float64)
    bvec = torch.zeros(n, 8, 1, dtype=torch.float64)
    for k in range(4):
        x, y = src[:, k, 0].double(), src[:, k, 1].double()
        u, v = dst[:, k, 0].double(), dst[:, k, 1].double()
        A[:, 2 * k] = torch.stack([x, y, torch.ones_like(x), torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(x), -u * x, -u * y], -1)
        A[:, 2 * k + 1] = torch.stack([torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(x), x, y, torch.ones_like(x), -v * x, -v * y], -1)
        bvec[:, 2 * k, 0], bvec[:, 2 * k + 1, 0] = u, v
    h8 = torch.linalg.solve(A, bvec)[:, :, 0]
    H01 = torch.cat([h8, torch.ones(n, 1, dtype=torch.float64)], 1).view(n, 3, 3)
    H10 = torch.linalg.inv(H01)
    # image1(p) = image0(H10 p): sample image0 at the back-warped pixel grid
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing='ij')
    p = torch.stack([xs, ys, torch.ones_like(xs)], -1).view(1, -1, 3) @ H10.transpose(1, 2)
    p = p[..., :2] / p[..., 2:]
    grid = torch.stack([p[..., 0] / (W - 1) * 2 - 1, p[..., 1] / (H - 1) * 2 - 1], -1).view(n, H, W, 2).float()
    img1 = torch.nn.functional.grid_sample(img0, grid, mode='bilinear', padding_mode='zeros', align_corners=True)
    return {'image0': img0.to(device), 'image1': img1.to(device), 'H_0to1': H01.float().to(device),
            'H_1to0': H10.float().to(device), 'dataset_name': ['oxford'] * n, 'pair_names': [f'synthetic{seed}'] * n}


def synthetic_megadepth_batch(n, hw, seed, device='cpu'):
    """A MegaDepth-style batch (BASELINE configs[3]; the keys lightning_depth_geoformer.py:87-99 feeds: image*, depth*, T_*, K*, scale*,
    mask*, dataset_name) without the dataset: a smooth random texture, image 1 = image 0 shifted by 8 px (one coarse cell); two cameras
    0.27 units apart over a slanted plane at depth ~6, focal length 180; zero-padded to (H, W) the way the MegaDepth loader pads to its
    square size - the bottom eighth of image 0 in the odd samples and the right eighth of image 1 in the even ones are padding
    (`mask0` / `mask1` at 1/8 scale) -, per-image scales (depth maps live at the original resolution, up to 1.5x the padded one).
    The SUPERVISION follows the declared geometry (cell * scale -> K, depth, T -> / scale, as spvs_coarse does), not the picture: every
    sample has one non-unit scale, so its labels say ((x - 8) / 1.5, (y - 8) / 1.25) where the image content says (x - 8, y - 8).  The
    batch exercises every branch of the supervision (scales, padding masks, depth consistency) at the shapes of configs[3] for the
    timing leg of bench.py and the loss-falls checks; it is not a realism check and nothing is concluded from the fitted matches."""
    H, W = hw
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(n, 1, H // 8 + 3, W // 8 + 3, generator=g)
    img = torch.nn.functional.interpolate(base, size=(H + 8, W + 8), mode='bicubic', align_corners=True)
    img = (img + 0.15 * torch.rand(n, 1, H + 8, W + 8, generator=g)).clamp(0, 1)
    image0, image1 = img[:, :, :H, :W].clone(), img[:, :, 8:, 8:].clone()
    mask0 = torch.ones(n, H // 8, W // 8, dtype=torch.bool)
    mask1 = torch.ones(n, H // 8, W // 8, dtype=torch.bool)
    r0, c1 = (H // 8) * 7 // 8, (W // 8) * 7 // 8
    mask0[1::2, r0:] = False
    mask1[0::2, :, c1:] = False
    image0[1::2, :, 8 * r0:] = 0
    image1[0::2, :, :, 8 * c1:] = 0
    scale0 = torch.ones(n, 2); scale1 = torch.ones(n, 2)
    scale0[1::2] = torch.tensor([1.25, 1.5]); scale1[0::2] = torch.tensor([1.5, 1.25])
    Hd, Wd = 2 * H, 2 * W
    ys, xs = torch.meshgrid(torch.arange(Hd, dtype=torch.float32), torch.arange(Wd, dtype=torch.float32), indexing='ij')
    depth0 = (6.0 + 0.004 * xs + 0.006 * ys)[None].repeat(n, 1, 1)
    depth1 = (6.1 + 0.004 * xs + 0.006 * ys)[None].repeat(n, 1, 1)
    K = torch.tensor([[[180., 0., W / 2.], [0., 180., H / 2.], [0., 0., 1.]]]).repeat(n, 1, 1)
    T = torch.eye(4)[None].repeat(n, 1, 1)
    T[:, :3, 3] = torch.tensor([-0.27, -0.27, 0.0])
    out = {'image0': image0, 'image1': image1, 'depth0': depth0, 'depth1': depth1, 'T_0to1': T, 'T_1to0': torch.inverse(T),
           'K0': K, 'K1': K.clone(), 'scale0': scale0, 'scale1': scale1, 'mask0': mask0, 'mask1': mask1}
    out = {k: v.to(device) for k, v in out.items()}
    out.update(dataset_name=['megadepth'] * n, pair_names=[f'synthetic{seed}'] * n)
    return out

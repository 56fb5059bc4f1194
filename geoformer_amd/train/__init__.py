"""Training step of GeoFormer: supervision, loss, schedule, DDP - see trainer.py."""
from .functional import forward_train
from .loss import GeoLoss
from .supervision import spvs_coarse, spvs_fine2
from .trainer import TrainStep, build_optimizer, build_scheduler, scale_trainer_cfg, synthetic_homography_batch, synthetic_megadepth_batch, warmup_lr

__all__ = ['forward_train', 'GeoLoss', 'spvs_coarse', 'spvs_fine2', 'TrainStep', 'build_optimizer', 'build_scheduler',
           'scale_trainer_cfg', 'synthetic_homography_batch', 'synthetic_megadepth_batch', 'warmup_lr']

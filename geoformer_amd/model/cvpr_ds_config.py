"""LoFTR-side defaults consumed by GeoFormer: same keys and values as the reference's
model/loftr_src/loftr/utils/cvpr_ds_config.py:10-48 (lower-cased dict `default_cfg`)."""
import copy

_DEFAULT = {
    'backbone_type': 'ResNetFPN',
    'resolution': (8, 2),
    'fine_window_size': 5,
    'fine_concat_coarse_feat': True,
    'resnetfpn': {'initial_dim': 128, 'block_dims': [128, 196, 256]},
    'coarse': {'d_model': 256, 'd_ffn': 256, 'nhead': 8, 'layer_names': ['self', 'cross'] * 4,
               'attention': 'linear', 'temp_bug_fix': False},
    'match_coarse': {'thr': 0.4, 'border_rm': 2, 'match_type': 'dual_softmax', 'dsmax_temperature': 0.1,
                     'skh_iters': 3, 'skh_init_bin_score': 1.0, 'skh_prefilter': True,
                     'train_coarse_percent': 0.4, 'train_pad_num_gt_min': 200},
    'fine': {'d_model': 128, 'd_ffn': 128, 'nhead': 8, 'layer_names': ['self', 'cross'] * 1, 'attention': 'linear'},
}

default_cfg = copy.deepcopy(_DEFAULT)


def get_default_cfg():
    return copy.deepcopy(_DEFAULT)

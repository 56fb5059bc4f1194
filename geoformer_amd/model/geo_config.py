"""GeoFormer hyper-parameters: same keys and values as the reference's model/geo_config.py:10-17
(lower-cased dict `default_cfg`), without the yacs dependency.  One optional key is added:
'precision' ('fp32' = parity mode, 'fp16' = fp16 storage / fp32 accumulate); absent -> 'fp32'."""
import copy

_DEFAULT = {
    'layer_names': ['self', 'cross'] * 2,
    'nhead': 4,
    'coarse_thr': 0.2,
    'fine_temperature': 0.1,
    'fine_thr': 0.1,
    'window_size': 5,
    'topk': 1,
}

default_cfg = copy.deepcopy(_DEFAULT)


def lower_config(cfg):
    if not isinstance(cfg, dict):
        return cfg
    return {k.lower(): lower_config(v) for k, v in cfg.items()}


def get_cfg_model():
    """A fresh copy of the defaults (the reference returns a clone of its yacs node)."""
    return copy.deepcopy(_DEFAULT)

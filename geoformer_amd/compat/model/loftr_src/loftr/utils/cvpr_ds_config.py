"""`from model.loftr_src.loftr.utils.cvpr_ds_config import default_cfg` (reference: cvpr_ds_config.py:50)."""
from geoformer_amd.model.cvpr_ds_config import *  # noqa: F401,F403
from geoformer_amd.model.cvpr_ds_config import default_cfg  # noqa: F401

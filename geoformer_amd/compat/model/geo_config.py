"""`from model.geo_config import default_cfg` (reference: model/geo_config.py:19)."""
from geoformer_amd.model.geo_config import *  # noqa: F401,F403
from geoformer_amd.model.geo_config import default_cfg  # noqa: F401

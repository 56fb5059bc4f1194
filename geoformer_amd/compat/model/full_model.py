"""`from model.full_model import GeoFormer` (reference: model/full_model.py:18) -> geoformer_amd's GeoFormer."""
from geoformer_amd.model.full_model import GeoFormer  # noqa: F401

from .full_model import GeoFormer  # noqa: F401

"""geoformer_amd - MI355X-native GeoFormer coarse-to-fine matching path.

Host side mirrors the reference's module interface (GeoFormer(loftr_config, geoformer_cfg)
.forward(data)); the path itself runs in hand-written HIP kernels behind a C ABI
(include/geoformer_hip.h, libgeoformer_hip.so).
"""
__version__ = '0.1.0'

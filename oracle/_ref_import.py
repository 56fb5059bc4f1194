"""Import harness for the *original* GeoFormer code (build container only).

TEST INFRASTRUCTURE - never imported by the product package.

/root/reference exists only in the build container.  This module makes
`model.full_model` importable there by registering inert stand-ins for the
third-party modules the reference imports but that are absent from the image
(cv2, kornia, yacs, skimage, imgaug, torchvision).  Of those, only
`cv2.findHomography` is ever *called* on the forward path
(model/geo_module.py:47-48); the fixture generator injects a deterministic
replacement and records its (M, mask) output inside every fixture so that
nothing downstream depends on OpenCV.

Used by oracle/gen_golden.py only.  Nothing from /root/reference is copied:
fixtures hold inputs and outputs (data), never source.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("GEOFORMER_REFERENCE", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "model"))


class _Cfg(dict):
    """Just enough of yacs.config.CfgNode: attribute access + clone()."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        import copy
        return copy.deepcopy(self)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_homography_hook = [None]


def set_find_homography(fn):
    """fn(kp0[n,2] ndarray, kp1[n,2] ndarray) -> (M float64[3,3] | None, mask uint8[n,1])."""
    _homography_hook[0] = fn


def _find_homography(src, dst, method=0, thr=3.0, *a, **k):
    if _homography_hook[0] is None:
        raise RuntimeError("no findHomography injected")
    return _homography_hook[0](src, dst)


def install_stubs():
    import torch

    def create_meshgrid(h, w, normalized_coordinates=True, device=None, dtype=torch.float32):
        ys, xs = torch.meshgrid(torch.arange(h, dtype=dtype), torch.arange(w, dtype=dtype), indexing="ij")
        return torch.stack([xs, ys], -1)[None]

    _mod("yacs")
    _mod("yacs.config", CfgNode=_Cfg)
    _mod("kornia")
    _mod("kornia.geometry")
    _mod("kornia.geometry.subpix", dsnt=types.ModuleType("dsnt"))
    _mod("kornia.utils", create_meshgrid=create_meshgrid)
    _mod("kornia.utils.grid", create_meshgrid=create_meshgrid)
    _mod("skimage")
    _mod("skimage.feature", peak_local_max=None)
    _mod("imgaug")
    _mod("imgaug.augmenters")
    _mod("torchvision")
    _mod("torchvision.transforms")
    _mod("cv2", RANSAC=8, findHomography=_find_homography)
    _mod("pydegensac")
    sys.modules["kornia"].geometry = sys.modules["kornia.geometry"]
    sys.modules["kornia"].utils = sys.modules["kornia.utils"]
    sys.modules["kornia.geometry"].subpix = sys.modules["kornia.geometry.subpix"]
    sys.modules["imgaug"].augmenters = sys.modules["imgaug.augmenters"]


def import_reference():
    """Returns the reference's modules as a namespace (build container only)."""
    if not reference_available():
        raise RuntimeError(f"{REFERENCE_ROOT} not present: fixtures can only be regenerated in the build container")
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    ns = types.SimpleNamespace()
    ns.full_model = importlib.import_module("model.full_model")
    ns.geo_config = importlib.import_module("model.geo_config")
    ns.cvpr_ds_config = importlib.import_module("model.loftr_src.loftr.utils.cvpr_ds_config")
    ns.position_encoding = importlib.import_module("model.loftr_src.loftr.utils.position_encoding")
    ns.coarse_matching = importlib.import_module("model.loftr_src.loftr.utils.coarse_matching")
    ns.loftr_transformer = importlib.import_module("model.loftr_src.loftr.loftr_module.transformer")
    ns.linear_attention = importlib.import_module("model.loftr_src.loftr.loftr_module.linear_attention")
    ns.fine_preprocess = importlib.import_module("model.loftr_src.loftr.loftr_module.fine_preprocess")
    ns.geo_module = importlib.import_module("model.geo_module")
    ns.geo_transformer = importlib.import_module("model.geo_transformer.transformer")
    ns.geo_attention = importlib.import_module("model.geo_transformer.geo_attention")
    ns.fine_matching2 = importlib.import_module("model.fine_matching2")
    ns.common_utils = importlib.import_module("utils.common_utils")
    ns.homography = importlib.import_module("utils.homography")
    return ns


def import_eval_helpers():
    """The pure-numpy/python helpers of the evaluation harness (build container only)."""
    import importlib
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    ns = types.SimpleNamespace()
    ns.hpatches_helper = importlib.import_module("eval_tool.immatch.utils.hpatches_helper")
    ns.data_io = importlib.import_module("eval_tool.immatch.utils.data_io")
    return ns


def import_training():
    """Supervision + loss of the training harness (build container only); loguru is absent and only logs."""
    import importlib
    install_stubs()
    _mod("loguru", logger=types.SimpleNamespace(warning=lambda *a, **k: None, info=lambda *a, **k: None))
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    ns = import_reference()
    ns.supervision = importlib.import_module("model.loftr_src.loftr.utils.supervision")
    ns.loftr_loss = importlib.import_module("model.loftr_src.losses.loftr_loss")
    return ns

"""ctypes front end of oracle/ransac_oracle.c (TEST INFRASTRUCTURE; see the C file's header:
homography parity with the reference's OpenCV call is UNPINNED, this states the build's own
RANSAC so the HIP kernel can be checked bit-for-bit on the inlier mask)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libransac_oracle.so')
_lib = None

DEFAULT_ITERS = 1024
DEFAULT_SEED = 0x5EED
DEFAULT_LM_ITERS = 10          # OpenCV's findHomography appends 10 Levenberg-Marquardt iterations to its RANSAC


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(['make', '-C', _HERE, '-s'], check=True)
        _lib = ctypes.CDLL(_SO)
        _lib.gf_oracle_ransac.restype = ctypes.c_int
        _lib.gf_oracle_ransac.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                          ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _lib.gf_oracle_ransac_f32.restype = ctypes.c_int
        _lib.gf_oracle_ransac_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                              ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                              ctypes.c_void_p]
    return _lib


def find_homography(kp0, kp1, thr=8.0, iters=DEFAULT_ITERS, seed=DEFAULT_SEED, sample=0, lm_iters=DEFAULT_LM_ITERS):
    """(kp0 [n,2] int, kp1 [n,2] int) -> (M float64[3,3] | None, mask uint8[n,1]) - the contract of
    cv2.findHomography(kp0, kp1, cv2.RANSAC, thr) as used at model/geo_module.py:47-48."""
    a = np.ascontiguousarray(kp0, dtype=np.int64)
    b = np.ascontiguousarray(kp1, dtype=np.int64)
    n = len(a)
    M = np.zeros(9, np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    ok = _load().gf_oracle_ransac(a.ctypes.data, b.ctypes.data, n, float(thr), int(iters), int(seed), int(sample), int(lm_iters),
                                  M.ctypes.data, mask.ctypes.data)
    return (M.reshape(3, 3) if ok else None), mask[:n, None]


def find_homography_subpixel(kp0, kp1, thr=3.0, iters=DEFAULT_ITERS, seed=DEFAULT_SEED, sample=0, lm_iters=DEFAULT_LM_ITERS,
                             min_points=4):
    """Sub-pixel keypoints (float32 [n,2]) and a caller-chosen gate: the evaluation harness' homography from the final
    matches (hpatches_helper.py:185-239: cv2.findHomography(p1, p2, cv2.RANSAC, 3)), the CPU statement of
    geoformer_amd.matcher.estimate_homography (device RANSAC with integer_keypoints=False, min_points=4)."""
    a = np.ascontiguousarray(kp0, dtype=np.float32)
    b = np.ascontiguousarray(kp1, dtype=np.float32)
    n = len(a)
    M = np.zeros(9, np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    ok = _load().gf_oracle_ransac_f32(a.ctypes.data, b.ctypes.data, n, float(thr), int(iters), int(seed), int(sample),
                                      int(lm_iters), int(min_points), M.ctypes.data, mask.ctypes.data)
    return (M.reshape(3, 3) if ok else None), mask[:n, None]


def make_homography_fn(thr=8.0, iters=DEFAULT_ITERS, seed=DEFAULT_SEED, lm_iters=DEFAULT_LM_ITERS):
    """homography_fn for geoformer_oracle.geoformer_forward: the sample index advances per call the
    way GeoModule.apply_RANSAC walks the batch."""
    state = {'sample': 0}

    def fn(kp0, kp1):
        s = state['sample']
        state['sample'] += 1
        return find_homography(kp0, kp1, thr, iters, seed, s, lm_iters)
    fn.reset = lambda: state.update(sample=0)
    return fn

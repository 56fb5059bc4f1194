"""Fixture g17: (M, mask) of the homography RANSAC as it stood BEFORE the Levenberg-Marquardt refinement was added
(`lm_iters = 0` must keep reproducing that M).  The generator builds the C oracle of commit a02ec81 (rounds 1-2:
`git show a02ec81:oracle/ransac_oracle.c`, no `lm_iters` argument) in a scratch directory and runs it on the point sets of
tests/test_ops_gpu.py::test_ransac_matches_c_oracle_bit_exact_mask; tests/test_ransac_oracle.py then requires today's oracle
with lm_iters = 0, and tests/test_ops_gpu.py the device kernel through the version-1 entry point gf_ransac_homography, to
return the same mask bit for bit and M to 1e-12.

    python oracle/gen_ransac_golden.py          (needs the git history of this repository; gcc)
"""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, 'tests', 'golden', 'g17_ransac_lm0.npz')
COMMIT = 'a02ec81'


def point_sets():
    """The planted sets of tests/test_ops_gpu.py (same construction, same seeds)."""
    def planted(n, n_out, H, seed):
        rng = np.random.default_rng(seed)
        p0 = np.stack([rng.integers(0, 80, n) * 8, rng.integers(0, 80, n) * 8], 1).astype(np.int64)
        q = np.c_[p0, np.ones(n)] @ H.T
        p1 = np.floor(q[:, :2] / q[:, 2:3] / 8).astype(np.int64) * 8
        out = rng.choice(n, n_out, replace=False)
        p1[out] = np.stack([rng.integers(0, 80, n_out) * 8, rng.integers(0, 80, n_out) * 8], 1)
        return p0, p1
    Hs = [np.array([[1., 0, 8], [0, 1, 8], [0, 0, 1]]), np.array([[0.93, -0.21, 44.3], [0.18, 1.07, -9.6], [0, 0, 1]]),
          np.array([[1.12, 0.08, -21.0], [-0.05, 0.9, 37.5], [2e-4, -1.3e-4, 1]])]
    return [planted(900, 300, Hs[0], 1), planted(8, 0, Hs[0], 2), planted(2500, 1500, Hs[1], 3),
            (np.zeros((30, 2), np.int64), np.zeros((30, 2), np.int64)), planted(640, 100, Hs[2], 4), planted(9, 0, Hs[0], 5)]


def main():
    with tempfile.TemporaryDirectory() as tmp:
        src = subprocess.run(['git', '-C', ROOT, 'show', f'{COMMIT}:oracle/ransac_oracle.c'], capture_output=True, check=True).stdout
        c = os.path.join(tmp, 'ransac_v1.c')
        open(c, 'wb').write(src)
        so = os.path.join(tmp, 'libransac_v1.so')
        subprocess.run(['gcc', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-o', so, c, '-lm'], check=True)
        lib = ctypes.CDLL(so)
        lib.gf_oracle_ransac.restype = ctypes.c_int
        lib.gf_oracle_ransac.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_uint32,
                                         ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        out = {}
        for b, (p0, p1) in enumerate(point_sets()):
            a, bb = np.ascontiguousarray(p0, np.int64), np.ascontiguousarray(p1, np.int64)
            M = np.zeros(9, np.float64)
            mask = np.zeros(max(len(a), 1), np.uint8)
            ok = lib.gf_oracle_ransac(a.ctypes.data, bb.ctypes.data, len(a), 8.0, 1024, 0x5EED, b, M.ctypes.data, mask.ctypes.data)
            out[f's{b}_valid'] = np.array(ok, np.int32)
            out[f's{b}_M'] = M.reshape(3, 3)
            out[f's{b}_mask'] = np.packbits(mask[:len(a)])
            out[f's{b}_n'] = np.array(len(a), np.int32)
        np.savez_compressed(OUT, **out)
        print('wrote', OUT, {k: (v.tolist() if v.size < 10 else v.shape) for k, v in out.items() if k.endswith('valid')})


if __name__ == '__main__':
    sys.exit(main())

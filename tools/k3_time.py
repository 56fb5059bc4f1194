"""Times the K3 shapes of one coarse encoder layer (batch 8 at 640x480: 102400 tokens) on the GPU.

    python tools/k3_time.py [M]
"""
import sys

import torch

sys.path.insert(0, '.')
from geoformer_amd import ops  # noqa: E402


def timeit(fn, n=20):
    """Average device-side duration (HIP events around each launch; host overhead excluded)."""
    import ctypes
    from geoformer_amd import _lib
    L = _lib.lib()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    L.gf_profile_enable(1)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    ms, cnt, work = ctypes.c_double(0), ctypes.c_int(0), ctypes.c_double(0)
    L.gf_profile_collect(b'k3_linear', ctypes.byref(ms), ctypes.byref(cnt), ctypes.byref(work))
    L.gf_profile_enable(0)
    return ms.value / max(cnt.value, 1)


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 102400
    dev = 'cuda'
    torch.manual_seed(0)
    for dt in (torch.float16, torch.float32):
        x = torch.randn(M, 256, device=dev, dtype=dt)
        y = torch.randn(M, 256, device=dev, dtype=dt)
        h = torch.randn(M, 512, device=dev, dtype=dt)
        w256 = torch.randn(256, 256, device=dev, dtype=dt) * 0.05
        w512 = torch.randn(512, 256, device=dev, dtype=dt) * 0.05
        wm1 = torch.randn(512, 512, device=dev, dtype=dt) * 0.05
        wm2 = torch.randn(256, 512, device=dev, dtype=dt) * 0.05
        g = torch.ones(256, device=dev)
        b = torch.zeros(256, device=dev)
        cases = {
            'q    256->256 none  ': (lambda: ops.linear(x, w256), 256 * 256),
            'kv   256->512 none  ': (lambda: ops.linear(x, w512), 256 * 512),
            'mrg  256->256 ln    ': (lambda: ops.linear(x, w256, epilogue=ops.EPI_LN, ln=(g, b)), 256 * 256),
            'mlp1 512->512 relu  ': (lambda: ops.linear(x, wm1, a2=y, epilogue=ops.EPI_RELU), 512 * 512),
            'mlp2 512->256 ln_res': (lambda: ops.linear(h, wm2, epilogue=ops.EPI_LN_RES, ln=(g, b), residual=x), 512 * 256),
        }
        for name, (fn, kn) in cases.items():
            ms = timeit(fn)
            print(f'{str(dt)[6:]:8s} {name} {ms * 1e3:8.1f} us  {2.0 * M * kn / ms * 1e-9:7.1f} TFLOP/s', flush=True)


if __name__ == '__main__':
    main()

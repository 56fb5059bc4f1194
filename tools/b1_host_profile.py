"""Host time of one HPatches-shaped pair (BASELINE configs[1]: 480x640 against 480x608, batch 1, bf16, nominal load) - cProfile of the
matcher's forward on one stream, top functions by own time.   python tools/b1_host_profile.py"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device('cuda:0')
mn, _ = bench.build_model('bf16', 0.2, 0.1, dev)
g0, g1 = (60, 80), (60, 76)
g = torch.Generator().manual_seed(70000)
big = torch.randn(1, 256, g0[0] + 1, g0[1] + 1, generator=g) * 0.5
bigf = torch.randn(1, 128, 4 * (g0[0] + 1), 4 * (g0[1] + 1), generator=g)
c0, f0 = big[:, :, :g0[0], :g0[1]], bigf[:, :, :4 * g0[0], :4 * g0[1]]
c1 = big[:, :, 1:1 + g1[0], 1:1 + g1[1]] + 0.35 * torch.randn(1, 256, *g1, generator=g)
f1 = bigf[:, :, 4:4 + 4 * g1[0], 4:4 + 4 * g1[1]] + 0.35 * torch.randn(1, 128, 4 * g1[0], 4 * g1[1], generator=g)
c0, f0, c1, f1 = [t.to(device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last) for t in (c0, f0, c1, f1)]
i0, i1 = torch.rand(1, 1, 480, 640, device=dev), torch.rand(1, 1, 480, 608, device=dev)


def step():
    (fc0, ff0), (fc1, ff1) = mn._backbone(i0), mn._backbone(i1)
    return mn.forward_features({'image0': i0, 'image1': i1}, torch.add(c0, fc0, alpha=0.0), torch.add(f0, ff0, alpha=0.0),
                               torch.add(c1, fc1, alpha=0.0), torch.add(f1, ff1, alpha=0.0))


with torch.no_grad():
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50):
        step()
    th = time.perf_counter() - t
    torch.cuda.synchronize()
    print(f'host enqueue time {th / 50 * 1e3:.3f} ms per pair, with the device {(time.perf_counter() - t) / 50 * 1e3:.3f} ms')
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        step()
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumtime').print_stats(40)

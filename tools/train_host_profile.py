"""Host time of the MegaDepth-style training step (batch 8, bf16, HIP Functions): cProfile of three steps, top functions by own and by
cumulative time.   python tools/train_host_profile.py"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from geoformer_amd import miopen; miopen.use_shipped_find_db()
from geoformer_amd.model.cvpr_ds_config import get_default_cfg
from geoformer_amd.model.full_model import GeoFormer
from geoformer_amd.model.geo_config import get_cfg_model
from geoformer_amd.weights import deterministic_init_
from geoformer_amd.train import TrainStep, synthetic_megadepth_batch
g = get_cfg_model(); g.update(coarse_thr=0.0, fine_thr=0.0, precision='fp32')
model = deterministic_init_(GeoFormer(get_default_cfg(), g)).cuda()
step = TrainStep(model, batch_size=8, fused_coarse_loss=True, precision='bf16', hip_backward=True, hip_conv=True)
base = [synthetic_megadepth_batch(8, (640, 640), seed=900 + i, device='cuda') for i in range(2)]
for i in range(5):
    step(dict(base[i % 2]))
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(3):
    step(dict(base[i % 2]))
torch.cuda.synchronize()
print(f'step {(time.perf_counter() - t) / 3 * 1e3:.1f} ms')
pr = cProfile.Profile(); pr.enable()
for i in range(3):
    step(dict(base[i % 2]))
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(40)
st.sort_stats('cumtime').print_stats(60)

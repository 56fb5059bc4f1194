"""Where does the FIRST training step spend its time?  (cProfile over one TrainStep call on a fresh process; VERDICT r05 #7d)"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geoformer_amd import miopen as gf_miopen
gf_miopen.use_shipped_find_db()
import torch
from geoformer_amd.model.cvpr_ds_config import get_default_cfg
from geoformer_amd.model.full_model import GeoFormer
from geoformer_amd.model.geo_config import get_cfg_model
from geoformer_amd.weights import deterministic_init_
from geoformer_amd.train import TrainStep, synthetic_megadepth_batch
dev = torch.device('cuda', 0)
g = get_cfg_model(); g.update(coarse_thr=0.0, fine_thr=0.0, precision='fp32')
model = deterministic_init_(GeoFormer(get_default_cfg(), g)).to(dev)
step = TrainStep(model, batch_size=8, fused_coarse_loss=True, precision='bf16', hip_backward=True, hip_conv=True)
data = [dict(synthetic_megadepth_batch(8, (640, 640), seed=900 + i, device=dev)) for i in range(3)]
torch.cuda.synchronize()
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable()
loss = float(step(data[0]))
torch.cuda.synchronize()
pr.disable()
print(f'step 0: {time.perf_counter() - t:.1f} s, loss {loss:.3f}')
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
t = time.perf_counter(); float(step(data[1])); torch.cuda.synchronize(); print(f'step 1: {time.perf_counter() - t:.2f} s')

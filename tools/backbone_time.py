import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geoformer_amd.model.full_model import GeoFormer
from geoformer_amd.model.cvpr_ds_config import get_default_cfg
from geoformer_amd.model.geo_config import get_cfg_model
from geoformer_amd.weights import deterministic_init_
torch.backends.cudnn.benchmark = True
m = deterministic_init_(GeoFormer(get_default_cfg(), get_cfg_model()).eval()).cuda()
for dt in (torch.float16, torch.bfloat16):
    for cl in (True, False):
        m.set_precision('fp16', backbone_dtype=dt)
        bb = m._inference_backbone()
        if not cl:
            bb = bb.to(memory_format=torch.contiguous_format)
        for B in (8, 16, 32):
            x = torch.rand(B, 1, 640, 640, device='cuda', dtype=dt)
            if cl: x = x.contiguous(memory_format=torch.channels_last)
            with torch.no_grad():
                for _ in range(3): bb(x)
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(5): bb(x)
                torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / 5 * 1e3
            print(dt, 'channels_last' if cl else 'nchw', 'images', B, 'ms %.2f' % ms, 'ms/pair %.3f' % (ms / B * 2), flush=True)

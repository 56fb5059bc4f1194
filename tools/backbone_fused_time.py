"""Times the fp16 inference backbone at the bench's batch (16 images of 640x640):
   python tools/backbone_fused_time.py new|old     (old = eager module with BN folded, no glue kernels)"""
import os, sys, time
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, here)
from geoformer_amd import miopen as gf_miopen
gf_miopen.use_shipped_find_db()
import torch
import torch.nn as nn
torch.backends.cudnn.benchmark = '--tune' in sys.argv
PAD = int(sys.argv[sys.argv.index('--pad') + 1]) if '--pad' in sys.argv else 32
from geoformer_amd.model.backbone import FusedInferenceBackbone, build_backbone
from geoformer_amd.model.cvpr_ds_config import get_default_cfg
bb = build_backbone(get_default_cfg()).cuda().eval()
if len(sys.argv) > 1 and sys.argv[1] == 'old':
    from torch.nn.utils.fusion import fuse_conv_bn_eval
    def fold(parent, c, b):
        setattr(parent, c, fuse_conv_bn_eval(getattr(parent, c), getattr(parent, b))); setattr(parent, b, nn.Identity())
    fold(bb, 'conv1', 'bn1')
    for layer in (bb.layer1, bb.layer2, bb.layer3):
        for blk in layer:
            fold(blk, 'conv1', 'bn1'); fold(blk, 'conv2', 'bn2')
            if blk.downsample is not None:
                blk.downsample = nn.Sequential(fuse_conv_bn_eval(blk.downsample[0], blk.downsample[1]), nn.Identity())
    for seq in (bb.layer2_outconv2, bb.layer1_outconv2):
        seq[0] = fuse_conv_bn_eval(seq[0], seq[1]); seq[1] = nn.Identity()
    m = bb.half().to(memory_format=torch.channels_last)
    fb = lambda x: m(x.half().contiguous(memory_format=torch.channels_last))
else:
    fb = FusedInferenceBackbone(bb, torch.float16, pad_multiple=PAD)
x = torch.rand(16, 1, 640, 640, device='cuda')
with torch.no_grad():
    for _ in range(2):
        fb(x)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        fb(x)
    torch.cuda.synchronize()
if '--tune' in sys.argv:
    gf_miopen.save_find_db()
print('pad', PAD, 'tune', torch.backends.cudnn.benchmark, 'backbone ms/call %.2f' % ((time.perf_counter() - t) / 5 * 1e3))

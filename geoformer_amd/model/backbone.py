"""ResNet-FPN 1/8 + 1/2 feature extractor.  Stays on PyTorch-ROCm / MIOpen by design (north_star;
SURVEY §2 row 10): only the module tree and parameter names are re-declared so that
`geoformer.ckpt` loads with the same keys as the reference's ResNetFPN_8_2
(model/loftr_src/loftr/backbone/resnet_fpn.py:43-118).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import fused, ops


class _Conv2d(nn.Conv2d):
    """nn.Conv2d (same parameters, same state-dict keys) whose TRAINING forward takes the HIP Function for the 3x3 / stride-1 shapes
    K10 is built for when `TRAIN_HIP_CONV` is on (TrainStep(hip_conv=True): 16-bit autocast, channels_last maps) - forward and backward-data
    on K10, backward-weights on the library (train/hip_autograd.py:HipConv3x3)."""

    def forward(self, x):
        if TRAIN_HIP_CONV and self.training and self.kernel_size == (3, 3) and self.stride == (1, 1) and x.is_cuda and torch.is_autocast_enabled():
            from ..train import hip_autograd as HA
            x16 = x.to(torch.get_autocast_dtype('cuda'))
            if HA.conv3x3_supported(x16, self.weight):
                x16 = x16.contiguous(memory_format=torch.channels_last)
                with torch.autocast('cuda', enabled=False):
                    return HA.conv3x3(x16, self.weight)
        if (TRAIN_HIP_CONV and self.training and self.kernel_size == (1, 1) and self.stride == (1, 1) and x.is_cuda and torch.is_autocast_enabled()
                and self.in_channels % 4 == 0 and self.out_channels % 4 == 0):
            from ..train import hip_autograd as HA                 # the FPN's 1x1 convolutions on the K3 engine (round 6)
            x16 = x.to(torch.get_autocast_dtype('cuda')).contiguous(memory_format=torch.channels_last)
            with torch.autocast('cuda', enabled=False):
                return HA.conv1x1(x16, self.weight)
        return super().forward(x)


TRAIN_HIP_CONV = False


def _upsample(x, size, training):
    """The FPN's bilinear upsampling (resnet_fpn.py:104-105, :110-111); in the HIP training step (TRAIN_HIP_CONV) with the own backward."""
    if (TRAIN_HIP_CONV and training and x.is_cuda and x.requires_grad and x.shape[1] % 4 == 0
            and x.is_contiguous(memory_format=torch.channels_last) and x.dtype in (torch.float32, torch.float16, torch.bfloat16)):
        from ..train import hip_autograd as HA
        return HA.upsample_bilinear(x, tuple(size))
    return F.interpolate(x, size=size, mode='bilinear', align_corners=True)


def _conv(cin, cout, k, stride=1):
    return _Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)


class BasicBlock(nn.Module):
    """Two 3x3 conv + BN; strided blocks project the identity with 1x1 conv + BN ('downsample')."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1, self.conv2 = _conv(cin, cout, 3, stride), _conv(cout, cout, 3)
        self.bn1, self.bn2 = nn.BatchNorm2d(cout), nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None if stride == 1 else nn.Sequential(_conv(cin, cout, 1, stride), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        if self.downsample is not None:
            x = self.downsample(x)
        return self.relu(x + y)


class ResNetFPN_8_2(nn.Module):
    def __init__(self, config):
        super().__init__()
        d0 = config['initial_dim']
        b1, b2, b3 = config['block_dims']
        self.conv1 = nn.Conv2d(1, d0, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(d0)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = nn.Sequential(BasicBlock(d0, b1, 1), BasicBlock(b1, b1, 1))   # 1/2
        self.layer2 = nn.Sequential(BasicBlock(b1, b2, 2), BasicBlock(b2, b2, 1))   # 1/4
        self.layer3 = nn.Sequential(BasicBlock(b2, b3, 2), BasicBlock(b3, b3, 1))   # 1/8
        self.layer3_outconv = _conv(b3, b3, 1)
        self.layer2_outconv = _conv(b2, b3, 1)
        self.layer2_outconv2 = nn.Sequential(_conv(b3, b3, 3), nn.BatchNorm2d(b3), nn.LeakyReLU(), _conv(b3, b2, 3))
        self.layer1_outconv = _conv(b1, b2, 1)
        self.layer1_outconv2 = nn.Sequential(_conv(b2, b2, 3), nn.BatchNorm2d(b2), nn.LeakyReLU(), _conv(b2, b1, 3))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x1 = self.layer1(self.relu(self.bn1(self.conv1(x))))
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        c3 = self.layer3_outconv(x3)
        c2 = self.layer2_outconv(x2)
        c2 = self.layer2_outconv2(c2 + _upsample(c3, c2.shape[2:], self.training))
        c1 = self.layer1_outconv(x1)
        c1 = self.layer1_outconv2(c1 + _upsample(c2, c1.shape[2:], self.training))
        return [c3, c1]


def _padded(c, multiple):
    return c if (multiple <= 1 or c == 1 or c % multiple == 0) else (c + multiple - 1) // multiple * multiple


def _fold(conv, bn, dtype, pad_multiple=1, pad_out=True):
    """Eval-mode BatchNorm folded into the convolution: (channels_last weight, fp32 shift or None).
    With pad_multiple > 1 the input (and, unless pad_out=False, output) channel counts are zero-padded to
    that multiple: the 196-wide maps of the (128, 196, 256) pyramid become 256-wide maps whose extra
    channels are identically zero through shift, ReLU / LeakyReLU, shortcut and upsampling."""
    w = conv.weight.detach().float()
    b = None
    if bn is not None:
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
        w = w * scale[:, None, None, None]
        b = bn.bias.detach().float() - bn.running_mean.float() * scale
    co, ci = w.shape[:2]
    cop, cip = (_padded(co, pad_multiple) if pad_out else co), _padded(ci, pad_multiple)
    if (cop, cip) != (co, ci):
        w = F.pad(w, (0, 0, 0, 0, 0, cip - ci, 0, cop - co))
        if b is not None:
            b = F.pad(b, (0, cop - co))
    return w.to(dtype).contiguous(memory_format=torch.channels_last), (None if b is None else b.contiguous())


class FusedInferenceBackbone:
    """Inference form of ResNetFPN_8_2 for the 16-bit modes: BatchNorm folded into the
    preceding convolution, channels_last throughout.  The 3x3 / stride-1 convolutions (13 of the 17, >90 % of
    the backbone's FLOPs) and, since round 4, the two 3x3 / stride-2 ones (GF_CONV_S2) run on K10 (csrc/k10_conv3x3.hip) with the BN
    shift, the BasicBlock shortcut add and ReLU / LeakyReLU in the kernel's epilogue; the 1x1 laterals and downsample shortcuts run on the
    K3 tile engine, the stem is its own kernel: in the 16-bit modes no convolution of the (128, 196, 256) pyramid goes to MIOpen any more
    (other widths, and the fp32 mode, still do: F.conv2d NHWC WITHOUT bias + the glue kernels of csrc/k_backbone_glue.hip).  Same arithmetic as resnet_fpn.py:85-118 in eval mode up to fp16 rounding points."""

    def __init__(self, bb: 'ResNetFPN_8_2', dtype, pad_multiple=32):
        self.dtype = dtype
        pm = pad_multiple
        self.stem = _fold(bb.conv1, bb.bn1, dtype)
        # 1 -> 128 channel 7x7 stem: own implicit-GEMM kernel (conv + shift + ReLU, NHWC out); other widths
        # go through MIOpen like the rest
        self.stem_hip = None
        if dtype in (torch.float16, torch.bfloat16) and tuple(bb.conv1.weight.shape) == (128, 1, 7, 7) and bb.conv1.stride == (2, 2):
            self.stem_hip = (_fold(bb.conv1, bb.bn1, torch.float32)[0].contiguous(), self.stem[1])
        self.blocks = []
        self._pad16 = {}
        self._rem8 = {}
        self._s2 = set()
        for layer in (bb.layer1, bb.layer2, bb.layer3):
            for blk in layer:
                w1, b1 = _fold(blk.conv1, blk.bn1, dtype, pm)
                w2, b2 = _fold(blk.conv2, blk.bn2, dtype, pm)
                wd = None
                if blk.downsample is not None:
                    wd, bd = _fold(blk.downsample[0], blk.downsample[1], dtype, pm)
                    wd = self._w1x1(wd)
                    b2 = (b2 + bd).contiguous()
                self.blocks.append((w1, b1, w2, b2, wd, blk.conv1.stride, self._stream(w1, blk.conv1.stride), self._stream(w2)))
        self.l3_out = self._w1x1(_fold(bb.layer3_outconv, None, dtype, pm)[0])
        self.l2_out = self._w1x1(_fold(bb.layer2_outconv, None, dtype, pm)[0])
        self.l1_out = self._w1x1(_fold(bb.layer1_outconv, None, dtype, pm)[0])
        self.l1_frags = None
        if dtype in (torch.float16, torch.bfloat16) and fused.lateral_supported(self.l1_out.shape[1], self.l1_out.shape[0]):
            self.l1_frags = fused.pack_lateral_frags(self.l1_out)
        self.l2_oc2 = (_fold(bb.layer2_outconv2[0], bb.layer2_outconv2[1], dtype, pm), bb.layer2_outconv2[2].negative_slope,
                       _fold(bb.layer2_outconv2[3], None, dtype, pm)[0])
        self.l1_oc2 = (_fold(bb.layer1_outconv2[0], bb.layer1_outconv2[1], dtype, pm), bb.layer1_outconv2[2].negative_slope,
                       _fold(bb.layer1_outconv2[3], None, dtype, pm, pad_out=False)[0])   # the fine map keeps its width
        self.l2_oc2 += (self._stream(self.l2_oc2[0][0]), self._stream(self.l2_oc2[2]))
        self.l1_oc2 += (self._stream(self.l1_oc2[0][0]), self._stream(self.l1_oc2[2]))

    def _w1x1(self, w):
        """1x1 kernels as contiguous [Cout, Cin] matrices for the K3 engine (16-bit modes); the fp32 mode keeps conv2d's 4-D form."""
        return w if self.dtype == torch.float32 else w.reshape(w.shape[0], w.shape[1]).contiguous()

    @staticmethod
    def _conv(x, w, stride=1):
        if w.dim() == 2:
            w = w[:, :, None, None]
        y = F.conv2d(x, w, None, stride, w.shape[-1] // 2)
        return y if y.is_contiguous(memory_format=torch.channels_last) else y.contiguous(memory_format=torch.channels_last)

    def _stream(self, w, stride=(1, 1)):
        """K10 fragment stream of a 3x3 convolution whose widths gf_conv3x3_nhwc is built for (16-bit modes; stride 1, or stride 2 =
        GF_CONV_S2: the first convolution of layer2 / layer3), else None (the convolution then goes through MIOpen + the glue kernel)."""
        if self.dtype == torch.float32 or tuple(w.shape[2:]) != (3, 3) or tuple(stride) not in ((1, 1), (2, 2)):
            return None
        if tuple(stride) == (2, 2):
            if not fused.conv3x3s2_supported(w.shape[1], w.shape[0]):
                return None
            self._s2.add(id(w))
            return fused.pack_conv3x3_stream(w, s2=True)
        if not fused.conv3x3_supported(w.shape[1], w.shape[0]):
            return None
        # 196 real input channels in a 224-wide map (zero weights from channel 196 on): channels 192 .. 199 as the 8-channel remainder
        # chunk (GF_CONV_REM8: 114 instead of 126 sub-steps per tile); built for Cout = 224 with padded outputs and for Cout = 128
        cout, cin = w.shape[:2]
        rem8 = bool(cin == 224 and not w[:, 200:].any() and ((cout == 224 and not w[196:].any()) or cout == 128))
        self._rem8[id(w)] = rem8
        return fused.pack_conv3x3_stream(w, rem8=rem8)

    def _conv3(self, x, w, ws, shift, shortcut, act, slope=0.01, stride=1):
        """act(conv(x, w) + shift + shortcut): one K10 launch when a stream exists."""
        if ws is not None:
            pad16 = self._pad16.get(id(w))
            if pad16 is None:             # 196 real channels in a 224-wide map: the output channels 196.. carry zero weights
                pad16 = self._pad16[id(w)] = bool(w.shape[0] == 224 and not w[196:].any())
            return fused.conv3x3(x, ws, w.shape[0], shift, shortcut, act, slope, pad16, self._rem8.get(id(w), False),
                                 2 if id(w) in self._s2 else 1)
        y = self._conv(x, w, stride)
        if shift is None and shortcut is None and act == ops.ACT_NONE:
            return y
        return ops.bias_act_(y, shift, shortcut, act, slope)

    def _conv1(self, x, w, stride=1):
        """1x1 convolution without bias: the K3 tile engine on the (strided) pixel rows in the 16-bit modes, MIOpen otherwise."""
        if self.dtype != torch.float32 and x.shape[1] % 32 == 0 and w.shape[0] % 32 == 0 and (stride == 1 or (x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)):
            return ops.conv1x1(x, w, stride)
        return self._conv(x, w, stride)

    def _block(self, x, p):
        w1, b1, w2, b2, wd, stride, s1, s2 = p
        y = self._conv3(x, w1, s1, b1, None, ops.ACT_RELU, stride=stride)
        shortcut = x if wd is None else self._conv1(x, wd, stride[0])
        return self._conv3(y, w2, s2, b2, shortcut, ops.ACT_RELU)

    def _head(self, x, p):
        (w0, b0), slope, w3, s0, s3 = p
        return self._conv3(self._conv3(x, w0, s0, b0, None, ops.ACT_LEAKY, slope), w3, s3, None, None, ops.ACT_NONE)

    def __call__(self, x):
        if self.stem_hip is not None:
            x = ops.stem_conv7x7(x.contiguous() if x.dtype in (torch.float32, self.dtype) else x.float().contiguous(), *self.stem_hip, dtype=self.dtype)
        else:
            x = x.to(self.dtype).contiguous(memory_format=torch.channels_last)
            x = ops.bias_act_(self._conv(x, self.stem[0], 2), self.stem[1], None, ops.ACT_RELU)
        x1 = self._block(self._block(x, self.blocks[0]), self.blocks[1])
        x2 = self._block(self._block(x1, self.blocks[2]), self.blocks[3])
        x3 = self._block(self._block(x2, self.blocks[4]), self.blocks[5])
        c3 = self._conv1(x3, self.l3_out)
        if self.l2_out.shape[1] % 32 == 0 and self.dtype != torch.float32:      # lateral 1x1 + merge in one K3 launch (K = 224: ragged K step)
            c2 = self._head(ops.conv1x1_upsample_add(x2, self.l2_out, c3), self.l2_oc2)
        else:
            c2 = self._head(ops.upsample_add_(self._conv(x2, self.l2_out), c3), self.l2_oc2)
        w1 = x1.shape[3]
        if self.l1_frags is not None and ((w1 % 16 == 0 and w1 == 2 * c2.shape[3]) or (self.dtype == torch.float16 and w1 % 2 == 0)):
            # lateral 1x1 + merge: the streaming kernel K12 (128 -> 224; bf16: its staged form only)
            c1 = self._head(fused.lateral_upsample_add(x1, self.l1_frags, self.l1_out.shape[0], c2), self.l1_oc2)
        elif self.l1_out.shape[1] % 32 == 0 and self.dtype != torch.float32:    # lateral 1x1 + merge in one K3 launch
            c1 = self._head(ops.conv1x1_upsample_add(x1, self.l1_out, c2), self.l1_oc2)
        else:
            c1 = self._head(ops.upsample_add_(self._conv(x1, self.l1_out), c2), self.l1_oc2)
        return [c3, c1]


def build_backbone(config):
    if config['backbone_type'] != 'ResNetFPN':
        raise ValueError(f"LOFTR.BACKBONE_TYPE {config['backbone_type']} not supported.")
    if tuple(config['resolution']) != (8, 2):
        raise ValueError('only the (8, 2) resolution used by GeoFormer is built')
    return ResNetFPN_8_2(config['resnetfpn'])

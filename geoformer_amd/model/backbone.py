"""ResNet-FPN 1/8 + 1/2 feature extractor.  Stays on PyTorch-ROCm / MIOpen by design (north_star;
SURVEY §2 row 10): only the module tree and parameter names are re-declared so that
`geoformer.ckpt` loads with the same keys as the reference's ResNetFPN_8_2
(model/loftr_src/loftr/backbone/resnet_fpn.py:43-118).
"""
import torch.nn as nn
import torch.nn.functional as F


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)


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
        c2 = self.layer2_outconv2(c2 + F.interpolate(c3, size=c2.shape[2:], mode='bilinear', align_corners=True))
        c1 = self.layer1_outconv(x1)
        c1 = self.layer1_outconv2(c1 + F.interpolate(c2, size=c1.shape[2:], mode='bilinear', align_corners=True))
        return [c3, c1]


def build_backbone(config):
    if config['backbone_type'] != 'ResNetFPN':
        raise ValueError(f"LOFTR.BACKBONE_TYPE {config['backbone_type']} not supported.")
    if tuple(config['resolution']) != (8, 2):
        raise ValueError('only the (8, 2) resolution used by GeoFormer is built')
    return ResNetFPN_8_2(config['resnetfpn'])

"""Row f4: the CNN trunks in front of the path, on the HIP kernels of csrc/conv_bf16.hip.

The reference builds its trunks from torchvision (MODEL:629 `models.resnet101(pretrained=...)`, MODEL:586-595
`models.__dict__['resnet50'](num_classes=365)` + the Places365 checkpoint) and keeps
`nn.Sequential(conv1, bn1, relu, maxpool, layer1, layer2, layer3, layer4)` of each (MODEL:274-294).  torchvision is a
third-party dependency that is neither vendored in the reference nor installed here, so this file restates the
published architecture (He et al. 2016 bottleneck ResNet; torchvision's "v1.5" places the stride of a stage's first block
on the 3x3 convolution) with torchvision's parameter names, which is what makes the checkpoints loadable:

  ResNet / Bottleneck   parameter containers with torchvision's state_dict surface (conv1, bn1, layer{1..4}.{i}.conv{1..3},
                        bn{1..3}, downsample.{0,1}, fc).  Their own forward() raises: there is no PyTorch compute path.
  ResNetFeatures        the reference's 8-entry nn.Sequential (same child indices -> same `object_features.4.0.conv1.weight`
                        keys), whose forward runs the HIP trunk: BatchNorm folded into bf16 weights once per parameter
                        version, NHWC bf16 activations, every convolution an implicit GEMM on the bf16 MFMA, the last
                        one writing the [B, 2048, h, w] fp32 NCHW map the fusion path consumes.

Any module with the same attribute structure (e.g. a torchvision ResNet on a machine that has torchvision) can be
wrapped: ResNetFeatures reads the geometry (stride / padding) from the nn.Conv2d children it is given.
"""
import torch
from torch import nn

from . import ops


class _ContainerOnly(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("%s holds parameters only: the trunk runs through mgnns_amd.trunk.ResNetFeatures "
                           "(HIP kernels); there is no PyTorch compute path" % type(self).__name__)


class Bottleneck(_ContainerOnly):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class ResNet(_ContainerOnly):
    """Bottleneck ResNet with torchvision.models.ResNet's attribute / state_dict names."""

    def __init__(self, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * Bottleneck.expansion, num_classes)    # unused by the path (MODEL:274-294 stops at layer4)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * Bottleneck.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * Bottleneck.expansion, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * Bottleneck.expansion))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * Bottleneck.expansion
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


def resnet50(num_classes=1000):
    """MODEL:589 `models.__dict__['resnet50'](num_classes=365)`."""
    return ResNet([3, 4, 6, 3], num_classes)


def resnet101(num_classes=1000):
    """MODEL:629 `models.resnet101(...)` (weights come from load_state_dict: there is no network here)."""
    return ResNet([3, 4, 23, 3], num_classes)


def _one(v):
    return v[0] if isinstance(v, (tuple, list)) else v


def _conv_geometry(conv, name):
    k, s, p = _one(conv.kernel_size), _one(conv.stride), _one(conv.padding)
    if (tuple(conv.kernel_size) != (k, k) or tuple(conv.stride) != (s, s) or tuple(conv.padding) != (p, p)
            or _one(conv.dilation) != 1 or conv.groups != 1 or k not in (1, 3)):
        raise ValueError("%s: only square 1x1 / 3x3, dilation 1, groups 1 convolutions are supported (got %s)" % (name, conv))
    return k, s, p


class ResNetFeatures(nn.Sequential):
    """MODEL:274-294 `nn.Sequential(conv1, bn1, relu, maxpool, layer1..layer4)` with a HIP forward.

    forward(img [B,3,H,W] fp32 NCHW on the GPU) -> [B, 2048, H/32, W/32] fp32 NCHW (eval mode only)."""

    def __init__(self, m):
        super().__init__(m.conv1, m.bn1, m.relu, m.maxpool, m.layer1, m.layer2, m.layer3, m.layer4)
        c, mp = m.conv1, m.maxpool
        if (tuple(c.kernel_size), tuple(c.stride), tuple(c.padding), c.in_channels, c.out_channels) != ((7, 7), (2, 2), (3, 3), 3, 64):
            raise ValueError("trunk stem must be Conv2d(3, 64, 7, stride 2, padding 3), got %s" % (c,))
        if (_one(mp.kernel_size), _one(mp.stride), _one(mp.padding)) != (3, 2, 1) or getattr(mp, "ceil_mode", False):
            raise ValueError("trunk max-pool must be MaxPool2d(3, 2, 1), got %s" % (mp,))
        self._plan = None
        self._plan_key = None

    # ---- one-off weight preparation, redone when a parameter or a running statistic changes ----
    def _tensors(self):
        return [t for t in list(self.parameters()) + list(self.buffers()) if t.is_floating_point()]

    @staticmethod
    def _fold(conv, bn, stem=False):
        return ops.conv_fold_bn(conv.weight.detach().contiguous(), None if conv.bias is None else conv.bias.detach(),
                                (bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps), stem=stem)

    def _prepare(self):
        key = tuple((t.data_ptr(), t._version) for t in self._tensors())
        if self._plan is not None and key == self._plan_key:
            return self._plan
        stem = self._fold(self[0], self[1], stem=True)
        blocks = []
        for li in range(4, 8):
            for bi, blk in enumerate(self[li]):
                name = "%d.%d" % (li, bi)
                convs = []
                for conv, bn in ((blk.conv1, blk.bn1), (blk.conv2, blk.bn2), (blk.conv3, blk.bn3)):
                    convs.append(self._fold(conv, bn) + _conv_geometry(conv, name))
                down = None
                if blk.downsample is not None:
                    dconv, dbn = blk.downsample[0], blk.downsample[1]
                    down = self._fold(dconv, dbn) + _conv_geometry(dconv, name + ".downsample")
                blocks.append((convs, down))
        self._plan, self._plan_key = (stem, blocks), key
        return self._plan

    def forward(self, img):
        if self.training:
            raise RuntimeError("ResNetFeatures: eval-mode forward only on the HIP path (BatchNorm uses running statistics); call .eval()")
        if img.dim() != 4 or img.shape[1] != 3:
            raise ValueError("trunk input must be [B, 3, H, W], got %s" % (tuple(img.shape),))
        stem, blocks = self._prepare()
        y = ops.stem_conv7(img.contiguous(), *stem)
        y = ops.maxpool3x3s2_nhwc(y)
        for i, (convs, down) in enumerate(blocks):
            last = i == len(blocks) - 1
            idn = y
            if down is not None:
                w, b, k, s, p = down
                idn = ops.conv_bf16_nhwc(y, w, b, k, s, p, relu=False)
            w, b, k, s, p = convs[0]
            o = ops.conv_bf16_nhwc(y, w, b, k, s, p)
            w, b, k, s, p = convs[1]
            o = ops.conv_bf16_nhwc(o, w, b, k, s, p)
            w, b, k, s, p = convs[2]
            y = ops.conv_bf16_nhwc(o, w, b, k, s, p, residual=idn, out_nchw_f32=last)
        return y


def features_flops(feats, size):
    """Algorithmic FLOPs (2 x MACs of the stem and of every bottleneck convolution) of ONE size x size image through a
    ResNetFeatures -- the figure the trunk's MFMA utilisation is quoted on (62.4 GFLOP for ResNet-101 at 448)."""
    h = (size - 1) // 2 + 1
    total = 2 * 64 * 147 * h * h
    h = (h - 1) // 2 + 1
    for li in range(4, 8):
        for blk in feats[li]:
            hin = h
            for conv in (blk.conv1, blk.conv2, blk.conv3):
                k, s, p = _one(conv.kernel_size), _one(conv.stride), _one(conv.padding)
                h = (h + 2 * p - k) // s + 1
                total += 2 * conv.out_channels * conv.in_channels * k * k * h * h
            if blk.downsample is not None:
                d = blk.downsample[0]
                ho = (hin - 1) // _one(d.stride) + 1
                total += 2 * d.out_channels * d.in_channels * ho * ho
    return total

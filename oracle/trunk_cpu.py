"""CPU restatement of the CNN trunks -- TEST INFRASTRUCTURE ONLY.

Imported only by tests/ (and tools/ that check the product); the product never imports it.

The reference takes its trunks from torchvision (Multi_GCN_Multihead_att.py:274-294, 586-595, 629-630:
`models.resnet101`, `models.__dict__['resnet50'](num_classes=365)`, cut after layer4).  torchvision is third-party, not
vendored under /root/reference, not pinned by any requirements file, and not installed in the build container, so this
row is **parity unpinned** with respect to torchvision itself: what is restated here is the published architecture
(He et al. 2016 bottleneck ResNet; torchvision >= 0.3 "v1.5" stride on the 3x3 convolution -- the geometry is read from the
checkpoint-independent structure below) on torch's own `conv2d` / `batch_norm` / `max_pool2d`, driven purely by a
state_dict with torchvision's key names.  The reference holds no test or golden vector for the trunks.

  features_fp32(sd, img)            eval forward in fp32 -> [B, 2048, h, w]
  features_bf16_emulated(sd, img)   the same network with the HIP path's rounding points (BatchNorm folded into the
                                    weights in fp32 then bf16; activations bf16 between layers; fp32 accumulation), so the
                                    kernels can be checked tightly; the remaining difference is summation order.
"""
import torch
import torch.nn.functional as F

STAGES = ("layer1", "layer2", "layer3", "layer4")


def _blocks(sd, prefix, layer):
    n = 0
    while "%s%s.%d.conv1.weight" % (prefix, layer, n) in sd:
        n += 1
    return n


def _bn(sd, name):
    return sd[name + ".weight"], sd[name + ".bias"], sd[name + ".running_mean"], sd[name + ".running_var"]


def _structure(sd, prefix=""):
    """[(block prefix, stride of the 3x3 conv, has downsample)] in forward order: a stage's first block carries the
    stride (1 for layer1, 2 otherwise), on conv2 (v1.5), and a 1x1 projection with the same stride."""
    out = []
    for li, layer in enumerate(STAGES):
        for b in range(_blocks(sd, prefix, layer)):
            p = "%s%s.%d." % (prefix, layer, b)
            out.append((p, 2 if (b == 0 and li > 0) else 1, (p + "downsample.0.weight") in sd))
    return out


def features_fp32(sd, img, prefix="", eps=1e-5):
    def cbn(x, conv, bn, stride, pad):
        g, b, m, v = _bn(sd, bn)
        return F.batch_norm(F.conv2d(x, sd[conv + ".weight"], None, stride, pad), m, v, g, b, False, 0.0, eps)

    x = F.relu(cbn(img, prefix + "conv1", prefix + "bn1", 2, 3))
    x = F.max_pool2d(x, 3, 2, 1)
    for p, stride, has_down in _structure(sd, prefix):
        idn = cbn(x, p + "downsample.0", p + "downsample.1", stride, 0) if has_down else x
        o = F.relu(cbn(x, p + "conv1", p + "bn1", 1, 0))
        o = F.relu(cbn(o, p + "conv2", p + "bn2", stride, 1))
        x = F.relu(cbn(o, p + "conv3", p + "bn3", 1, 0) + idn)
    return x


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def fold(sd, conv, bn, eps=1e-5):
    """(bf16-rounded folded weight as fp32 [Cout,Cin,KH,KW], fp32 bias) -- the arithmetic of conv_fold_bn_kernel."""
    g, b, m, v = _bn(sd, bn)
    scale = g / torch.sqrt(v + eps)
    return bf16_round(sd[conv + ".weight"] * scale[:, None, None, None]), b + (0.0 - m) * scale


def features_bf16_emulated(sd, img, prefix="", eps=1e-5, round_output=False):
    def cbn(x, conv, bn, stride, pad, res=None, relu=True, rnd=True):
        w, b = fold(sd, conv, bn, eps)
        y = F.conv2d(x, w, None, stride, pad) + b[None, :, None, None]
        if res is not None:
            y = y + res
        if relu:
            y = F.relu(y)
        return bf16_round(y) if rnd else y

    x = cbn(bf16_round(img), prefix + "conv1", prefix + "bn1", 2, 3)
    x = F.max_pool2d(x, 3, 2, 1)
    st = _structure(sd, prefix)
    for i, (p, stride, has_down) in enumerate(st):
        last = i == len(st) - 1
        idn = cbn(x, p + "downsample.0", p + "downsample.1", stride, 0, relu=False) if has_down else x
        o = cbn(x, p + "conv1", p + "bn1", 1, 0)
        o = cbn(o, p + "conv2", p + "bn2", stride, 1)
        x = cbn(o, p + "conv3", p + "bn3", 1, 0, res=idn, rnd=(not last) or round_output)
    return x

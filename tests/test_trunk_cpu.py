"""Row f4 on the CPU: the trunk containers' state_dict surface, the oracle's two forms against each other, the FLOP
count the MFMA utilisation is quoted on, and that the trunk has no PyTorch compute path."""
import numpy as np
import pytest
import torch

from mgnns_amd import synth, trunk
from oracle import trunk_cpu


def test_state_dict_surface_matches_reference_sequential_names():
    f = trunk.ResNetFeatures(trunk.resnet101())
    keys = set(f.state_dict().keys())
    # MODEL:274-283: children 0 = conv1, 1 = bn1, 4..7 = layer1..4
    for k in ("0.weight", "1.running_var", "4.0.conv1.weight", "4.0.downsample.0.weight", "4.0.downsample.1.running_mean",
              "6.22.bn3.weight", "7.2.conv3.weight"):
        assert k in keys, k
    assert len([k for k in keys if k.endswith("conv2.weight")]) == 33
    # torchvision's own key names on the un-cut network (what the published checkpoints hold)
    full = set(trunk.resnet50(num_classes=365).state_dict().keys())
    assert {"conv1.weight", "bn1.num_batches_tracked", "layer3.5.conv3.weight", "layer4.0.downsample.1.bias", "fc.weight"} <= full
    assert trunk.resnet50(num_classes=365).fc.weight.shape == (365, 2048)
    assert "layer3.6.conv1.weight" not in full and "layer3.22.conv1.weight" in trunk.resnet101().state_dict()


def test_stride_sits_on_the_3x3_convolution():
    m = trunk.resnet50()
    for li, layer in enumerate((m.layer1, m.layer2, m.layer3, m.layer4)):
        s = 1 if li == 0 else 2
        assert layer[0].conv1.stride == (1, 1) and layer[0].conv2.stride == (s, s) and layer[0].conv3.stride == (1, 1)
        assert layer[0].downsample[0].stride == (s, s)
        assert all(b.downsample is None and b.conv2.stride == (1, 1) for b in list(layer)[1:])


def test_features_flops_resnet101_448():
    f = trunk.ResNetFeatures(trunk.resnet101())
    assert abs(trunk.features_flops(f, 448) / 1e9 - 62.39) < 0.01          # 4 x the 224^2 figure of 7.8 GMAC
    assert abs(trunk.features_flops(trunk.ResNetFeatures(trunk.resnet50()), 224) / 1e9 - 8.17) < 0.05


def test_no_pytorch_compute_path():
    m = trunk.resnet50().eval()
    with pytest.raises(RuntimeError, match="holds parameters only"):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="holds parameters only"):
        m.layer1[0](torch.zeros(1, 64, 16, 16))
    f = trunk.ResNetFeatures(m)
    with pytest.raises(RuntimeError, match="eval-mode"):
        f.train()(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="no CPU path"):
        f.eval()(torch.zeros(1, 3, 64, 64))
    with pytest.raises(ValueError, match="stem must be"):
        bad = trunk.resnet50()
        bad.conv1 = torch.nn.Conv2d(3, 64, 3, stride=1, padding=1, bias=False)
        trunk.ResNetFeatures(bad)


def test_oracle_rounding_emulation_tracks_the_fp32_network():
    m = synth.fill_trunk_(trunk.resnet50(365), 5).eval()
    sd = m.state_dict()
    img = torch.from_numpy(np.random.RandomState(3).standard_normal((2, 3, 64, 64)).astype(np.float32))
    f32 = trunk_cpu.features_fp32(sd, img)
    emu = trunk_cpu.features_bf16_emulated(sd, img)
    assert tuple(f32.shape) == (2, 2048, 2, 2) and (f32 >= 0).all() and f32.max() > 0.1
    rel = ((emu - f32).pow(2).mean().sqrt() / f32.pow(2).mean().sqrt()).item()
    assert 1e-4 < rel < 3e-2                       # bf16 weights + activations through 16 blocks: ~1 %, and not identical

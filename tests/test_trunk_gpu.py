"""Row f4: the ResNet trunks on the HIP kernels (csrc/conv_bf16.hip) against the CPU restatement oracle/trunk_cpu.py
(torch conv2d / batch_norm / max_pool2d on the CPU).  The kernels compute on bf16 operands with fp32 accumulation and
store bf16 activations, so layer tests compare with the fp32 result on the SAME bf16-rounded operands (tolerance: one
bf16 rounding of the output, 2^-8 relative, written below); the whole-trunk tests use the oracle that emulates the
rounding points, and report the distance to the pure fp32 network."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mgnns_amd import ops, synth, trunk
from oracle import trunk_cpu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rnd(shape, seed, scale=1.0):
    return torch.from_numpy((scale * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _close_bf16(y, ref, what):
    """|y - ref| <= one bf16 rounding of ref (2^-8 relative) + 2^-9 of the tensor's scale (fp32 summation order near 0)"""
    tol = 2.0 ** -8 * ref.abs() + 2.0 ** -9 * ref.abs().max()
    bad = (y - ref).abs() > tol
    assert not bad.any(), "%s: %d / %d outside tolerance, max err %.3e (scale %.3e)" % (
        what, int(bad.sum()), bad.numel(), (y - ref).abs().max().item(), ref.abs().max().item())


def test_conv_fold_bn_is_the_oracle_fold():
    for stem, shape in ((False, (96, 64, 3, 3)), (False, (256, 128, 1, 1)), (True, (64, 3, 7, 7))):
        Cout = shape[0]
        sd = {"c.weight": _rnd(shape, 1, 0.1), "b.weight": _rnd((Cout,), 2).abs() + 0.3, "b.bias": _rnd((Cout,), 3, 0.2),
              "b.running_mean": _rnd((Cout,), 4, 0.2), "b.running_var": _rnd((Cout,), 5).abs() + 0.2}
        wref, bref = trunk_cpu.fold(sd, "c", "b", eps=1e-5)
        wt, bias = ops.conv_fold_bn(sd["c.weight"].to(DEV), None,
                                    (sd["b.weight"].to(DEV), sd["b.bias"].to(DEV), sd["b.running_mean"].to(DEV),
                                     sd["b.running_var"].to(DEV), 1e-5), stem=stem)
        wt = wt.float().cpu()
        if stem:
            assert wt.shape == (64, 160) and (wt[:, 147:] == 0).all()
            assert torch.equal(wt[:, :147], wref.reshape(64, 147))                      # k = (c, kh, kw)
        else:
            assert torch.equal(wt, wref.permute(0, 2, 3, 1).reshape(Cout, -1))          # k = (kh, kw, c), bit exact
        assert torch.allclose(bias.cpu(), bref, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("k,stride,Cin,Cout,B,H,W,res,relu,nchw", [
    (1, 1, 64, 64, 2, 56, 56, True, True, False),      # narrow tile (NJ = 2), one BK slice
    (1, 1, 64, 256, 3, 19, 23, True, True, False),     # ragged M (not a multiple of 256), residual
    (3, 1, 64, 64, 2, 28, 28, False, True, False),     # 3x3, padding taps through the zero source, narrow tile
    (3, 2, 128, 128, 2, 30, 30, False, True, False),   # stride on the 3x3 (torchvision v1.5)
    (3, 2, 128, 128, 1, 15, 17, False, False, False),  # odd sizes, no ReLU
    (1, 2, 256, 512, 2, 28, 28, False, False, False),  # strided 1x1 projection (downsample)
    (1, 1, 2048, 512, 2, 14, 14, False, True, False),  # K = 2048: 32 slices through the 3-stage ring
    (3, 1, 512, 512, 1, 14, 14, False, True, False),   # K = 4608
    (1, 1, 512, 2048, 2, 14, 14, True, True, True),    # last layer: fp32 NCHW output + residual
    (1, 1, 256, 1024, 1, 3, 5, True, True, True),      # tiny M (one partial tile), NCHW
    (1, 1, 128, 104, 2, 9, 9, False, True, False),     # C_out not a multiple of the tile
])
def test_conv_igemm_vs_conv2d_on_rounded_operands(k, stride, Cin, Cout, B, H, W, res, relu, nchw):
    pad = k // 2
    x = _bf(_rnd((B, Cin, H, W), 10))
    w = _bf(_rnd((Cout, Cin, k, k), 11, (2.0 / (Cin * k * k)) ** 0.5))
    bias = _rnd((Cout,), 12, 0.3)
    ref = F.conv2d(x.double(), w.double(), bias.double(), stride, pad)
    r = None
    if res:
        r = _bf(_rnd(tuple(ref.shape), 13))
        ref = ref + r.double()
    if relu:
        ref = F.relu(ref)
    ref = ref.float()
    wt = w.permute(0, 2, 3, 1).reshape(Cout, -1).to(torch.bfloat16).contiguous().to(DEV)
    xg = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    rg = None if r is None else r.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    y = ops.conv_bf16_nhwc(xg, wt, bias.to(DEV), k, stride, pad, residual=rg, relu=relu, out_nchw_f32=nchw)
    torch.cuda.synchronize()
    if nchw:
        assert y.dtype == torch.float32 and tuple(y.shape) == tuple(ref.shape)
        assert torch.allclose(y.cpu(), ref, rtol=1e-5, atol=2e-5 * ref.abs().max().item())      # fp32 out: summation order only
    else:
        assert y.dtype == torch.bfloat16
        _close_bf16(y.float().cpu().permute(0, 3, 1, 2), ref, "conv %dx%d s%d %d->%d" % (k, k, stride, Cin, Cout))


def test_conv_argument_errors():
    x = torch.zeros(1, 4, 4, 96, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="power of two"):
        ops.conv_bf16_nhwc(x, torch.zeros(64, 96, dtype=torch.bfloat16, device=DEV), torch.zeros(64, device=DEV), 1)
    x = torch.zeros(1, 4, 4, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(ValueError, match="do not match"):
        ops.conv_bf16_nhwc(x, torch.zeros(64, 128, dtype=torch.bfloat16, device=DEV), torch.zeros(64, device=DEV), 1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.conv_bf16_nhwc(x.cpu(), torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(64), 1)
    with pytest.raises(RuntimeError, match="holds parameters only"):
        trunk.resnet50()(torch.zeros(1, 3, 32, 32))


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (1, 75, 53), (2, 224, 224)])
def test_stem_and_maxpool(B, H, W):
    img = _rnd((B, 3, H, W), 20)
    sd = {"conv1.weight": _rnd((64, 3, 7, 7), 21, 0.12), "bn1.weight": _rnd((64,), 22).abs() + 0.4, "bn1.bias": _rnd((64,), 23, 0.2),
          "bn1.running_mean": _rnd((64,), 24, 0.2), "bn1.running_var": _rnd((64,), 25).abs() + 0.3}
    w, b = trunk_cpu.fold(sd, "conv1", "bn1")
    ref = F.relu(F.conv2d(_bf(img).double(), w.double(), b.double(), 2, 3)).float()
    wt, bias = ops.conv_fold_bn(sd["conv1.weight"].to(DEV), None, tuple(sd["bn1." + n].to(DEV) for n in
                                ("weight", "bias", "running_mean", "running_var")) + (1e-5,), stem=True)
    y = ops.stem_conv7(img.to(DEV), wt, bias)
    assert tuple(y.shape) == (B, ref.shape[2], ref.shape[3], 64)
    _close_bf16(y.float().cpu().permute(0, 3, 1, 2), ref, "stem")
    p = ops.maxpool3x3s2_nhwc(y)
    pref = F.max_pool2d(y.float().cpu().permute(0, 3, 1, 2), 3, 2, 1)
    assert torch.equal(p.float().cpu().permute(0, 3, 1, 2), pref)              # max of bf16 values: exact


def _trunk_pair(ctor, salt):
    m = synth.fill_trunk_(ctor(), salt).eval()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    return trunk.ResNetFeatures(m).to(DEV).eval(), sd


@pytest.mark.parametrize("name,ctor,B,S", [("resnet50", lambda: trunk.resnet50(365), 3, 96),
                                           ("resnet101", trunk.resnet101, 1, 128)])
def test_trunk_matches_rounding_emulating_oracle(name, ctor, B, S):
    feats, sd = _trunk_pair(ctor, 5)
    img = _rnd((B, 3, S, S), 30)
    y = feats(img.to(DEV))
    torch.cuda.synchronize()
    ref = trunk_cpu.features_bf16_emulated(sd, img)
    f32 = trunk_cpu.features_fp32(sd, img)
    assert tuple(y.shape) == tuple(ref.shape) == (B, 2048, S // 32, S // 32) and y.dtype == torch.float32
    scale = f32.abs().max().item()
    err = (y.cpu() - ref).abs().max().item() / scale
    err32 = (y.cpu() - f32).abs().max().item() / scale
    rms32 = ((y.cpu() - f32).pow(2).mean().sqrt() / f32.pow(2).mean().sqrt()).item()
    print("%s: max err vs emulated oracle %.3e, vs fp32 network %.3e (rms %.3e) of max |f32| = %.3g" % (name, err, err32, rms32, scale))
    # same rounding points: what is left are bf16 ulp flips from the fp32 summation order, propagated through the blocks
    assert err < 2e-2
    assert rms32 < 3e-2                                  # bf16 activations + weights through 16 / 33 blocks


def test_full_size_trunk_448_properties():
    """BASELINE-size input (448 x 448 -> the [B, 2048, 14, 14] map of SURVEY 8a5/a6): shape, finiteness, ReLU range,
    batch independence (image i alone == image i in the batch, bit exact: no cross-sample arithmetic)."""
    feats, _ = _trunk_pair(trunk.resnet101, 7)
    img = _rnd((3, 3, 448, 448), 31).to(DEV)
    y = feats(img)
    assert tuple(y.shape) == (3, 2048, 14, 14)
    assert torch.isfinite(y).all() and (y >= 0).all() and y.max() > 0
    y1 = feats(img[1:2].contiguous())
    assert torch.equal(y1[0], y[1])


def test_model_from_raw_images_equals_model_from_trunk_features():
    """ENGINE:825 hands the model raw images; the benchmark entry hands it the [B,2048,14,14] maps.  Both
    must be the same computation: run the two trunks alone, feed their maps to the model, compare with the model run
    on the images (bit exact), and check the state_dict carries the reference's trunk key names."""
    from mgnns_amd.harness import build_model, call_args, synthetic_adjacencies
    cfg = synth.CONFIGS["mvsa_multiple_b256"]
    pmi, count = synth.synth_pmi(cfg.V, seed=91)
    A_obj, A_place = synthetic_adjacencies(cfg)
    B = 4
    inp = synth.make_inputs(cfg, B=B, seed=99, pmi=pmi)
    model = build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], DEV, trunks=True)
    keys = model.state_dict().keys()
    assert "object_features.6.22.conv3.weight" in keys and "place_features.6.5.bn3.running_var" in keys
    assert "place_features.6.6.conv1.weight" not in keys                      # ResNet-50: six blocks in layer3
    args = list(call_args(inp, DEV))
    imgs_o, imgs_p = _rnd((B, 3, 448, 448), 41).to(DEV), _rnd((B, 3, 448, 448), 42).to(DEV)
    fo, fp = model.object_features(imgs_o), model.place_features(imgs_p)
    assert tuple(fo.shape) == tuple(fp.shape) == (B, 2048, 14, 14)
    args[3], args[4] = fo, fp
    ref = model(*args)
    args[3], args[4] = imgs_o, imgs_p
    out = model(*args)
    assert torch.isfinite(out).all()
    assert torch.equal(out, ref)
    # images -> logits as ONE hipGraph (both trunks + the path, four streams): replay == eager
    from mgnns_amd.graph import GraphedForward
    g = GraphedForward(model, args)
    assert torch.equal(g.replay(), ref)
    imgs2 = _rnd((B, 3, 448, 448), 43).to(DEV)
    args2 = list(args)
    args2[3] = imgs2
    assert torch.equal(g(*args2), model(*args2))

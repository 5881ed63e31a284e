"""Trunk micro-benchmark (row f4): ResNet-101 / ResNet-50 features at 448 x 448 on the HIP implicit-GEMM kernels.

    python tools/bench_trunk.py [--batch 32] [--arch resnet101] [--iters 5] [--per-layer]

Prints images/s, achieved TFLOP/s (algorithmic 2*MACs of every convolution) and, with --per-layer, the time per
distinct convolution geometry (HIP events around each launch; serialises the stream, so the sum exceeds the wall time).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mgnns_amd import ops, synth, trunk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=448)
    ap.add_argument("--arch", default="resnet101")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--per-layer", action="store_true")
    a = ap.parse_args()
    dev = "cuda:0"
    m = synth.fill_trunk_(getattr(trunk, a.arch)(), 3).eval()
    feats = trunk.ResNetFeatures(m).to(dev).eval()
    img = torch.randn(a.batch, 3, a.size, a.size, device=dev)
    fl = trunk.features_flops(feats, a.size)
    for _ in range(2):
        y = feats(img)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        y = feats(img)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    out = {"arch": a.arch, "batch": a.batch, "size": a.size, "ms": round(ms, 3), "images_per_s": round(a.batch / ms * 1e3, 1),
           "gflop_per_image": round(fl / 1e9, 2), "tflops": round(fl * a.batch / ms / 1e9, 1),
           "mfma_frac_of_2.5PF": round(fl * a.batch / ms / 1e9 / 2500, 4), "out": list(y.shape)}
    print(json.dumps(out))
    if a.per_layer:
        t = ops.KernelTimer()
        ops.set_timer(t)
        feats(img)
        torch.cuda.synchronize()
        ops.set_timer(None)
        rows = []
        for key, v in t.durations_ms().items():
            rows.append((sum(v), len(v), key))
        tot = sum(r[0] for r in rows)
        for s, n, key in sorted(rows, reverse=True):
            extra = ""
            if key[0] == "mgnns_conv_bf16_nhwc_fwd":
                _, k, cin, cout, stride, oh = key
                f = 2.0 * a.batch * oh * oh * cout * cin * k * k * n
                byts = a.batch * oh * oh * 2.0 * n * (cout * 2 + cin * (stride * stride if k == 1 else 1))
                extra = "  %7.1f TFLOP/s  >=%6.0f GB/s" % (f / s / 1e9, byts / s / 1e6)
            print("%8.3f ms %5.1f%%  x%-3d %s%s" % (s, 100 * s / tot, n, key, extra))
        print("sum %.3f ms" % tot)


if __name__ == "__main__":
    main()

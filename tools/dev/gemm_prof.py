"""A few launches of the dense bf16 GEMM of configs[4] (adjacency [n, n] . X [n, F]) for rocprofv3 (--pmc / --kernel-trace):
    python3 tools/dev/gemm_prof.py <F> [launches] [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mgnns_amd import ops  # noqa: E402

dev = "cuda:0"
F = int(sys.argv[1])
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
g = torch.Generator(device=dev).manual_seed(0)
kp = (n + 63) // 64 * 64
adjs = [torch.zeros(n, kp, device=dev, dtype=torch.bfloat16) for _ in range(2)]
for a in adjs:
    a[:, :n] = torch.randn(n, n, device=dev, generator=g).bfloat16()
xts = [torch.zeros(F, kp, device=dev, dtype=torch.bfloat16) for _ in range(2)]
for x in xts:
    x[:, :n] = torch.randn(F, n, device=dev, generator=g).bfloat16()
out = torch.empty(n, F, device=dev)
for i in range(launches):
    ops.gemm_bf16_nt(adjs[i % 2], xts[i % 2], out=out)
torch.cuda.synchronize()
print("done", n, F)

"""Cache-cold timing of one SpMM configuration (MGNNS_SPMM_EXP, experiment build) at N = 10 000, density 4e-4."""
import sys, os, json, torch
sys.path.insert(0, ".")
from mgnns_amd import ops, stress
dev = "cuda:0"; n = 10000
g = torch.Generator(device=dev).manual_seed(0)
csr_np = stress.random_csr(n, 4e-4, 1); csr = stress.csr_to_device(csr_np, dev); nnz = csr_np[1].size
res = {}
for F in [int(f) for f in os.environ.get("SPMM_F", "1024,2048").split(",")]:
    by = nnz * 8.0 + 2.0 * n * F * 4
    k = stress._sets_for(2.0 * n * F * 4)
    xs = [torch.randn(n, F, device=dev, generator=g) for _ in range(k)]
    ys = [torch.empty_like(x) for x in xs]
    run = lambda x, y: ops.spmm_csr(csr, x, act=ops.ACT_LRELU2, out=y)
    ms = min(stress.time_cold(run, list(zip(xs, ys))) for _ in range(3))
    res[F] = (round(ms * 1e3, 1), round(by / ms / 1e6))
    del xs, ys
print(os.environ.get("MGNNS_SPMM_EXP", "default"), json.dumps(res))

import sys, json, torch
sys.path.insert(0, ".")
from mgnns_amd import ops, stress
dev = "cuda:0"
n = 10000
g = torch.Generator(device=dev).manual_seed(0)
out = {}
for dens in stress.DENSITIES:
    csr_np = stress.random_csr(n, dens, 1); csr = stress.csr_to_device(csr_np, dev); nnz = csr_np[1].size
    for F in (1024, 2048):
        by = nnz * 8.0 + 2.0 * n * F * 4
        k = stress._sets_for(2.0 * n * F * 4)
        xs = [torch.randn(n, F, device=dev, generator=g) for _ in range(k)]
        ys = [torch.empty_like(x) for x in xs]
        run = lambda x, y: ops.spmm_csr(csr, x, act=ops.ACT_LRELU2, out=y)
        ms = min(stress.time_cold(run, list(zip(xs, ys))) for _ in range(3))
        msw = stress.time_warm(run, (xs[0], ys[0]))
        msc = min(stress.time_cold(lambda d, s: d.copy_(s), list(zip(ys, xs))) for _ in range(2))
        out["d%g_F%d" % (dens, F)] = (round(ms * 1e3, 1), round(by / ms / 1e6), round(msw * 1e3, 1), round(by / msw / 1e6), "copy", round(2.0 * n * F * 4 / msc / 1e6))
        del xs, ys
print(json.dumps(out))

"""Where does the cold SpMM lose against a copy?  Same kernel on (a) the random graph, (b) the identity (sequential gathers),
(c) a random PERMUTATION (one random gather per row), (d) 4 nnz per row all in the diagonal band (local gathers)."""
import sys, os, json, torch, numpy as np
sys.path.insert(0, ".")
from mgnns_amd import ops, stress
dev = "cuda:0"; n = 10000
g = torch.Generator(device=dev).manual_seed(0)
rs = np.random.RandomState(0)
def to_dev(rp, col, val):
    return (torch.from_numpy(rp.astype(np.int32)).to(dev), torch.from_numpy(col.astype(np.int32)).to(dev), torch.from_numpy(val.astype(np.float32)).to(dev))
graphs = {}
graphs["random 4e-4"] = stress.csr_to_device(stress.random_csr(n, 4e-4, 1), dev)
graphs["identity"] = to_dev(np.arange(n + 1), np.arange(n), np.ones(n))
graphs["permutation"] = to_dev(np.arange(n + 1), rs.permutation(n), np.ones(n))
band = np.sort((np.arange(n)[:, None] + np.array([0, 1, 2, 3])[None, :]) % n, axis=1)
graphs["band 4"] = to_dev(np.arange(0, 4 * n + 1, 4), band.reshape(-1), np.ones(4 * n))
rnd4 = np.sort(rs.randint(0, n, size=(n, 4)), axis=1)
graphs["random exactly 4/row"] = to_dev(np.arange(0, 4 * n + 1, 4), rnd4.reshape(-1), np.ones(4 * n))
F = 1024
k = stress._sets_for(2.0 * n * F * 4)
xs = [torch.randn(n, F, device=dev, generator=g) for _ in range(k)]
ys = [torch.empty_like(x) for x in xs]
for name, csr in graphs.items():
    nnz = int(csr[1].numel())
    by = nnz * 8.0 + 2.0 * n * F * 4
    run = lambda x, y: ops.spmm_csr(csr, x, act=ops.ACT_LRELU2, out=y)
    ms = min(stress.time_cold(run, list(zip(xs, ys))) for _ in range(3))
    print("%-22s nnz %6d: %.1f us  %.0f GB/s" % (name, nnz, ms * 1e3, by / ms / 1e6))
msc = min(stress.time_cold(lambda d, s: d.copy_(s), list(zip(ys, xs))) for _ in range(2))
print("copy: %.1f us %.0f GB/s" % (msc * 1e3, 2.0 * n * F * 4 / msc / 1e6))

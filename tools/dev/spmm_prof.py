"""A few cache-cold launches of one bf16 SpMM variant for rocprofv3 (--pmc / --kernel-trace):
    python3 tools/dev/spmm_prof.py <graph: random|identity|perm|rnd4|dense> <F> <path: direct|tiled> <variant> [launches]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mgnns_amd import ops, stress
dev = "cuda:0"; n = 10000
kind, F, path, variant = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
launches = int(sys.argv[5]) if len(sys.argv) > 5 else 34
rs = np.random.RandomState(0)
def graph(kind):
    if kind == "random": return stress.random_csr(n, 4e-4, 1)
    if kind == "dense": return stress.random_csr(n, 1e-2, 1)
    if kind == "identity": return (np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), np.ones(n, np.float32))
    if kind == "perm": return (np.arange(n + 1, dtype=np.int32), rs.permutation(n).astype(np.int32), np.ones(n, np.float32))
    if kind == "rnd4": return (np.arange(0, 4 * n + 1, 4, dtype=np.int32), np.sort(rs.randint(0, n, size=(n, 4)), axis=1).reshape(-1).astype(np.int32), np.ones(4 * n, np.float32))
adj = ops.SparseAdjBf16(stress.csr_to_device(graph(kind), dev))
g = torch.Generator(device=dev).manual_seed(0)
k = max(4, -(-stress.COLD_BYTES // (2 * n * F * 2)))
xs = [torch.randn(n, F, device=dev, generator=g).bfloat16() for _ in range(k)]
ys = [torch.empty_like(x) for x in xs]
for i in range(launches):
    if path in ("tiled", "sell"):
        ops.spmm_bf16(adj, xs[i % k], act=ops.ACT_LRELU2, out=ys[i % k], path=path)
    else:
        ops.spmm_bf16(adj, xs[i % k], act=ops.ACT_LRELU2, out=ys[i % k], path="direct", variant=variant)
torch.cuda.synchronize()
print("done", kind, F, path, variant, adj.nnz)

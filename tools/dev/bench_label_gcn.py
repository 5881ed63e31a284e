"""Stand-alone timing of the persistent label-GCN launch vs the chain of separate operators (GPU box)."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from mgnns_amd import ops, synth
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.05
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
w1, w2 = rn(300, 1024), rn(1024, 2048)
lq, wq, bq = rn(7, 300), rn(300, 300), rn(300)
for C, edges in ((80, 53), (365, 217)):
    A = torch.as_tensor(synth.synth_adjacency(C, edges, 5)).to(dev)
    X = rn(C, 300)
    def chain():
        _, csr = ops.gen_adj(A, want_csr=True)
        G = ops.spmm_csr(csr, ops.matmul(ops.spmm_csr(csr, ops.matmul(X, w1), act=ops.ACT_LRELU2), w2))
        return ops.pack_weight_bf16_split(G), ops.linear(lq, wq, bq)
    print("C=%d: separate operators %.1f us" % (C, t(chain)))
    for split in (False, True):
        pk = ops.label_gcn_pack(w1, w2, split)
        for grid in (16, 32, 64, 128, 256):
            print("C=%d %s grid=%d: %.1f us" % (C, "split-bf16" if split else "exact-f32", grid,
                                                t(lambda: ops.label_gcn(A, X, pk, want_packed_g=True, query=(lq, wq, bq), grid=grid))))

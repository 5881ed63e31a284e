import sys, torch, numpy as np
sys.path.insert(0, ".")
from mgnns_amd import ops
dev = "cuda:0"; B = 256; Hn = 8
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.05
sp = lambda w: ops.pack_weight_bf16_split(w.contiguous())
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
pk = {"fc_b": rn(300), "g1": rn(300) + 1, "be1": rn(300), "b1": rn(300), "b2": rn(300), "g2": rn(300) + 1, "be2": rn(300),
      "fc": sp(rn(300, 1024)), "w1": sp(rn(300, 300)), "w2": sp(rn(300, 300))}
nx = (sp(rn(1024, 300)), rn(1024), 1024)
wp = ops.pack_kv_weights_bf16(rn(1024, 300), rn(1024, 300), Hn, 128)
bk, bv = rn(1024), rn(1024)
counters = torch.zeros(64, dtype=torch.int32, device=dev)
for L in (196, 100):
    bank = ops.cast_pad_bf16(torch.randn(B, L, 300, device=dev, generator=g))
    qh = torch.randn(B, 1024, device=dev, generator=g); q = torch.randn(B, 300, device=dev, generator=g)
    def sep():
        o, _ = ops.sq_mha_core_bf16(qh, bank, None, Hn, 128, wp, bk, bv, want_attn=False)
        return ops.mha_tail_bf16(o, q, pk, 1e-6, nx, terms=1)
    print("L=%d: core %.1f us, core+tail separate %.1f us, fused layer %.1f us" % (
        L, t(lambda: ops.sq_mha_core_bf16(qh, bank, None, Hn, 128, wp, bk, bv, want_attn=False)), t(sep),
        t(lambda: ops.sq_mha_layer_bf16(qh, bank, None, Hn, 128, wp, bk, bv, q, pk, 1e-6, counters, nx))))

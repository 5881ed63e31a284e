#!/usr/bin/env python
"""A few launches of sq_mha_core_bf16 alone (B=256, L=196, H=8) -- the program behind rocprofv3 PMC passes (tools/dev/mha_pmc.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mgnns_amd import ops  # noqa: E402

DEV = "cuda:0"
B, L, H = 256, 196, 8
g = torch.Generator(device=DEV).manual_seed(0)
bank = ops.cast_pad_bf16(torch.randn(B, L, 300, device=DEV, generator=g))
qh = torch.randn(B, H * 128, device=DEV, generator=g)
wk = torch.randn(H * 128, 300, device=DEV, generator=g) * 0.05
wv = torch.randn(H * 128, 300, device=DEV, generator=g) * 0.05
bk = torch.zeros(H * 128, device=DEV)
wp = ops.pack_kv_weights_bf16(wk, wv, H, 128)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    ops.sq_mha_core_bf16(qh, bank, None, H, 128, wp, bk, bk, want_attn=False)
torch.cuda.synchronize()

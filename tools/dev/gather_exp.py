"""Gather throughput out of the XCD's L2: k pseudo-random 256-B (or 512-B) piece reads per piece slot, no writes, WARM (one
buffer, 2.5 MB per XCD) and cold (rotation)."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mgnns_amd import _lib, stress
from tools.dev.slabcopy_exp_lib import time_graph
dev = "cuda:0"; n = 10000
L = _lib.lib()
for pitch in (2048, 4096):
    kk = max(4, -(-stress.COLD_BYTES // (n * pitch)))
    xs = [torch.randint(0, 255, (n, pitch), device=dev, dtype=torch.uint8) for _ in range(kk)]
    y = torch.empty(64, device=dev, dtype=torch.uint8)
    for piece in (256, 512):
        for k in (1, 2, 4, 8):
            for wgx in (128, 256, 512):
                m = 3 | (k << 12)
                def run(x):
                    _lib.check(L.mgnns_debug_slabcopy(x.data_ptr(), y.data_ptr(), n, pitch, piece, m, wgx, torch.cuda.current_stream().cuda_stream), "slabcopy")
                warm = time_graph(run, [(xs[0],)] * 8)
                cold = time_graph(run, [(x,) for x in xs])
                by = n * pitch * k
                print(json.dumps({"pitch": pitch, "piece": piece, "k": k, "wgx": wgx, "warm_us": round(warm * 1e3, 2), "warm_GBps": round(by / warm / 1e6), "warm_B_per_clk_per_CU": round(by / warm / 1e6 / 256 / 2.1, 1),
                                  "cold_us": round(cold * 1e3, 2), "cold_GBps": round(by / cold / 1e6)}), flush=True)

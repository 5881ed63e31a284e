"""A few eager single-stream forwards in a given precision / attention mode, for rocprofv3 --kernel-trace --stats:
    python3 tools/dev/prof_mode.py <fp32|bf16|bf16x3> <faithful|folded> [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mgnns_amd import harness, synth
prec, attn = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
cfg = synth.CONFIGS["mvsa_multiple_b256"]; dev = "cuda:0"
pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
A_obj, A_place = harness.synthetic_adjacencies(cfg)
inp = synth.make_inputs(cfg, B=B, seed=cfg.seed, pmi=pmi)
model = harness.build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], dev)
model.set_precision(prec).set_attention(attn)
model.use_streams = False
call = harness.call_args(inp, dev)
with torch.no_grad():
    for _ in range(12):
        model(*call)
torch.cuda.synchronize()

"""Where do the __amd_rocclr_copyBuffer dispatches of a forward come from?  torch.profiler with stacks on one eager forward."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from mgnns_amd import harness, synth
cfg = synth.CONFIGS["mvsa_multiple_b256"]; dev = "cuda:0"
pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
A_obj, A_place = harness.synthetic_adjacencies(cfg)
inp = synth.make_inputs(cfg, B=256, seed=cfg.seed, pmi=pmi)
model = harness.build_model(cfg, pmi, count, A_obj, A_place, inp["label_query"], dev)
model.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
call = harness.call_args(inp, dev)
with torch.no_grad():
    for _ in range(3): model(*call)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        model(*call); torch.cuda.synchronize()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::cat", "aten::amax", "aten::max", "aten::contiguous", "aten::clone", "aten::_to_copy") or "Memcpy" in ev.name or "copyBuffer" in ev.name:
        st = [f for f in (ev.stack or []) if "mgnns_amd" in f or "bench" in f][:3]
        print(ev.name, ev.input_shapes if hasattr(ev, "input_shapes") else "", "|", " <- ".join(st))

from collections import Counter
c = Counter(ev.name for ev in prof.events())
print("EVENTS", len(prof.events()))
for k, v in c.most_common(40): print("  ", v, k[:100])

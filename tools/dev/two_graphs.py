import sys, time, torch
sys.path.insert(0, ".")
from mgnns_amd import harness, synth
from mgnns_amd.graph import GraphedForward
dev = torch.device("cuda:0")
cfg = synth.CONFIGS["mvsa_multiple_b256"]
pmi, count = synth.synth_pmi(cfg.V, seed=cfg.seed + 17)
A_obj, A_place = harness.synthetic_adjacencies(cfg)
lq = synth.make_inputs(cfg, B=1, seed=cfg.seed, pmi=pmi)["label_query"]
model = harness.build_model(cfg, pmi, count, A_obj, A_place, lq, dev)
model.set_precision("bf16")
def run(B, keep=None):
    inp = synth.make_inputs(cfg, B=B, seed=5, pmi=pmi)
    call = harness.call_args(inp, dev)
    gf = GraphedForward(model, call)
    for _ in range(3): gf.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): gf.replay()
    torch.cuda.synchronize()
    print("B=%d mode=%s %.3f ms/step" % (B, gf.mode, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
    return gf
g1 = run(256)
g2 = run(128)
del g1
g3 = run(128)
g4 = run(32)

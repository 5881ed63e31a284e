"""What the slab access pattern costs without any sparse matrix: cache-cold copies / reads / writes of a [10000, pitch] matrix
in XCD-dealt pieces (mgnns_debug_slabcopy), graph-timed over a rotation of >640 MiB."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mgnns_amd import _lib, stress
dev = "cuda:0"; n = 10000
L = _lib.lib()
def time_graph(fn, arg_sets, reps=5):
    for a in arg_sets: fn(*a)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            for a in arg_sets: fn(*a)
    torch.cuda.synchronize(); gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps): gr.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (reps * len(arg_sets)))
    return best
for pitch in (2048, 4096):
    k = max(4, -(-stress.COLD_BYTES // (2 * n * pitch)))
    xs = [torch.randint(0, 255, (n, pitch), device=dev, dtype=torch.uint8) for _ in range(k)]
    ys = [torch.empty_like(x) for x in xs]
    ms = time_graph(lambda d, s_: d.copy_(s_), list(zip(ys, xs)))
    print(json.dumps({"pitch": pitch, "torch_copy_us": round(ms * 1e3, 2)}), flush=True)
    for mode, name in ((0, "copy"), (1, "read"), (2, "write"), (16, "copy_plainstore")):
        for piece, lin in ((256, 0), (512, 0), (1024, 0), (2048, 0), (256, 1), (2048, 1)):
            if piece > pitch: continue
            for wgx in (64, 128, 256, 512):
                m = mode | (256 if lin else 0)
                def run(x, y):
                    _lib.check(L.mgnns_debug_slabcopy(x.data_ptr(), y.data_ptr(), n, pitch, piece, m, wgx, torch.cuda.current_stream().cuda_stream), "slabcopy")
                ms = time_graph(run, list(zip(xs, ys)))
                by = n * pitch * (2 if name.startswith("copy") else 1)
                print(json.dumps({"pitch": pitch, "what": name, "piece": piece, "linear": lin, "wgx": wgx, "us": round(ms * 1e3, 2), "GBps": round(by / ms / 1e6)}), flush=True)
    del xs, ys

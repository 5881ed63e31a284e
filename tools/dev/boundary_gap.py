#!/usr/bin/env python
"""How long after an event fires does dependent work on ANOTHER stream finish -- launched as a hipGraph vs. as a plain kernel?
(The forward replays 13 per-segment graphs with event waits in between: tools/trace_timeline.py shows 15-28 us at the boundaries.)"""
import torch

dev = torch.device("cuda:0")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros(1024, device=dev)
y = torch.zeros(1024, device=dev)

# a one-kernel graph on s2
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s2):
    y.add_(1.0)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s2):
        y.add_(1.0)
torch.cuda.synchronize()


def trial(mode, sleep_cycles=600000):
    eA, eB = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s1):
        torch.cuda._sleep(sleep_cycles)
        eA.record(s1)
    with torch.cuda.stream(s2):
        s2.wait_event(eA)
        if mode == "graph":
            g.replay()
        else:
            y.add_(1.0)
        eB.record(s2)
    torch.cuda.synchronize()
    return eA.elapsed_time(eB) * 1e3


for mode in ("kernel", "graph", "kernel", "graph"):
    v = sorted(trial(mode) for _ in range(30))
    print("%-6s after an event on another stream: median %.1f us  (min %.1f, p90 %.1f)" % (mode, v[15], v[0], v[27]))

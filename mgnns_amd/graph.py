"""Whole-forward hipGraph: capture one forward (all streams, ~130 kernel launches) once, replay it per batch.

The eval forward is static for a given batch shape -- no host-side decisions, no device-to-host syncs (the
reference's Text_GCN forward syncs and loops in Python per document, Text_GCN.py:232-234) -- so it is captured
with PyTorch's HIP-graph support: kernels launched through the C ABI go to the capturing stream like any other
launch.  Inputs live in static device buffers that the caller (or `copy_inputs`) fills before each replay.
"""
import torch


class GraphedForward:
    def __init__(self, model, example_args, warmup=3, post=None):
        """example_args: the 7 forward arguments on the GPU (text_lens included, int64 on the device).
        post: optional callable applied to the logits INSIDE the capture (e.g. ShardedForward.gather: the RCCL
        all-gather becomes a node of the same graph)."""
        self.model = model
        self.static_in = [a.clone() if torch.is_tensor(a) else a for a in example_args]
        for a in self.static_in:
            if torch.is_tensor(a) and not a.is_cuda:
                raise RuntimeError("GraphedForward needs every tensor argument on the GPU (text_lens too)")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                 # builds weight caches / workspaces / function attributes
                self.static_out = model(*self.static_in)
                if post is not None:
                    self.static_out = post(self.static_out)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = model(*self.static_in)
            if post is not None:
                self.static_out = post(self.static_out)

    def copy_inputs(self, *args):
        for dst, src in zip(self.static_in, args):
            if torch.is_tensor(dst):
                dst.copy_(src, non_blocking=True)

    def replay(self):
        """Run the captured forward on the static inputs; returns the static logits tensor."""
        self.graph.replay()
        return self.static_out

    def __call__(self, *args):
        self.copy_inputs(*args)
        return self.replay()

"""Batch-sharded forward across the GPUs of one node: one process per GPU, replicated weights,
independent samples per rank, ONE collective -- an all-gather of the [B_local, num_labels] logits
(RCCL over xGMI when the backend is "nccl").  The reference is single-GPU (hard-coded cuda:0,
Multi_GCN_Multihead_att.py:85,465,493); its forward has no cross-sample reduction in eval, so this
is exact: gathered logits equal the single-process logits of the concatenated batch.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world, rank):
    """Contiguous near-equal split of n samples: [lo, hi) of `rank`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ShardedForward:
    """Wraps a forward callable; gathers logits of all ranks in rank order."""

    def __init__(self, forward_fn, group=None, comm=None):
        """comm: an mgnns_amd.comm.AbiComm -- the all-gather then goes through the C ABI (mgnns_allgather_logits) instead
        of torch.distributed; same RCCL underneath, same result."""
        self.forward_fn = forward_fn
        self.group = group
        self.comm = comm
        self.world = comm.world if comm is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self._out = None

    def __call__(self, *local_args):
        return self.gather(self.forward_fn(*local_args))

    def gather(self, logits):
        """The collective alone (also usable as GraphedForward's `post` hook: RCCL collectives capture into the
        forward's hipGraph, so a step is one graph launch including the all-gather)."""
        if self.comm is not None:
            return self.comm.all_gather(logits)
        if self.world == 1 and not dist.is_initialized():
            return logits
        shape = (self.world * logits.shape[0],) + tuple(logits.shape[1:])
        if self._out is None or self._out.shape != shape or self._out.device != logits.device:
            self._out = torch.empty(shape, dtype=logits.dtype, device=logits.device)
        if logits.is_cuda and dist.get_backend(self.group) == "gloo":
            # gloo stages device tensors through the host anyway; joining the device FIRST keeps its copy streams from waiting on
            # a hipGraph replay still in flight.  Without it, two processes sharing one GPU (the one-GPU test hooks) degrade to
            # 20-230 ms per step -- reproduced with twelve trivial kernels and nothing of this library
            # (tools/dev/two_proc_gloo.py, NOTES_r05 section 6); with RCCL (one process per GPU) this branch is never taken
            torch.cuda.current_stream(logits.device).synchronize()
        dist.all_gather_into_tensor(self._out, logits.contiguous(), group=self.group)
        return self._out


def gather_variable(logits, group=None):
    """All-gather for UNEQUAL local batch sizes (last shard shorter): pads to the max, trims after."""
    world = dist.get_world_size(group)
    n = torch.tensor([logits.shape[0]], dtype=torch.int64, device=logits.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(logits.shape[1:]), dtype=logits.dtype, device=logits.device)
    pad[:logits.shape[0]] = logits
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0)

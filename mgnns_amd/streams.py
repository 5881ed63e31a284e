"""HIP streams that really run concurrently.

A HIP stream is bound to one of a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) when it is created, and a
hardware queue executes its packets IN ORDER: two streams that share a queue serialise, whatever their dependencies
say (measured: the text->place and place->text fusion stacks ran back to back because their streams shared a queue,
~200 us of a 1 ms forward).  The runtime does not expose the binding, so it is measured once per process and device:
a one-thread spin kernel on stream X and a stamp kernel on stream Y; the stamp landing before the spin ends proves
X and Y sit on different queues.  `independent_streams` returns side streams that are pairwise independent and
independent of the caller's current stream.
"""
import torch

from . import _lib

_cache = {}


def _concurrent(a, b, slots, spin_us=150):
    """True iff a kernel on stream b can run while stream a is busy."""
    L = _lib.lib()
    slots.zero_()
    torch.cuda.synchronize()
    _lib.check(L.mgnns_debug_spin(spin_us, slots.data_ptr(), 0, a.cuda_stream), "mgnns_debug_spin")
    _lib.check(L.mgnns_debug_stamp(slots.data_ptr(), 1, b.cuda_stream), "mgnns_debug_stamp")
    torch.cuda.synchronize()
    end_a, at_b = slots.cpu().tolist()[:2]
    return at_b < end_a


def independent_streams(device, n=3, candidates=16):
    """n side streams, pairwise on different hardware queues and different from the current stream's queue.  Falls back
    to fewer DISTINCT queues when the device has fewer (the remaining streams then share, which is only slower)."""
    device = torch.device(device)
    main = torch.cuda.current_stream(device)
    key = (str(device), main.cuda_stream, n)
    if key in _cache:
        return _cache[key]
    slots = torch.zeros(8, dtype=torch.int64, device=device)
    chosen = []
    pool = [torch.cuda.Stream(device=device) for _ in range(candidates)]
    with torch.cuda.device(device):
        # a stream is bound to its hardware queue (and the queue created) at its FIRST launch, which takes 0.1-5 ms:
        # touch every candidate before timing anything
        for c in [main] + pool:
            _lib.check(_lib.lib().mgnns_debug_stamp(slots.data_ptr(), 2, c.cuda_stream), "mgnns_debug_stamp")
        torch.cuda.synchronize(device)
        for c in pool:
            if len(chosen) == n:
                break
            if all(_concurrent(x, c, slots) and _concurrent(c, x, slots) for x in [main] + chosen):
                chosen.append(c)
    distinct = len(chosen)
    for c in pool:                       # not enough hardware queues: fill up with whatever is left
        if len(chosen) == n:
            break
        if all(c.cuda_stream != x.cuda_stream for x in chosen):
            chosen.append(c)
    _cache[key] = (chosen, distinct)
    return _cache[key]


"""The forward's one collective through the C ABI (include/mgnns_hip.h, section e): an RCCL communicator per rank
created from a 128-byte id that travels over an existing host channel, and `all_gather` of the local logits on the
caller's current stream.  torch.distributed is only the rendezvous (any backend, gloo included); the data path is
libmgnns_hip.so -> RCCL -> xGMI.
"""
import ctypes

import torch

from . import _lib

ID_BYTES = 128


class AbiComm:
    """One rank's communicator.  AbiComm.from_torch_distributed() for the one-process-per-GPU launch; AbiComm(world=1)
    works without any process group."""

    def __init__(self, world=1, rank=0, unique_id=None, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        L = _lib.lib()
        if unique_id is None:
            if world != 1:
                raise ValueError("world > 1 needs the unique id rank 0 made (AbiComm.make_unique_id)")
            unique_id = self.make_unique_id()
        if len(unique_id) != ID_BYTES:
            raise ValueError("unique id must be %d bytes" % ID_BYTES)
        buf = ctypes.create_string_buffer(bytes(unique_id), ID_BYTES)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.mgnns_comm_init_rank(world, rank, ctypes.addressof(buf), ID_BYTES, ctypes.byref(handle)),
                       "mgnns_comm_init_rank")
        self._h = handle
        w, r = ctypes.c_int(), ctypes.c_int()
        _lib.check(L.mgnns_comm_info(self._h, ctypes.byref(w), ctypes.byref(r)), "mgnns_comm_info")
        self.world, self.rank = w.value, r.value
        self._out = None

    @staticmethod
    def make_unique_id():
        buf = ctypes.create_string_buffer(ID_BYTES)
        _lib.check(_lib.lib().mgnns_comm_unique_id(ctypes.addressof(buf), ID_BYTES), "mgnns_comm_unique_id")
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, group=None, device=None):
        """Rank 0 makes the id, torch.distributed's object broadcast (host side) ships it, every rank joins."""
        import torch.distributed as dist
        if not dist.is_initialized():
            return cls(1, 0, None, device)
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [cls.make_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(world, rank, box[0], device)

    def all_gather(self, logits, out=None):
        """[rows_local, NL] fp32 on this rank -> [world * rows_local, NL] in rank order, enqueued on the current stream."""
        if logits.dtype != torch.float32 or logits.dim() != 2 or not logits.is_cuda:
            raise ValueError("all_gather takes a [rows, labels] fp32 CUDA tensor")
        x = logits.contiguous()
        shape = (self.world * x.shape[0], x.shape[1])
        if out is None:
            if self._out is None or tuple(self._out.shape) != shape or self._out.device != x.device:
                self._out = torch.empty(shape, dtype=torch.float32, device=x.device)
            out = self._out
        elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous fp32 %s tensor" % (shape,))
        _lib.check(_lib.lib().mgnns_allgather_logits(self._h, x.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(),
                                                     torch.cuda.current_stream(x.device).cuda_stream), "mgnns_allgather_logits")
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.check(_lib.lib().mgnns_comm_destroy(self._h), "mgnns_comm_destroy")
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

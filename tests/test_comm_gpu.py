"""The C-ABI collective (include/mgnns_hip.h section e) on the one GPU a test box has: a world-1 communicator through
mgnns_comm_unique_id / mgnns_comm_init_rank, the all-gather on a side stream and captured into a hipGraph, the
single-process form mgnns_comm_init_all, and ShardedForward routed through it.  (World > 1 needs one GPU per rank: RCCL
refuses two ranks on a device; the N-rank protocol is covered on CPU by tests/test_sharded_cpu.py and
tests/test_bench_launch_cpu.py.)"""
import ctypes

import pytest
import torch

from mgnns_amd import _lib
from mgnns_amd.comm import AbiComm, ID_BYTES
from mgnns_amd.sharded import ShardedForward

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_world1_allgather_on_a_stream_and_in_a_graph():
    torch.cuda.set_device(0)
    uid = AbiComm.make_unique_id()
    assert len(uid) == ID_BYTES and uid != AbiComm.make_unique_id()
    comm = AbiComm(1, 0, uid)
    assert (comm.world, comm.rank) == (1, 0)
    x = torch.randn(256, 3, device=DEV)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y = comm.all_gather(x)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    assert torch.equal(y, x)
    out = torch.empty(256, 3, device=DEV)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        comm.all_gather(x, out=out)
    for it in range(3):
        x.fill_(float(it))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, x)
    sf = ShardedForward(lambda t: t * 2.0, comm=comm)
    assert torch.equal(sf(x), x * 2.0)
    assert tuple(comm.all_gather(x[:0]).shape) == (0, 3)
    with pytest.raises(ValueError):
        comm.all_gather(x.double())
    comm.close()
    comm.close()          # idempotent


def test_init_all_single_process_form_and_argument_errors():
    L = _lib.lib()
    h = (ctypes.c_void_p * 1)()
    _lib.check(L.mgnns_comm_init_all(1, None, h), "mgnns_comm_init_all")
    w, r = ctypes.c_int(), ctypes.c_int()
    _lib.check(L.mgnns_comm_info(h[0], ctypes.byref(w), ctypes.byref(r)), "mgnns_comm_info")
    assert (w.value, r.value) == (1, 0)
    x = torch.arange(12, device=DEV, dtype=torch.float32).view(4, 3)
    y = torch.empty_like(x)
    _lib.check(L.mgnns_comm_group_start(), "mgnns_comm_group_start")          # the single-thread multi-device form
    _lib.check(L.mgnns_allgather_logits(h[0], x.data_ptr(), 4, 3, y.data_ptr(), torch.cuda.current_stream().cuda_stream),
               "mgnns_allgather_logits")
    _lib.check(L.mgnns_comm_group_end(), "mgnns_comm_group_end")
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    assert L.mgnns_allgather_logits(None, x.data_ptr(), 4, 3, y.data_ptr(), None) == -1
    assert b"null communicator" in L.mgnns_last_error()
    assert L.mgnns_comm_init_all(torch.cuda.device_count() + 1, None, h) != 0          # more devices than visible
    buf = ctypes.create_string_buffer(8)
    assert L.mgnns_comm_unique_id(ctypes.addressof(buf), 8) == -1                       # short id buffer
    _lib.check(L.mgnns_comm_destroy(h[0]), "mgnns_comm_destroy")
    with pytest.raises(ValueError):
        AbiComm(2, 0, None)

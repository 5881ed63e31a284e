import torch, sys
sys.path.insert(0, ".")
from mgnns_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
main = torch.cuda.current_stream()
pool = [torch.cuda.Stream() for _ in range(10)]
slots = torch.zeros(8, dtype=torch.int64, device=dev)
for c in [main] + pool:
    L.mgnns_debug_stamp(slots.data_ptr(), 3, c.cuda_stream)
torch.cuda.synchronize()
def test(a, b, us=150):
    slots.zero_(); torch.cuda.synchronize()
    L.mgnns_debug_stamp(slots.data_ptr(), 2, a.cuda_stream)
    L.mgnns_debug_spin(us, slots.data_ptr(), 0, a.cuda_stream)
    L.mgnns_debug_stamp(slots.data_ptr(), 1, b.cuda_stream)
    torch.cuda.synchronize()
    v = slots.cpu().tolist()
    return (v[0]-v[2])/100.0, (v[1]-v[2])/100.0
print("main->pool", [test(main, p) for p in pool[:6]])
print("pool->main", [test(p, main) for p in pool[:6]])
print("pool0->pool", [test(pool[0], p) for p in pool[1:]])
print("pool1->pool", [test(pool[1], p) for p in pool[2:]])

from mgnns_amd.streams import independent_streams
ch, d = independent_streams("cuda:0", 3)
print("independent streams:", [pool.index(c) if c in pool else hex(c.cuda_stream) for c in ch], "distinct queues:", d)

"""Import shims for running the UNMODIFIED reference classes on CPU -- build container only.

TEST INFRASTRUCTURE.  Used by oracle/gen_goldens.py to produce tests/golden/*.npz.
Nothing in here (and nothing under /root/reference) travels to the GPU box; the
committed fixtures do.

The reference cannot be imported as shipped; the shims below
only repair the *import* and the hard-coded device, never the arithmetic:

  1. cwd with data/glove/tumblr_label_glove.pkl -> the shipped pickle (MODEL:20 path bug)
  2. numpy.int alias (utils/util.py:397)
  3. stub modules: torchvision(.models), word2vec, dgl (documented-semantics stand-in)
  4. models.multi_head_att.submodules -> models.submodules (moudles.py:4-5)
  5. torch.device('cuda:0') -> cpu inside the model module; Tensor.cuda() -> identity
  6. 3-argument gen_A call (MODEL:338,344) -> gen_A(..., gama=0.2)

The dgl stand-in implements exactly the calls Text_GCN.py makes, with DGL's
documented semantics: update_all(src_mul_edge('h','w',m), max(m,'h')) computes
h'_v = max over in-edges (h_src * w_edge), nodes with no in-edge get 0; dgl.batch
concatenates graphs; dgl.sum_nodes is a per-graph sum.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"


# ---------------------------------------------------------------------------
# dgl stand-in
# ---------------------------------------------------------------------------
class _FakeGraph:
    def __init__(self):
        self.n = 0
        self.src = []
        self.dst = []
        self.ndata = {}
        self.edata = {}
        self.sizes = None

    def to(self, device):
        return self

    def add_nodes(self, n):
        self.n += int(n)

    def add_edges(self, srcs, dsts):
        self.src.extend(int(s) for s in srcs)
        self.dst.extend(int(d) for d in dsts)

    def update_all(self, message_func, reduce_func):
        kind, sf, ef, mf = message_func
        rkind, rmf, of = reduce_func
        assert kind == "src_mul_edge" and rkind == "max" and rmf == mf
        h, w = self.ndata[sf], self.edata[ef]
        src = torch.tensor(self.src, dtype=torch.long)
        dst = torch.tensor(self.dst, dtype=torch.long)
        msg = h[src] * w                                  # [E,D] * [E,1]
        out = torch.full((self.n, h.shape[1]), float("-inf"), dtype=h.dtype)
        out = out.scatter_reduce(0, dst[:, None].expand_as(msg), msg, reduce="amax", include_self=True)
        out = torch.where(torch.isinf(out), torch.zeros_like(out), out)   # zero in-degree -> 0
        self.ndata[of] = out


def _fake_batch(graphs):
    g = _FakeGraph()
    off = 0
    for s in graphs:
        g.src.extend(x + off for x in s.src)
        g.dst.extend(x + off for x in s.dst)
        off += s.n
    g.n = off
    g.sizes = [s.n for s in graphs]
    for k in graphs[0].ndata:
        g.ndata[k] = torch.cat([s.ndata[k] for s in graphs], 0)
    for k in graphs[0].edata:
        g.edata[k] = torch.cat([s.edata[k] for s in graphs], 0)
    return g


def _fake_sum_nodes(g, feat):
    return torch.stack([c.sum(0) for c in torch.split(g.ndata[feat], g.sizes, 0)])


def _make_fake_dgl():
    dgl = types.ModuleType("dgl")
    dgl.DGLGraph = _FakeGraph
    dgl.batch = _fake_batch
    dgl.sum_nodes = _fake_sum_nodes
    fn = types.ModuleType("dgl.function")
    fn.src_mul_edge = lambda s, e, out: ("src_mul_edge", s, e, out)
    fn.max = lambda m, out: ("max", m, out)
    dgl.function = fn
    return dgl, fn


class _FakeW2V(dict):
    """word2vec.load() result: model[word] -> 300-d vector (only used at __init__)."""

    def __missing__(self, key):
        return np.zeros(300, dtype=np.float32)


class _TorchProxy:
    """`torch` as seen by the reference model module: device('cuda:0') -> cpu."""

    def __init__(self, real):
        self._real = real

    def device(self, *a, **k):
        return self._real.device("cpu")

    def __getattr__(self, name):
        return getattr(self._real, name)


_INSTALLED = {}


def install():
    """Make `import models.Multi_GCN_Multihead_att` etc. work; returns a namespace."""
    if _INSTALLED:
        return _INSTALLED["ns"]
    if not os.path.isdir(REF):
        raise RuntimeError("reference not present; goldens can only be generated in the build container")
    tmp = tempfile.mkdtemp(prefix="mgnns_ref_cwd_")
    os.makedirs(os.path.join(tmp, "data", "glove"))
    os.symlink(os.path.join(REF, "data", "tumblr_label_glove.pkl"),
               os.path.join(tmp, "data", "glove", "tumblr_label_glove.pkl"))
    os.chdir(tmp)
    if not hasattr(np, "int"):
        np.int = int
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tv.models = tvm
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tvm)
    w2v = types.ModuleType("word2vec")
    w2v.load = lambda path: _FakeW2V()
    sys.modules["word2vec"] = w2v
    dgl, fn = _make_fake_dgl()
    sys.modules["dgl"] = dgl
    sys.modules["dgl.function"] = fn
    sys.path.insert(0, REF)
    import models.submodules as SUBM
    pkg = types.ModuleType("models.multi_head_att")
    pkg.submodules = SUBM
    sys.modules["models.multi_head_att"] = pkg
    sys.modules["models.multi_head_att.submodules"] = SUBM
    torch.Tensor.cuda = lambda self, *a, **k: self
    import models.moudles as MOUD
    import models.Text_GCN as TGCN
    import models.Multi_GCN_Multihead_att as MODEL
    import utils.util as UTIL
    MODEL.torch = _TorchProxy(torch)
    MODEL.gen_A = lambda n, t, f: UTIL.gen_A(n, t, f, 0.2)
    ns = types.SimpleNamespace(SUBM=SUBM, MOUD=MOUD, TGCN=TGCN, MODEL=MODEL, UTIL=UTIL, tmp=tmp)
    _INSTALLED["ns"] = ns
    return ns


class PresetTrunk:
    """Stand-in for a torchvision ResNet: conv1 returns whatever is fed (the post-trunk
    map is passed straight in), every other stage is identity (MODEL:274-294)."""

    def __init__(self):
        ident = torch.nn.Identity
        self.conv1 = ident()
        self.bn1 = ident()
        self.relu = ident()
        self.maxpool = ident()
        self.layer1 = ident()
        self.layer2 = ident()
        self.layer3 = ident()
        self.layer4 = ident()

"""Build libmgnns_hip.so (all HIP kernels + the C ABI) for gfx950, in-tree.

    python -m mgnns_amd.build          # -> mgnns_amd/libmgnns_hip.so

hipcc cross-compiles without a GPU.  The library links only against the HIP runtime; at load
time it binds to the libamdhip64 that PyTorch already loaded (same soname), so kernels and
torch tensors share one runtime, one set of streams.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmgnns_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("MGNNS_HIPCC_FLAGS", "").split()      # e.g. -DMG_MHA_TRACE for the in-kernel phase timer
# per-file additions.  sq_mha_bf16: the SLP vectoriser packs the epilogues' fp32 FMAs into v_pk_fma_f32, which issue no
# faster next to a partner wave's MFMAs and cost the kernel 3 % (60.5 -> 58.6 us)
FILE_FLAGS = {"sq_mha_bf16.hip": ["-fno-slp-vectorize"], "sq_mha32_bf16.hip": ["-fno-slp-vectorize"],
              "sq_mha_split_bf16.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_fingerprint():
    """sha256 (16 hex digits) over every source the library is built from (csrc/*.hip, csrc/*.hpp, include/*.h; names + bytes).
    Compiled into the library as mgnns_source_fingerprint(): a profile or a bench line can say which sources produced it, and
    _lib.check_sources() refuses a library older than the tree it sits in."""
    import hashlib
    h = hashlib.sha256()
    files = sources() + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + sorted(glob.glob(os.path.join(HERE, "..", "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    hdrs = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
        if verbose and out.strip():
            print(out)
    # the fingerprint of the sources, as one more (generated) translation unit
    fp, fp_src, fp_obj = source_fingerprint(), os.path.join(objdir, "fingerprint.cpp"), os.path.join(objdir, "fingerprint_gen.o")
    text = 'extern "C" const char* mgnns_source_fingerprint(void) { return "%s"; }\n' % fp
    if not os.path.exists(fp_src) or open(fp_src).read() != text or not os.path.exists(fp_obj):
        with open(fp_src, "w") as f:
            f.write(text)
        r = subprocess.run([hipcc, "-O1", "-fPIC", "-x", "c++", "-c", fp_src, "-o", fp_obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on the fingerprint unit:\n" + r.stdout)
        procs.append((fp_src, None))
    objs.append(fp_obj)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout)
    return LIB


if __name__ == "__main__":
    if "--fingerprint" in sys.argv:
        print(source_fingerprint())
    else:
        print(build(force="--force" in sys.argv, verbose=True))

#!/usr/bin/env python
"""Instrumented build of ONE source file linked with the regular objects: python tools/dev/build_variant.py <name> <file.hip> <flags...>
-> mgnns_amd/variants/lib_<name>.so (use with MGNNS_LIB=...)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
out = os.path.join(ROOT, "mgnns_amd", "variants")
os.makedirs(out, exist_ok=True)
base = os.path.basename(src)[:-4]
obj = "/tmp/%s_%s.o" % (base, name)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function"] + flags +
                      ["-c", os.path.join(ROOT, "mgnns_amd", "csrc", src), "-o", obj])
objs = [o for o in glob.glob(os.path.join(ROOT, "mgnns_amd", "csrc", "build", "*.o")) if os.path.basename(o) != base + ".o"] + [obj]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(out, "lib_%s.so" % name)] + objs)
print(os.path.join(out, "lib_%s.so" % name))

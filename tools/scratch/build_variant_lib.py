"""Build a libuc_hip variant that differs from the tree's build only by extra compiler flags on some sources (same-box A/B runs:
UC_HIP_LIB=tools/_libs/libuc_<name>.so).  usage: python tools/build_variant_lib.py <name> "<flags>" src1.hip [src2.hip ...]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uniception_amd import build as B
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name, flags, srcs = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
out = os.path.join(root, "tools", "_libs"); os.makedirs(out, exist_ok=True)
B.build(verbose=False)
tmp = f"/tmp/ucvar_{name}"; os.makedirs(tmp, exist_ok=True)
objs = []
procs = []
for s in B.SOURCES:
    if s in srcs:
        o = f"{tmp}/{s[:-4]}.o"
        procs.append(subprocess.Popen([B.hipcc_path()] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, s), "-o", o]))
        objs.append(o)
    else:
        objs.append(os.path.join(B.OBJ, s.replace(".hip", ".o")))
assert all(p.wait() == 0 for p in procs)
lib = os.path.join(out, f"libuc_{name}.so")
subprocess.check_call([B.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", lib])
print(lib)

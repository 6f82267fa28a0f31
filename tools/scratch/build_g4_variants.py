"""Build libuc_hip variants that differ only in the generated four-wave K-loop (csrc/gen/gen_glds4_loop.py experiment switches).
usage: python tools/build_g4_variants.py name:ENV=VAL,ENV=VAL ...   -> tools/_libs/libuc_<name>.so (travels to the GPU box; git-ignored)"""
import os, shutil, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uniception_amd import build as B
from concurrent.futures import ThreadPoolExecutor
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(root, "tools", "_libs"); os.makedirs(out, exist_ok=True)
B.build(verbose=False)
DENSE = ["gemm_glds_dense_bf16.hip", "gemm_glds_dense_f32.hip", "gemm_glds_dense_bs.hip", "gemm_glds_dense_all.hip"]


def one(spec):
    name, _, envs = spec.partition(":")
    env = dict(os.environ); env.update(dict(e.split("=") for e in envs.split(",") if e))
    tmp = f"/tmp/g4_{name}"; shutil.rmtree(tmp, ignore_errors=True); shutil.copytree(os.path.join(root, "uniception_amd", "csrc"), tmp + "/pkg/csrc")
    os.makedirs(tmp + "/include"); shutil.copy(os.path.join(root, "include", "uc_hip.h"), tmp + "/include/uc_hip.h")
    inc = subprocess.check_output([sys.executable, tmp + "/pkg/csrc/gen/gen_glds4_loop.py"], env=env)
    open(tmp + "/pkg/csrc/gemm_glds4_loop.inc", "wb").write(inc)
    objs = []
    for s in B.SOURCES:
        if s in DENSE:
            o = f"{tmp}/{s[:-4]}.o"
            subprocess.check_call([B.hipcc_path()] + B.FLAGS + ["-c", f"{tmp}/pkg/csrc/{s}", "-o", o])
            objs.append(o)
        else:
            objs.append(os.path.join(B.OBJ, s.replace(".hip", ".o")))
    lib = os.path.join(out, f"libuc_{name}.so")
    subprocess.check_call([B.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", lib])
    return lib


with ThreadPoolExecutor(max_workers=4) as ex:
    for lib in ex.map(one, sys.argv[1:]):
        print(lib)

"""Build libuc_hip.so (the C-ABI kernel library) for gfx950 with hipcc, in-tree.

Usage:  python -m uniception_amd.build [--force]
The .so is written next to this file so that it travels with a snapshot of the repository.
"""
import os
import subprocess
import sys
import hashlib

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libuc_hip.so")
SOURCES = ["error.hip", "rope_norm.hip", "gemm.hip", "gemm_glds.hip", "gemm_tn.hip", "attention.hip", "attention_fp8.hip", "attention_bwd.hip", "elementwise.hip", "train.hip", "dpt_bwd.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-Wno-unused-result"]


def _fingerprint():
    h = hashlib.sha256()
    names = sorted(os.listdir(CSRC)) + [os.path.join("..", "..", "include", "uc_hip.h")]
    for n in names:
        p = os.path.join(CSRC, n)
        if os.path.isfile(p):
            with open(p, "rb") as f:
                h.update(n.encode())
                h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def hipcc_path():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build(force=False, verbose=True):
    stamp = LIB + ".stamp"
    fp = _fingerprint()
    if not force and os.path.exists(LIB) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == fp:
                return LIB
    cmd = [hipcc_path()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print("[uniception_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(fp)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

"""Build libuc_hip.so (the C-ABI kernel library) for gfx950 with hipcc, in-tree.

Usage:  python -m uniception_amd.build [--force]
The .so is written next to this file so that it travels with a snapshot of the repository.  Every source is compiled to
its own object (in parallel, re-done only when that source or a header changed) and the objects are linked once.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libuc_hip.so")
SOURCES = ["error.hip", "rope_norm.hip", "gemm.hip", "gemm_glds.hip", "gemm_glds_dense_bf16.hip", "gemm_glds_dense_f32.hip", "gemm_glds_dense_bs.hip", "gemm_glds_dense_all.hip", "gemm_glds_conv.hip", "gemm_glds_conv_f16.hip", "gemm_glds_dense_all_f16.hip", "gemm_tn.hip", "attention.hip", "attention_x3.hip", "attention_fp8.hip",
           "attention_bwd.hip", "elementwise.hip", "train.hip", "dpt_bwd.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result", "-Wno-inline-asm"]


def _read(path):
    with open(path, "rb") as f:
        return f.read()


def _headers_digest():
    h = hashlib.sha256()
    for n in sorted(os.listdir(CSRC)):
        if n.endswith((".h", ".inc")):
            h.update(n.encode())
            h.update(_read(os.path.join(CSRC, n)))
    h.update(_read(os.path.join(HERE, "..", "include", "uc_hip.h")))
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _fingerprint():
    """Digest of everything the library is built from (sources, headers, flags): identifies a build of the kernels."""
    h = hashlib.sha256(_headers_digest().encode())
    for n in SOURCES:
        h.update(n.encode())
        h.update(_read(os.path.join(CSRC, n)))
    return h.hexdigest()


def loaded_fingerprint():
    """Fingerprint recorded next to libuc_hip.so when it was linked (None when the stamp is missing)."""
    try:
        with open(LIB + ".stamp") as f:
            return f.read().strip()
    except OSError:
        return None


def hipcc_path():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _compile_one(src, hdr, force, verbose, extra=(), objdir=None):
    obj = os.path.join(objdir or OBJ, src.replace(".hip", ".o"))
    stamp = obj + ".stamp"
    fp = hashlib.sha256(hdr.encode() + _read(os.path.join(CSRC, src))).hexdigest()
    if not force and os.path.exists(obj) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == fp:
                return obj, False
    cmd = [hipcc_path()] + FLAGS + list(extra) + ["-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print("[uniception_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(fp)
    return obj, True


def build(force=False, verbose=True):
    stamp = LIB + ".stamp"
    fp = _fingerprint()
    if not force and os.path.exists(LIB) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == fp:
                return LIB
    os.makedirs(OBJ, exist_ok=True)
    hdr = _headers_digest()
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = [o for o, _ in ex.map(lambda s: _compile_one(s, hdr, force, verbose), SOURCES)]
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", LIB]
    if verbose:
        print("[uniception_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    check_glds4_agprs(fp, verbose)
    with open(stamp, "w") as f:
        f.write(fp)
    return LIB


def check_glds4_agprs(fp=None, verbose=True):
    """Proofs on the code THIS compiler generated (uniception_amd/check_kernels.py), once per build fingerprint; a violation fails the
    build — the library is not stamped and never loads silently wrong:
    * gemm_bf16_glds4_kernel hands 256 accumulators from its asm K-loop to the C++ epilogues in the PHYSICAL registers a0..a255: sound
      only while the compiler itself touches no AGPR in that kernel.  UC_GEMM_4WAVE=0 in the environment (the knob that keeps the
      kernel from ever being launched) skips this proof, so that a compiler that fails it still gets a usable library;
    * attn_bf16_p64_kernel places every MFMA through inline asm and guarantees the wait states to the readers IT places: it must
      compile without scratch, register spills or AGPRs; the 64-row attention backward kernels (attention_bwd64.h) likewise without
      scratch or spills."""
    fp = fp or _fingerprint()
    mark = LIB + ".agpr"
    skip4 = os.environ.get("UC_GEMM_4WAVE", "") == "0"
    try:
        with open(mark) as f:
            if f.read().strip() == fp and not skip4:
                return
    except OSError:
        pass
    from . import check_kernels as chk
    n = 0
    if skip4:
        if verbose:
            print("[uniception_amd.build] UC_GEMM_4WAVE=0: the four-wave GEMM is off, its AGPR hand-over proof is skipped (this library is "
                  "NOT marked as checked: unset the variable and rebuild to use the kernel)", flush=True)
    else:
        with ThreadPoolExecutor(max_workers=4) as ex:
            reports = list(ex.map(chk.check, chk.TUS))
        for tu, rep in zip(chk.TUS, reports):
            for name, (blocks, bad) in rep.items():
                n += 1
                if bad or blocks < 257:
                    raise RuntimeError(f"[uniception_amd.build] {tu}: {name}: the compiler uses AGPRs outside the asm K-loop ({bad[:3]}, {blocks} asm "
                                       "statements): the four-wave GEMM would be silently wrong with this compiler — rebuild with UC_GEMM_4WAVE=0 "
                                       "in the environment (skips this proof; keep the variable set at run time so the kernel is never launched) and report")
        if n < 4:
            raise RuntimeError(f"[uniception_amd.build] AGPR check found only {n} gemm_bf16_glds4_kernel instantiations (expected 4)")
    p64 = chk.check_p64()
    if len(p64) < 2:
        raise RuntimeError(f"[uniception_amd.build] found {len(p64)} attn_bf16_p64_kernel instantiations in attention.hip (expected 2)")
    for name, r in p64.items():
        if r["scratch"] != 0 or r["vgpr_spills"] != 0 or r["agprs"] != 0:
            raise RuntimeError(f"[uniception_amd.build] {name}: {r}: the persistent attention kernel must compile without scratch, spills or AGPRs "
                               "(a spill of an asm MFMA's result is read before it is written) — set UC_ATTN_P64=0 at run time and report")
    b64 = chk.check_p64("attention_bwd.hip", "attn_bwd_d(kv|q)64_kernel")
    if len(b64) < 2:
        raise RuntimeError(f"[uniception_amd.build] found {len(b64)} 64-row attention backward kernels in attention_bwd.hip (expected 2)")
    for name, r in b64.items():
        if r["scratch"] != 0 or r["vgpr_spills"] != 0:
            raise RuntimeError(f"[uniception_amd.build] {name}: {r}: the 64-row attention backward kernels must compile without scratch or spills "
                               "(a spill of an asm MFMA's result is read before it is written) — set UC_ATTN_BWD64=0 at run time and report")
    if verbose:
        print(f"[uniception_amd.build] generated-code proofs: {n} gemm_bf16_glds4_kernel instantiations clean, {len(p64)} attn_bf16_p64_kernel "
              f"instantiations without scratch / spills / AGPRs, {len(b64)} 64-row attention backward kernels without scratch / spills", flush=True)
    if not skip4:
        with open(mark, "w") as f:
            f.write(fp)


def build_diag(verbose=True):
    """The diagnostics flavour (-DUC_DIAG): libuc_hip_diag.so next to the product library, with the anatomy switches (UC_GEMM_DBG,
    UC_ATTN_DBG: WRONG results by construction) and the per-workgroup timeline (UC_GEMM_TRACE: allocates, synchronises) compiled
    in.  Tools load it with UNICEPTION_AMD_DIAG_LIB=1; it is git-ignored and never the library the package ships or tests."""
    objdir = os.path.join(HERE, "_obj_diag")
    os.makedirs(objdir, exist_ok=True)
    hdr = _headers_digest() + "+diag"
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = [o for o, _ in ex.map(lambda s: _compile_one(s, hdr, False, verbose, ("-DUC_DIAG",), objdir), SOURCES)]
    lib = os.path.join(HERE, "libuc_hip_diag.so")
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", lib]
    if verbose:
        print("[uniception_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    if "--diag" in sys.argv:
        print(build_diag())
    else:
        build(force="--force" in sys.argv)
        print(LIB)

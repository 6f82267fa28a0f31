"""Build-time proofs about the code hipcc generated for the hand-written kernels (run by uniception_amd.build after linking, once per
build fingerprint; tools/check_glds4_agprs.py is the command-line front end).

1. gemm_bf16_glds4_kernel hands its 256 accumulators from the inline-asm K-loop to the C++ epilogue in the PHYSICAL registers
   a0..a255 (hipcc's pinned-tuple asm outputs miscompile, so they cannot be declared as outputs).  Sound only while the compiler
   itself never touches an AGPR in that kernel — it has no reason to (all MFMAs are inside the asm; VGPR pressure stays below the
   spill-to-AGPR point): every instruction of every instantiation that names an AGPR must sit between #ASMSTART / #ASMEND markers.
2. attn_bf16_p64_kernel (attention_p64.h) issues every MFMA through inline asm: hipcc inserts no wait states between such an MFMA and
   a reader of its result, and the kernel guarantees them by construction for the readers IT places — a compiler-generated spill of a
   score register right behind the MFMA that writes it would read garbage.  The kernel must therefore compile without any scratch
   (.private_segment_fixed_size 0, .vgpr_spill_count 0) and without AGPRs (a kernel that uses any gets its budget split 128 / 128).
3. attn_bwd_dkv64_kernel / attn_bwd_dq64_kernel (attention_bwd64.h): the same construction at one wave per SIMD (accumulators and
   stationary operands in AGPRs by design): no scratch, no spills to memory.
"""
import os
import re
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
TUS = ["gemm_glds_dense_bf16.hip", "gemm_glds_dense_bs.hip", "gemm_glds_dense_f32.hip", "gemm_glds_dense_all.hip"]


def _device_asm(tu):
    from . import build
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "dev.s")
        subprocess.run([build.hipcc_path()] + build.FLAGS + ["--cuda-device-only", "-S", os.path.join(CSRC, tu), "-o", out], check=True,
                       stderr=subprocess.DEVNULL)
        return open(out).read()


def check(tu):
    "{kernel name: (asm statements, [compiler-generated instructions that name an AGPR])} for the glds4 instantiations of one translation unit"
    txt = _device_asm(tu)
    report = {}
    for f in re.split(r"\n(?=_Z[0-9A-Za-z_]+:)", txt):
        name = f.split(":", 1)[0]
        if "glds4" not in name:
            continue
        inasm, bad, blocks = False, [], 0
        for ln in f.split("\n"):
            if "#ASMSTART" in ln:
                inasm, blocks = True, blocks + 1
                continue
            if "#ASMEND" in ln:
                inasm = False
                continue
            body = ln.split(";")[0].strip()
            if not inasm and body and not body.startswith(".") and re.search(r"\ba(\[\d+|\d+\b)", body):
                bad.append(body)
        report[name] = (blocks, bad)
    return report


def check_p64(tu="attention.hip", pattern="attn_bf16_p64_kernel"):
    "{kernel name: {scratch bytes, vgpr spills, agprs}} of the kernels whose name contains `pattern` (code-object metadata of the device assembly)"
    txt = _device_asm(tu)
    report = {}
    # the amdhsa.kernels metadata: one YAML map per kernel
    for blk in re.split(r"\n  - (?=\.agpr_count:|\.args:)", txt):
        m = re.search(r"\.name:\s+(\S*" + pattern + r"\S*)", blk)
        if not m:
            continue
        def num(key):
            mm = re.search(r"\." + key + r":\s+(\d+)", blk)
            return int(mm.group(1)) if mm else -1
        report[m.group(1)] = {"scratch": num("private_segment_fixed_size"), "vgpr_spills": num("vgpr_spill_count"), "agprs": num("agpr_count"),
                              "vgprs": num("vgpr_count")}
    return report

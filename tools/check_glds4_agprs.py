"""The four-wave GEMM kernel (gemm_bf16_glds4_kernel) hands its 256 accumulators from the inline-asm K-loop to the C++ epilogue in
the PHYSICAL registers a0..a255 (hipcc's pinned-tuple asm outputs miscompile, so they cannot be declared as outputs).  That is sound
only while the compiler itself never touches an AGPR in that kernel — it has no reason to (all MFMAs are inside the asm; VGPR
pressure stays below the spill-to-AGPR point), and this script PROVES it for the code actually generated: every instruction of
every gemm_bf16_glds4_kernel instantiation that names an AGPR must sit between #ASMSTART / #ASMEND markers.

    python tools/check_glds4_agprs.py [gemm_glds_dense_bf16.hip ...]      exit status 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "uniception_amd", "csrc")
TUS = ["gemm_glds_dense_bf16.hip", "gemm_glds_dense_bs.hip", "gemm_glds_dense_f32.hip", "gemm_glds_dense_all.hip"]


def check(tu):
    from uniception_amd import build
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "dev.s")
        subprocess.run([build.hipcc_path()] + build.FLAGS + ["--cuda-device-only", "-S", os.path.join(CSRC, tu), "-o", out], check=True,
                       stderr=subprocess.DEVNULL)
        txt = open(out).read()
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


def main(argv):
    sys.path.insert(0, ROOT)
    worst = 0
    for tu in (argv or TUS):
        for name, (blocks, bad) in check(tu).items():
            print(f"{tu}: {name}: {blocks} asm statements, {len(bad)} compiler-generated AGPR uses")
            for b in bad[:5]:
                print("    ", b)
            worst |= bool(bad) or blocks < 257
    return 1 if worst else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

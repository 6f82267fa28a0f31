"""Command-line front end of uniception_amd.check_kernels (the proofs build.py runs after linking):

    python tools/check_glds4_agprs.py [gemm_glds_dense_bf16.hip ...]      exit status 1 on a violation
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uniception_amd.check_kernels import TUS, check, check_p64  # noqa: E402,F401


def main(argv):
    worst = 0
    for tu in (argv or TUS):
        for name, (blocks, bad) in check(tu).items():
            print(f"{tu}: {name}: {blocks} asm statements, {len(bad)} compiler-generated AGPR uses")
            for b in bad[:5]:
                print("    ", b)
            worst |= bool(bad) or blocks < 257
    if not argv:
        for name, r in check_p64().items():
            print(f"attention.hip: {name}: {r}")
            worst |= r["scratch"] != 0 or r["vgpr_spills"] != 0 or r["agprs"] != 0
    return 1 if worst else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

"""Where does a kernel spill?  Lists scratch loads/stores of one kernel of a csrc file by position, with landmark instructions
(MFMA loop, sin/cos = RoPE epilogue, row_ror = LN statistics, v_exp = GELU, nt loads = fp32 residual epilogue).
usage: python tools/spills.py gemm_glds.hip '<mangled substring>'"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "uniception_amd", "csrc", sys.argv[1])
os.makedirs("/tmp/st", exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-save-temps", "-c", src, "-o", "/tmp/st/x.o"],
               cwd="/tmp/st", capture_output=True)
s = open("/tmp/st/" + os.path.basename(src).replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s").read()
key = sys.argv[2]
m = re.search(r"^(\S*" + re.escape(key) + r"\S*):", s, flags=re.M)
i = m.start()
lines = s[i:s.index("s_endpgm", i)].split("\n")
marks = {"mfma": "v_mfma", "sincos": "v_sin_f32", "row_ror": "row_ror", "exp": "v_exp_f32", "nt_load": " nt", "ds_write": "ds_write", "spill": "scratch_"}
B = 400
hist = {}
for k, l in enumerate(lines):
    for name, pat in marks.items():
        if pat in l: hist.setdefault(k // B, {}).setdefault(name, 0); hist[k // B][name] += 1
print(m.group(1), len(lines), "lines")
for b in sorted(hist): print(f"{b*B:6d}: " + " ".join(f"{n}={c}" for n, c in sorted(hist[b].items())))

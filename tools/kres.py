"""Per-kernel register / LDS / scratch usage of one csrc file (hipcc -Rpass-analysis=kernel-resource-usage), as a table.
usage: python tools/kres.py gemm_glds.hip [name-filter]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "uniception_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", *os.environ.get("KRES_FLAGS", "").split(), "-c", src, "-o",
                      "/tmp/_kres.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd="/tmp").stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["/usr/bin/c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur); continue
    m = re.search(r"remark: \S+\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line) or re.search(r":\s{2,}([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][:110]:110s} vgpr {r.get('VGPRs',0):3d} agpr {r.get('AGPRs',0):3d} sgpr {r.get('SGPRs',0):3d} "
              f"scratch {r.get('ScratchSize',0):4d} vspill {r.get('VGPRs Spill',0):3d} occ {r.get('Occupancy',0)} lds {r.get('LDS Size',0)}")

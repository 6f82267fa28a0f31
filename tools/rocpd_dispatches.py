"""Per-(kernel, grid) duration summary from a rocprofv3 rocpd database."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
q = f"select name, {gx}, start, end from kernels" if gx else "select name, 0, start, end from kernels"
agg = {}
for name, g, s, e in cur.execute(q):
    k = (re.sub(r"\(.*$", "", name).replace("void ", "")[:70], g)
    a = agg.setdefault(k, []); a.append((e - s) * 1e-3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort(); print(f"{k[0]:72s} grid={k[1]:8d} n={len(v):5d} med={v[len(v)//2]:9.2f}us min={v[0]:9.2f} max={v[-1]:9.2f}")

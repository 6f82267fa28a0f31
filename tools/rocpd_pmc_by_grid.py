"""PMC counters per (kernel, grid size) from a rocprofv3 rocpd database (counter collection run)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
ci = {c: i for i, c in enumerate(cols)}
kn = "kernel_name" if "kernel_name" in ci else [c for c in cols if "name" in c and "kernel" in c][0]
cn = "counter_name" if "counter_name" in ci else [c for c in cols if "counter" in c and "name" in c][0]
vn = "value" if "value" in ci else [c for c in cols if "value" in c][0]
gn = next((c for c in ("grid_size", "grid_size_x", "grid_x") if c in ci), None)
if gn is None:
    print("columns:", cols)
agg = {}
for r in cur.execute("select * from counters_collection"):
    k = (re.sub(r"\(.*$", "", r[ci[kn]]).replace("void ", "")[:64], r[ci[gn]] if gn else 0, r[ci[cn]])
    a = agg.setdefault(k, [0.0, 0]); a[0] += r[ci[vn]]; a[1] += 1
for (k, g, c), (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    if v / max(n, 1) < 1000: continue
    print(f"{k:66s} grid={g:10d} {c:12s} per-dispatch={v/max(n,1):14.1f} KB  (n={n})")

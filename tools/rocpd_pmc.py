"""Sum PMC counters per kernel from a rocprofv3 rocpd database (counter collection run)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
rows = cur.execute("select * from counters_collection").fetchall()
ci = {c: i for i, c in enumerate(cols)}
kn = "kernel_name" if "kernel_name" in ci else [c for c in cols if "name" in c and "kernel" in c][0]
cn = "counter_name" if "counter_name" in ci else [c for c in cols if "counter" in c and "name" in c][0]
vn = "value" if "value" in ci else [c for c in cols if "value" in c][0]
agg = {}
for r in rows:
    k = re.sub(r"\(.*$", "", r[ci[kn]]).replace("void ", "")[:60]
    d = agg.setdefault(k, {})
    a = d.setdefault(r[ci[cn]], [0.0, 0])
    a[0] += r[ci[vn]]; a[1] += 1
for k, d in agg.items():
    if "gemm" not in k and "attn" not in k and "layernorm" not in k and len(sys.argv) < 3: continue
    print(k)
    for c, (v, n) in sorted(d.items()):
        print(f"   {c:32s} total={v:16.0f}  per-dispatch={v/max(n,1):14.1f}  (n={n})")

"""Timeline view of the LAST replay of a step in a rocprofv3 rocpd database (kernel-trace): per kernel family the busy time, the
union of busy intervals (two streams overlap), the idle gaps between dispatches, and the largest kernels of the step.
usage: python tools/rocpd_timeline.py <results.db> <kernels per step (0 = guess from the patch_gather kernel)>"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else "0")
rows = cur.execute(f"select name, start, end, {gx}, {wx} from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(.*$", "", n).replace("void ", "")[:64]
marks = [i for i, r in enumerate(rows) if "patch_gather" in r[0] or "patch_embed" in r[0]]
# one forward = from one patch kernel of view 1 to the next forward's; take the last complete step
firsts = [m for k, m in enumerate(marks) if k == 0 or m - marks[k - 1] > 20]
if len(firsts) < 3: sys.exit("not enough steps in the trace")
a, b = firsts[-2], firsts[-1]
step = rows[a:b]
t0, t1 = step[0][1], max(r[2] for r in step)
busy = sum(r[2] - r[1] for r in step)
iv = sorted((r[1], r[2]) for r in step); union = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: union += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
union += ce - cs
print(f"step: {len(step)} dispatches, wall {(t1-t0)*1e-3:.1f} us, sum of kernel durations {busy*1e-3:.1f} us, union of busy intervals {union*1e-3:.1f} us, idle {(t1-t0-union)*1e-3:.1f} us")
agg = {}
for n, s, e, g, w in step:
    k = short(n); x = agg.setdefault(k, [0, 0.0, 0]); x[0] += 1; x[1] += (e - s) * 1e-3; x[2] = max(x[2], (g // w) if w else 0)
print(f"{'kernel':66s} {'n':>4s} {'total us':>9s} {'avg us':>8s} {'max WGs':>8s}")
for k, x in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:66s} {x[0]:4d} {x[1]:9.1f} {x[1]/x[0]:8.2f} {x[2]:8d}")
if len(sys.argv) > 2 and sys.argv[2] == "list":
    for n, s, e, g, w in step:
        print(f"{(s-t0)*1e-3:9.1f} {(e-s)*1e-3:8.2f} wg={(g//w) if w else 0:6d} {short(n)}")

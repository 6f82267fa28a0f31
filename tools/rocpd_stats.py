"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table (markdown/CSV-ish text).
usage: python tools/rocpd_stats.py gpurun_out/prof/xxx_results.db [> profiles/xxx.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:110]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall() if "name" in cols else []
    agg = {}
    for name, s, e in rows:
        d = (e - s) * 1e-3  # us
        a = agg.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"total kernel time {total/1e3:.3f} ms over {sum(a[0] for a in agg.values())} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k} | {a[0]} | {a[1]/1e3:.3f} | {a[1]/a[0]:.2f} | {a[2]:.2f} | {a[3]:.2f} | {100*a[1]/total:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])

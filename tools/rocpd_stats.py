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
    # the dominant kernel of bench.py's roofline object is the dense bf16 GEMM across its instantiations (16-wave kernels with
    # A_MODE = 0, every epilogue family, and the eight-wave kernel): one aggregate line to compare with roofline.avg_launch_us
    dense = [a for k, a in agg.items() if re.match(r"gemm_bf16_glds_kernel<\d+, \d+, \d+, \d+, \d+, 0,", k) or k.startswith(("gemm_bf16_glds8_kernel<", "gemm_bf16_glds4_kernel<"))]
    if dense:
        n = sum(a[0] for a in dense); t = sum(a[1] for a in dense)
        print(f"\ndense bf16 GEMM (all dense instantiations): {n} calls, {t/1e3:.3f} ms, avg {t/n:.2f} us, {100*t/total:.1f} % of kernel time")


if __name__ == "__main__":
    main(sys.argv[1])

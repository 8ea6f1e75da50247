"""Compact a rocprofv3 `*_kernel_stats.csv` — or the rocpd `*_results.db` newer rocprofv3 writes by default — into a
short markdown table (kernel names truncated).

    python tools/summarize_rocprof.py gpurun_out/prof1/runc/842_kernel_stats.csv [--top 20] > profiles/r01_bench_kernel_stats.md
"""
import csv
import re
import sys


def short(name: str, n: int = 72) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if name.startswith("Cijk_"):
        mt = re.search(r"MT(\d+x\d+x\d+)", name)
        return f"hipBLASLt/Tensile GEMM {name[:22]}.. MT{mt.group(1) if mt else '?'}"
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"<.*", lambda m: m.group(0)[:40], name, count=1)
    return name[:n]


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 20
    if path.endswith(".db"):
        import sqlite3

        con = sqlite3.connect(path)
        rows = [dict(Name=n, Calls=str(c), TotalDurationNs=str(t), AverageNs=str(t / c), MinNs=str(lo), MaxNs=str(hi))
                for n, c, t, lo, hi in con.execute('SELECT name, COUNT(*), SUM("end" - start), MIN("end" - start), '
                                                   'MAX("end" - start) FROM kernels GROUP BY name')]
    else:
        rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"source: {path}\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for r in rows[:top]:
        print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | "
              f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | "
              f"{100 * float(r['TotalDurationNs']) / total:.1f} |")


if __name__ == "__main__":
    main()

"""Where a decode step's time goes: kernels of ONE steady-state step in launch order with their durations and the idle
gaps between them, from a rocprofv3 --kernel-trace results db.

    python tools/step_timeline.py /tmp/prof/run_results.db [--anchor decode_mfma] [--layers 28]

A step is delimited by its `--layers` launches of the anchor kernel; the median step of the trace is printed as a
per-kernel-name table (count, busy us, gap-before us) plus the totals: busy, gaps, span.
"""
import re
import sqlite3
import sys
from collections import OrderedDict


def short(name: str, n: int = 56) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    if name.startswith("Cijk_"):
        mt = re.search(r"MT(\d+x\d+x\d+)", name)
        return f"hipBLASLt GEMM MT{mt.group(1) if mt else '?'}"
    name = re.sub(r"^void ", "", name)
    return name[:n]


def main():
    path = sys.argv[1]
    anchor = sys.argv[sys.argv.index("--anchor") + 1] if "--anchor" in sys.argv else "decode_mfma"
    layers = int(sys.argv[sys.argv.index("--layers") + 1]) if "--layers" in sys.argv else 28
    con = sqlite3.connect(path)
    rows = con.execute('SELECT name, start, "end" FROM kernels ORDER BY start').fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(idx) < 3 * layers:
        raise SystemExit(f"only {len(idx)} launches of {anchor!r}")
    # steps: consecutive groups of `layers` anchor launches; a step spans from the first kernel after the previous
    # step's last anchor-layer tail to the kernel before the next step's first anchor... simpler: cut at the midpoint
    # between the last anchor of step s and the first anchor of step s+1 is wrong (head/tail kernels), so cut where the
    # largest gap between those two anchors is (the host-side step boundary)
    steps = []
    groups = [idx[i:i + layers] for i in range(0, len(idx) - layers + 1, layers)]
    cuts = []
    for g0, g1 in zip(groups, groups[1:]):
        a, b = g0[-1], g1[0]
        best = max(range(a, b), key=lambda i: rows[i + 1][1] - rows[i][2])
        cuts.append(best + 1)
    for c0, c1 in zip(cuts, cuts[1:]):
        steps.append(rows[c0:c1])
    steps = steps[len(steps) // 3:]           # skip warm-up / capture
    steps.sort(key=lambda s: s[-1][2] - s[0][1])
    step = steps[len(steps) // 2]
    per = OrderedDict()
    prev_end = step[0][1]
    busy = gaps = 0
    for name, s, e in step:
        d = per.setdefault(short(name), [0, 0.0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
        d[2] += max(0, s - prev_end) / 1e3
        busy += e - s
        gaps += max(0, s - prev_end)
        prev_end = max(prev_end, e)
    span = (step[-1][2] - step[0][1]) / 1e3
    print(f"source: {path}\nmedian of {len(steps)} steady steps: {len(step)} kernels, span {span:.1f} us, "
          f"busy {busy / 1e3:.1f} us, idle between kernels {gaps / 1e3:.1f} us\n")
    print("| kernel | launches | busy us | idle before (sum) us | avg busy | avg idle before |")
    print("|---|---:|---:|---:|---:|---:|")
    for k, (n, b, g) in per.items():
        print(f"| `{k}` | {n} | {b:.1f} | {g:.1f} | {b / n:.2f} | {g / n:.2f} |")


if __name__ == "__main__":
    main()

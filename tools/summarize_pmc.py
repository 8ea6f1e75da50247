"""Summarise a rocprofv3 --pmc counter_collection CSV: per kernel (name filter), mean of each counter per dispatch.

    python tools/summarize_pmc.py <counter_collection.csv> [kernel-substring]
"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else "decode_kernel"
    acc = defaultdict(lambda: defaultdict(float))   # counter -> dispatch -> value
    for r in csv.DictReader(open(path)):
        if filt not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, d in sorted(acc.items()):
        vals = list(d.values())
        print(f"{c:28s} dispatches={len(vals):3d} mean={sum(vals) / len(vals):.6g}")


if __name__ == "__main__":
    main()

"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel (name filter), mean of each counter per dispatch.

    python tools/summarize_pmc.py <counter_collection.csv> [kernel-substring]
    python tools/summarize_pmc.py --json OUT.json <fetch.csv> <write.csv> <kernel-substring>
        (the decode kernel's HBM bytes per launch for bench.py: FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE exact)
"""
import csv
import json
import sys
from collections import defaultdict


def means(path: str, filt: str) -> dict:
    acc = defaultdict(lambda: defaultdict(float))   # counter -> dispatch -> value
    for r in csv.DictReader(open(path)):
        if filt not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {c: (sum(d.values()) / len(d), len(d)) for c, d in acc.items()}


def main():
    if sys.argv[1] == "--json":
        out, fetch_csv, write_csv, filt = sys.argv[2:6]
        f, nf = means(fetch_csv, filt)["FETCH_SIZE"]
        w, nw = means(write_csv, filt)["WRITE_SIZE"]
        hbm = int(2 * f * 1024 + w * 1024)
        doc = {
            "what": "HBM traffic per launch of mi::decode_mfma_kernel in `python bench.py --steps 2 --warmup 1` "
                    "(B=256, ctx=4096, TP=1, bf16, head-major blocks)",
            "how": f"two separate rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE; they do not fit one pass), mean over "
                   f"the {nf} / {nw} decode dispatches of each run; tools/measure_decode_traffic.sh",
            "units_and_corrections": "FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of "
                                     "wide (16 B/lane) coalesced streaming reads -> doubled (MI355X_MICROARCH.md HBM section); "
                                     "WRITE_SIZE is exact",
            "batch": 256, "ctx": 4096, "tp": 1,
            "FETCH_SIZE_KiB_mean": f, "WRITE_SIZE_KiB_mean": w,
            "algorithmic_bytes_per_launch": 2147483648, "hbm_bytes_per_launch": hbm,
            "traffic_over_algorithmic": round(hbm / 2147483648, 4),
            "source": "round 3 re-measurement after the hi + lo P split of the decode kernel",
        }
        with open(out, "w") as fh:
            json.dump(doc, fh, indent=1)
        print(json.dumps(doc))
        return
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else "decode_kernel"
    for c, (m, n) in sorted(means(path, filt).items()):
        print(f"{c:28s} dispatches={n:3d} mean={m:.6g}")


if __name__ == "__main__":
    main()

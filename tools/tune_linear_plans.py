"""Measures the tile plan of the decode-batch linear kernel for every (tensor-parallel degree, rows per rank) pair the
headline talker's plans produce, and writes vllm_omni_amd/tuned/linear_plans_gfx950.json — the table
`vllm_omni_amd.tuned.linear_plans` registers with the library (mi_linear_plan_set) and `layers.py` routes by.  The
counterpart of tools/tune_gemms.py (TunableOp selections of the library GEMMs the same layers fall back to).

  python tools/tune_linear_plans.py [--out PATH] [--tp 1,2,4] [--rows 32,64,128,256]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_shard_gemm import predictor_shapes, sweep, talker_shapes  # noqa: E402
from vllm_omni_amd import _C  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    here = os.path.dirname(os.path.abspath(__file__))
    ap.add_argument("--out", default=os.path.join(here, "..", "vllm-omni_amd", "vllm_omni_amd", "tuned", "linear_plans_gfx950.json"))
    ap.add_argument("--tp", default="1,2,4")
    ap.add_argument("--rows", default="32,64,128,256")
    ap.add_argument("--predictor", action="store_true",
                    help="the Qwen3-Omni code predictor's four projections at 32..512 rows instead (weights warm in the "
                         "Infinity Cache, as they are across the 31 steps of a call); entries are tagged tp 0")
    ap.add_argument("--only", default="", help="comma-separated projection names: measure just these")
    ap.add_argument("--merge", default=None, help="an existing table whose entries of OTHER shapes are kept")
    a = ap.parse_args()
    from vllm_omni_amd.tuned import enable_tuned_gemms

    print("tuned library selections loaded:", enable_tuned_gemms(), flush=True)
    _C.lib.mi_linear_plan_clear()          # the "built-in" column must be the heuristic, not an earlier table
    g = torch.Generator(device="cuda").manual_seed(0)
    entries = []
    if a.predictor:
        os.environ["COPIES_MB"] = "150"
        for m in (32, 64, 128, 256, 512):
            for name, n, k, gated, bias in predictor_shapes():
                if a.only and name not in a.only.split(","):
                    continue
                e = sweep(name, n, k, gated, bias, m, g, log=lambda s: print(s, flush=True), top=3)
                e["tp"] = 0
                entries.append(e)
    else:
        for tp in (int(t) for t in a.tp.split(",")):
            for m in (int(r) for r in a.rows.split(",")):
                for name, n, k, gated, bias in talker_shapes(tp):
                    if a.only and name not in a.only.split(","):
                        continue
                    e = sweep(name, n, k, gated, bias, m, g, log=lambda s: print(s, flush=True), top=3)
                    e["tp"] = tp
                    entries.append(e)
    extra = {}
    if a.merge and os.path.exists(a.merge):
        with open(a.merge) as f:
            old = json.load(f)
        mine = {(e["gated"], e["n"], e["k"]) for e in entries}
        entries = [e for e in old.get("entries", []) if (e["gated"], e["n"], e["k"]) not in mine] + entries
        extra = {k: v for k, v in old.items() if k in ("scope",)}
    doc = {"device": torch.cuda.get_device_name(0), "arch": "gfx950", "generated_by": "tools/tune_linear_plans.py",
           "note": "us per call replayed from a hipGraph, weights rotating over > 256 MB; library = TunableOp-selected "
                   "hipBLASLt/rocBLAS GEMM (gated: + the SiluAndMul launch); full = mi_linear incl. its reduce launch; "
                   "partials = mi_linear_partials alone (the reduce rides in the consumer)",
           **extra, "entries": entries}
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", a.out, len(entries), "entries")


if __name__ == "__main__":
    main()

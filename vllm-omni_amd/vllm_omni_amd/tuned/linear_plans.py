"""Measured tile plans of the decode-batch linear kernel (csrc/skinny_gemm.hip), made by tools/tune_linear_plans.py on an
MI355X: which (row block, weight rows per tile, split-K) each projection shape runs, and how it compares with the
TunableOp-selected library GEMM at that shape.  ``load()`` registers the plans with the library
(``mi_linear_plan_set``) and keeps the timings for ``layers._apply_linear`` / ``fused_gate_up_silu`` to route by — the
same role ``tunableop_gfx950.csv`` plays for the library GEMMs.  Shapes that are not in the table keep the built-in
heuristic and the round-3 routing rules.  ``VLLM_OMNI_AMD_LINEAR_PLANS=0`` disables the table (A/B runs)."""
from __future__ import annotations

import json
import os
from typing import Optional

_DIR = os.path.dirname(os.path.abspath(__file__))
RESULTS = os.path.join(_DIR, "linear_plans_gfx950.json")
_table: dict = {}      # (gated, n, k) -> [(m_hi, entry), ...] ascending
_loaded = False


def load(path: Optional[str] = None) -> int:
    """Idempotent.  Returns the number of shapes registered (0 when disabled or the file is absent)."""
    global _loaded
    if _loaded:
        return sum(len(v) for v in _table.values())
    _loaded = True
    path = path or RESULTS
    if os.environ.get("VLLM_OMNI_AMD_LINEAR_PLANS", "1") != "1" or not os.path.exists(path):
        return 0
    from .. import _C

    with open(path) as f:
        doc = json.load(f)
    by_shape: dict = {}
    for e in doc.get("entries", []):
        by_shape.setdefault((bool(e["gated"]), int(e["n"]), int(e["k"])), {})[int(e["m"])] = e   # a duplicate shape: last wins
    n = 0
    for key, by_m in by_shape.items():
        gated, nn, kk = key
        lo = 0
        rows = []
        for m in sorted(by_m):
            e = by_m[m]
            full, part = e.get("full"), e.get("partials")
            if full:
                rc = _C.lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_GATED if gated else _C.MI_LINEAR_PLAN_FULL, lo, m, nn, kk,
                                               int(full["rows"]), int(full["wrows"]), int(full["split"]))
                if rc != _C.MI_OK:      # a table from another build: that shape keeps the heuristic
                    lo = m
                    continue
            if part and not gated:
                _C.lib.mi_linear_plan_set(_C.MI_LINEAR_PLAN_PARTIALS, lo, m, nn, kk, int(part["rows"]), int(part["wrows"]),
                                          int(part["split"]))
            rows.append((m, e))
            lo = m
            n += 1
        _table[key] = rows
    return n


def lookup(m: int, n: int, k: int, gated: bool) -> Optional[dict]:
    """The table entry that covers ``m`` rows of the shape (n = output columns; k the contraction), or None."""
    rows = _table.get((gated, n, k))
    if not rows:
        return None
    for m_hi, e in rows:
        if m <= m_hi:
            return e
    return None


def reset() -> None:
    """Forget the table (tests; the tuner)."""
    global _loaded
    _table.clear()
    _loaded = False
    from .. import _C

    _C.lib.mi_linear_plan_clear()

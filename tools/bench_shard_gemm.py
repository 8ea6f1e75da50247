"""Tile-plan sweep of the decode-batch linear kernel at the per-rank shapes of the tensor-parallel plans (qwen2_old.py:89-102
head partition; 47-60,109-124 the four projections).  Every candidate (row block, weight rows per tile, split-K) is checked
against the library GEMM and timed REPLAYED FROM A hipGraph, weights rotating over > 256 MB (no Infinity-Cache hits).

  M=128 TP=4 python tools/bench_shard_gemm.py            # all four projections
  M=128 TP=4 ONLY=down python tools/bench_shard_gemm.py

tools/tune_linear_plans.py runs the same sweep over every (TP, rows) pair and writes the table the product loads.
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vllm-omni_amd"))
from vllm_omni_amd import ops  # noqa: E402


def timed(fn, iters=64, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    n = 16
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    reps = max(2, iters // n)
    best = 1e9
    for _ in range(3):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(reps):
            g.replay()
        ev[1].record()
        torch.cuda.synchronize()
        best = min(best, ev[0].elapsed_time(ev[1]) / (reps * n) * 1e3)
    return best


def talker_shapes(tp: int):
    """(name, weight rows n, k, gated, bias) of one rank's four projections of the headline talker (S-A)."""
    nq, nkv = 28 // tp, max(1, 4 // tp)
    return [("qkv", (nq + 2 * nkv) * 128, 3584, False, True), ("o", 3584, nq * 128, False, False),
            ("gate_up", 2 * 18944 // tp, 3584, True, False), ("down", 3584, 18944 // tp, False, False),
            # once per step: the vocab-parallel lm_head (qwen2_old.py:409-418) and, un-sharded, thinker_to_talker_proj
            # (qwen2_5_omni_talker.py:72-75: embedding_size 3584 -> hidden 3584, bias) — at TP 1 the o_proj shape with a bias
            ("lm_head", 8448 // tp, 3584, False, False)] + ([("t2t_proj", 3584, 3584, False, True)] if tp > 1 else [])


def predictor_shapes():
    """The Qwen3-Omni code predictor's four projections (models/common/qwen3_code_predictor.py:143-191: hidden 1024, 16 / 8
    heads of 128, intermediate 3072, no biases): 157 MB of weights over its five layers, re-read at every one of the 31
    steps of a call — they live in the 256 MB Infinity Cache, so they are timed warm (COPIES_MB=150)."""
    return [("cp_qkv", 4096, 1024, False, False), ("cp_o", 1024, 2048, False, False),
            ("cp_gate_up", 6144, 1024, True, False), ("cp_down", 1024, 3072, False, False),
            # one of the G - 1 heads after every stack pass (qwen3_code_predictor.py:556-560: vocab 2048)
            ("cp_lm_head", 2048, 1024, False, False)]


def candidates(m: int, n_cols: int, k: int, gated: bool):
    """(rows per block / 16, weight rows per tile / 32, split) of every built tile shape that gives 96-640 workgroups."""
    mts = [4] if m <= 64 else [4, 8] if m <= 128 else [4, 8, 16]
    lo_wgs = 32 if n_cols * k < (8 << 20) else 96      # small weights (the code predictor's): fewer workgroups may win
    out = []
    for mt in mts:
        for ns in (1, 2, 3, 4, 5, 6, 8):
            if gated:
                ok = (mt == 4 and ns <= 6) or (mt == 8 and ns <= 5) or (mt == 16 and ns in (1, 2, 4, 5))
                splits = (1,)
            else:
                ok = (mt == 4) or (mt == 8 and ns <= 5) or (mt == 16 and ns in (4, 5))
                splits = tuple(sp for sp in (1, 2, 3, 4, 6, 8) if sp <= k // 128)
            if not ok:
                continue
            for sp in splits:
                rb = -(-m // (16 * mt))
                tiles = -(-n_cols // ((16 if gated else 32) * ns))
                if lo_wgs <= tiles * sp * rb <= 640:
                    out.append((mt, ns, sp))
    return out


def sweep(name, n, k, gated, bias, m, g, log=print, top=8):
    """Times the library, the kernel's built-in plan and every candidate plan.  Returns a dict for the tuned table."""
    copies = max(2, int(float(os.environ.get("COPIES_MB", 600)) * 1e6 // (n * k * 2)))
    ws = [(torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16) for _ in range(copies)]
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    b = (torch.randn(n, device="cuda", generator=g) * 0.1).to(torch.bfloat16) if bias else None
    i = [0]
    gb = n * k * 2 / 1e9

    def nxt():
        i[0] = (i[0] + 1) % copies
        return ws[i[0]]

    for v in ("MI_LINEAR_PLAN", "MI_GATED_PLAN"):
        os.environ.pop(v, None)
    if gated:
        ref = ops.silu_and_mul(F.linear(x, ws[0])).float()
        t_lib = timed(lambda: ops.silu_and_mul(F.linear(x, nxt())))
        t_def = timed(lambda: ops.linear_silu_mul(x, nxt()))
    else:
        ref = F.linear(x, ws[0], b).float()
        t_lib = timed(lambda: F.linear(x, nxt(), b))
        t_def = timed(lambda: ops.linear(x, nxt(), b))
    log(f"== {name} m={m} n={n} k={k} ({gb * 1e3:.1f} MB): library{' pair' if gated else ''} {t_lib:6.1f} us, "
        f"kernel built-in plan {t_def:6.1f} us")
    tol = 0.02 * ref.abs().max().item() + 1e-3
    rows = []
    for mt, ns, sp in candidates(m, n // 2 if gated else n, k, gated):
        os.environ["MI_GATED_PLAN" if gated else "MI_LINEAR_PLAN"] = f"{mt},{ns},{sp}"
        if gated:
            err = (ops.linear_silu_mul(x, ws[0]).float() - ref).abs().max().item()
            t = timed(lambda: ops.linear_silu_mul(x, nxt()))
            tp_ = None
        else:
            err = (ops.linear(x, ws[0], b).float() - ref).abs().max().item()
            t = timed(lambda: ops.linear(x, nxt(), b))
            tp_ = timed(lambda: ops.linear_partials(x, nxt(), b)) if sp > 1 else None
        if not err <= tol:
            log(f"   WRONG rows {16 * mt} w-rows {32 * ns} split {sp}: max|diff| {err:.3g} (tolerance {tol:.3g})")
            continue
        rows.append((t, mt, ns, sp, err, tp_))
    for v in ("MI_LINEAR_PLAN", "MI_GATED_PLAN"):
        os.environ.pop(v, None)
    rows.sort()
    for t, mt, ns, sp, err, tp_ in rows[:top]:
        extra = f"   partials only {tp_:6.1f} us" if tp_ is not None else ""
        log(f"   rows {16 * mt:3d} x{-(-m // (16 * mt))}  w-rows {(32 * ns):3d}  split {sp}: {t:6.1f} us "
            f"({gb / t * 1e3:4.2f} TB/s)  max|diff| {err:.3g}{extra}")
    entry = {"name": name, "m": m, "n": n // 2 if gated else n, "k": k, "gated": gated, "library_us": round(t_lib, 2),
             "builtin_us": round(t_def, 2)}
    if rows:
        t, mt, ns, sp, _, _ = rows[0]
        # the built-in plan stays unless a candidate is ahead by more than the run-to-run noise of these ~10-70 us timings
        import ctypes

        from vllm_omni_amd import _C
        buf = (ctypes.c_int32 * 3)()
        _C.lib.mi_linear_plan_get(_C.MI_LINEAR_PLAN_GATED if gated else _C.MI_LINEAR_PLAN_FULL, m, n // 2 if gated else n, k, buf)
        purpose = _C.MI_LINEAR_PLAN_GATED if gated else _C.MI_LINEAR_PLAN_FULL
        registrable = _C.lib.mi_linear_plan_set(purpose, 0, 1, 16, 64, buf[0], buf[1], buf[2]) == _C.MI_OK   # a throw-away entry
        _C.lib.mi_linear_plan_clear()
        if t_def <= 1.03 * t and registrable:
            entry["full"] = {"rows": buf[0], "wrows": buf[1], "split": buf[2], "us": round(t_def, 2), "builtin": True}
        else:
            entry["full"] = {"rows": 16 * mt, "wrows": 32 * ns, "split": sp, "us": round(t, 2)}
        part = sorted((r[5], r[1], r[2], r[3]) for r in rows if r[5] is not None)
        if part:
            tp_, mt, ns, sp = part[0]
            entry["partials"] = {"rows": 16 * mt, "wrows": 32 * ns, "split": sp, "us": round(tp_, 2)}
    return entry


def main():
    from vllm_omni_amd.tuned import enable_tuned_gemms

    print("tuned library selections loaded:", enable_tuned_gemms())
    m = int(os.environ.get("M", 128))
    tp = int(os.environ.get("TP", 4))
    only = os.environ.get("ONLY", "")
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, n, k, gated, bias in (predictor_shapes() if os.environ.get("PREDICTOR") == "1" else talker_shapes(tp)):
        if only and name not in only.split(","):
            continue
        sweep(name, n, k, gated, bias, m, g, log=lambda s: print(s, flush=True), top=14)


if __name__ == "__main__":
    main()

"""CPU: the parallel-plan logic of bench.py (no GPU, no process group): the widest tensor-parallel plan obeys the
head-partition rules of qwen2_old.py:89-102, labels are stable, the plans a run measures are the TP-scaling line
first and pure data parallelism beside it, and `--gpus N` without a launcher never reports a smaller job as N."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_plans(monkeypatch):
    from vllm_omni_amd.talker import TalkerConfig

    b = _bench()
    cfg = TalkerConfig.from_json("qwen2_5_omni_talker")          # 28 query heads, 4 KV heads
    assert [b.parallel_plan(n, cfg)[:2] for n in (1, 2, 4, 8)] == [(1, 1), (2, 1), (4, 1), (4, 2)]
    assert b.parallel_plan(8, cfg)[2] == "dp2xtp4" and b.plan_label(1, 8) == "dp8" and b.plan_label(2, 1) == "tp2"
    q3 = TalkerConfig.from_json("qwen3_omni_talker_attn")        # 16 query heads, 2 KV heads: KV replicated at tp 4, 8
    assert [b.parallel_plan(n, q3)[0] for n in (2, 4, 8)] == [2, 4, 8]
    for var in ("MI_BENCH_PLAN", "MI_BENCH_ALT"):
        monkeypatch.delenv(var, raising=False)
    # headline = the TP scaling line; the data-parallel alternative is measured beside it
    assert [(t, d, lab) for t, d, lab, _ in b.plans_to_run(8, cfg, 256)] == [(4, 2, "dp2xtp4"), (1, 8, "dp8")]
    assert [(t, d) for t, d, _, _ in b.plans_to_run(2, cfg, 256)] == [(2, 1), (1, 2)]
    assert b.plans_to_run(8, cfg, 256)[0][3].startswith("headline")
    assert [(t, d) for t, d, _, _ in b.plans_to_run(4, cfg, 6)] == [(4, 1)]          # batch 6 does not split 4 ways
    monkeypatch.setenv("MI_BENCH_PLAN", "dp")
    assert [(t, d) for t, d, _, _ in b.plans_to_run(8, cfg, 256)] == [(1, 8)]
    monkeypatch.setenv("MI_BENCH_PLAN", "tp2")
    assert [(t, d, lab) for t, d, lab, _ in b.plans_to_run(8, cfg, 256)] == [(2, 4, "dp4xtp2")]
    monkeypatch.setenv("MI_BENCH_PLAN", "tpmax")
    assert [(t, d) for t, d, _, _ in b.plans_to_run(8, cfg, 256)] == [(4, 2)]


def test_gpus_n_without_enough_devices_exits_non_zero():
    """`python bench.py --gpus 8` with no launcher and fewer than 8 visible GPUs must not print a line at all
    (it used to run on one GPU and print n_gpus: 1)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MI_BENCH_REHEARSAL")}
    import torch

    want = torch.cuda.device_count() + 7
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr and not r.stdout.strip()

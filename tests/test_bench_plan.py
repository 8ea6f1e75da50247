"""CPU: the parallel-plan logic of bench.py (no GPU, no process group): the widest tensor-parallel plan obeys the
head-partition rules of qwen2_old.py:89-102, labels are stable, and the measured per-rank compute table covers
every (tp, dp) split the driver can ask for at N = 1, 2, 4, 8."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_plans_and_compute_table():
    from vllm_omni_amd.talker import TalkerConfig

    b = _bench()
    cfg = TalkerConfig.from_json("qwen2_5_omni_talker")          # 28 query heads, 4 KV heads
    assert [b.parallel_plan(n, cfg)[:2] for n in (1, 2, 4, 8)] == [(1, 1), (2, 1), (4, 1), (4, 2)]
    assert b.parallel_plan(8, cfg)[2] == "dp2xtp4" and b.plan_label(1, 8) == "dp8" and b.plan_label(2, 1) == "tp2"
    q3 = TalkerConfig.from_json("qwen3_omni_talker_attn")        # 16 query heads, 2 KV heads: KV replicated at tp 4, 8
    assert [b.parallel_plan(n, q3)[0] for n in (2, 4, 8)] == [2, 4, 8]
    for n in (1, 2, 4, 8):
        tp_max = b.parallel_plan(n, cfg)[0]
        for tp in (1, 2, 4, 8):
            if tp <= tp_max and n % tp == 0:
                assert (tp, n // tp) in b.COMPUTE_MS, (tp, n // tp)
    # the table is per-rank compute: more ranks never cost more than fewer
    assert b.COMPUTE_MS[(4, 2)] < b.COMPUTE_MS[(4, 1)] < b.COMPUTE_MS[(2, 1)] < b.COMPUTE_MS[(1, 1)]

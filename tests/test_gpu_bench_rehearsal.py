"""GPU: `bench.py --gpus 2` rehearsed on the one-GPU box (MI_BENCH_REHEARSAL=1: both ranks on cuda:0, torch.distributed
over gloo) so that the N > 1 path — self-spawned ranks, the safe -> fast stage ladder, the tensor-parallel plan with its
all-reduce probe, the failure net — does not rot between the driver's 8-GPU runs.  A fresh CHILD
process each time (subprocess: never an exec of the test process); a tiny model and two steps: a functional check, not
a measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env):
    env = dict(os.environ, MI_BENCH_REHEARSAL="1", MI_BENCH_TP_WATCHDOG_S="150", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "mini_talker", "--batch", "8", "--ctx", "64",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--traffic", "none", "--roofline-steps", "1"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_two_rank_rehearsal_produces_one_line_with_a_tensor_parallel_plan():
    r, d = _run({})
    assert r.returncode == 0, r.stderr[-2000:]
    assert d is not None and d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["parallelism"].startswith("tp2"), d["config"]
    assert "allreduce_probe" in d and d["allreduce_probe"].get("us_per_message", 0) > 0, d.get("allreduce_probe")
    assert any(p["parallelism"].startswith("dp2") for p in d.get("alt_plans", [])), d.get("alt_plans")
    assert "ms_per_step_p50" in d


def test_a_raising_tensor_parallel_stage_ends_in_rc_4_with_the_failure_in_the_line():
    r, d = _run({"MI_BENCH_INJECT_FAULT": "raise:tp2"})
    assert r.returncode == 4, (r.returncode, r.stderr[-2000:])
    assert d is not None and d["n_gpus"] == 2 and d.get("failed_plans"), d
    assert d.get("exit_code") == 4
    assert d["config"]["parallelism"].startswith("dp2")     # the line is built from the stage that did run

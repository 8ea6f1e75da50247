"""Is the periodic stall of the eager prefill CPU-bandwidth throttling of the container (CFS quota, 100 ms periods)?

    python tools/scratch/cpu_throttle_probe.py [bench_prefill_model args]

Prints cgroup cpu.max / cpu.stat before and after the run and the process' CPU time against wall time."""
import os
import runpy
import sys
import time


def read(p):
    try:
        return open(p).read().strip().replace("\n", "; ")
    except OSError as e:
        return f"<{e.__class__.__name__}>"


def cg():
    for base in ("/sys/fs/cgroup", "/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"):
        for f in ("cpu.max", "cpu.stat", "cpu.cfs_quota_us", "cpu.cfs_period_us"):
            p = os.path.join(base, f)
            if os.path.exists(p):
                print(f"  {p}: {read(p)}")


print("before:")
cg()
print("  cpus allowed:", len(os.sched_getaffinity(0)), "nproc", os.cpu_count())
t0, c0 = time.perf_counter(), os.times()
sys.argv = ["bench_prefill_model.py"] + sys.argv[1:]
here = os.path.dirname(os.path.abspath(__file__))
runpy.run_path(os.path.join(here, "..", "bench_prefill_model.py"), run_name="__main__")
t1, c1 = time.perf_counter(), os.times()
print("after:")
cg()
print(f"  wall {t1 - t0:.2f} s, user {c1.user - c0.user:.2f} s, sys {c1.system - c0.system:.2f} s; threads now:",
      read("/proc/self/status").split("Threads:")[1].split(";")[0].strip())

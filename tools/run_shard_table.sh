#!/bin/bash
# One-GPU measurement of every (TP, DP) split of the headline job: per-rank step time with collectives stubbed
# (bench.py --simulate-shard) + a rocprofv3 kernel table per plan.  Usage: tools/run_shard_table.sh <tag> [plans...]
# Writes gpurun_out/<tag>_shard_<plan>.json and gpurun_out/<tag>_shard_<plan>_kernel_stats.md
set -u
TAG=${1:-r03}; shift || true
PLANS=${*:-"1,1 2,1 4,1 4,2 2,4 1,8"}
mkdir -p gpurun_out
export TMPDIR=/tmp
for p in $PLANS; do
  name=tp${p%,*}dp${p#*,}
  if [ "$p" = "1,1" ]; then args=""; else args="--simulate-shard $p"; fi
  python bench.py $args --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_shard_${name}.json 2> gpurun_out/${TAG}_shard_${name}.err || { echo "bench $name failed"; tail -5 gpurun_out/${TAG}_shard_${name}.err; exit 1; }
  python - "$name" gpurun_out/${TAG}_shard_${name}.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"{sys.argv[1]:8s} {d['ms_per_step']:8.3f} ms/step  decode kernel {d['roofline'].get('avg_launch_ms')} ms  frac {d['roofline'].get('frac')}")
PY
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o run -- python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2> gpurun_out/${TAG}_shard_${name}_rocprof.err || { echo "rocprof $name failed"; tail -5 gpurun_out/${TAG}_shard_${name}_rocprof.err; exit 1; }
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  if [ -z "$f" ]; then f=$(find /tmp/prof_$name -name "*results.db" | head -1); fi
  python tools/summarize_rocprof.py "$f" --top 16 > gpurun_out/${TAG}_shard_${name}_kernel_stats.md
  db=$(find /tmp/prof_$name -name "*results.db" | head -1)
  if [ -n "$db" ]; then python tools/step_timeline.py "$db" > gpurun_out/${TAG}_shard_${name}_timeline.md 2>&1 || true; fi
done

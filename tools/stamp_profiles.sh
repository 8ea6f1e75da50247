#!/bin/bash
# Copies gpurun_out/<tag>_* evidence into profiles/ and names the commit the numbers were measured at (the GPU boxes get a
# snapshot without .git, so the stamp is put on here).  Usage: tools/stamp_profiles.sh <tag> [commit]
set -eu
TAG=$1; COMMIT=${2:-$(git rev-parse --short HEAD)}
for f in gpurun_out/${TAG}_shard_*kernel_stats.md gpurun_out/${TAG}_shard_*timeline.md gpurun_out/${TAG}_shard_table.log gpurun_out/${TAG}_gpu_suite_summary.txt gpurun_out/${TAG}_configs.log gpurun_out/${TAG}_prefill_kernel.log gpurun_out/${TAG}_code_predictor.log gpurun_out/${TAG}_code_predictor_b*_kernel_stats.md gpurun_out/${TAG}_code_predictor_b*_timeline.md; do
  [ -f "$f" ] || continue
  { echo "measured at commit ${COMMIT} ($(date -u +%Y-%m-%d))"; cat "$f"; } > profiles/$(basename "$f")
done
for f in gpurun_out/${TAG}_shard_*.json gpurun_out/${TAG}_bench_line.json; do
  [ -f "$f" ] || continue
  cp "$f" profiles/$(basename "$f")
done
echo "${COMMIT}" > profiles/${TAG}_measured_at_commit.txt

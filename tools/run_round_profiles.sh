#!/bin/bash
# One GPU call that regenerates the round's evidence at the code it is run on: the full `-m gpu` suite, the per-plan
# shard table with a rocprofv3 kernel table + timeline per plan, the driver-shaped bench line (incl. its PMC child passes
# and the CPU baseline), the other BASELINE configs, the prefill kernel.  Usage: tools/run_round_profiles.sh <tag>
# Everything lands in gpurun_out/<tag>_*; copy what is to be judged into profiles/ (tools/stamp_profiles.sh).
set -u
TAG=${1:-r04}
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_gpu_suite.log 2>&1
tail -3 gpurun_out/${TAG}_gpu_suite.log | tee gpurun_out/${TAG}_gpu_suite_summary.txt
bash tools/run_shard_table.sh ${TAG} 2>&1 | tee gpurun_out/${TAG}_shard_table.log
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_err.log; echo "bench rc=$?"
run() { python bench.py "$@" --steps 30 --warmup 5 --no-cpu-baseline --traffic none 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['metric'][:72], '|', d['ms_per_step'], 'ms', d['value'], 'tok/s p10/p50/p90', d.get('ms_per_step_p10'), d.get('ms_per_step_p50'), d.get('ms_per_step_p90'), '| kernel', r.get('avg_launch_ms'), r.get('frac'))"; }
{ run --batch 64 --ctx 2048; run --quantization int8; run --config qwen3_omni_talker --batch 512; run --config qwen3_omni_talker --batch 512 --code-predictor;
  run --config qwen3_omni_talker_attn --batch 512; run --config qwen3_omni_talker_attn --batch 512 --quantization int8; run --batch 1; run --batch 8; run --batch 32; } 2>&1 | tee gpurun_out/${TAG}_configs.log
{ python tools/bench_prefill_kernel.py --iters 20; python tools/bench_prefill_kernel.py --iters 20 --nq 16 --nkv 2 --d 64 --chunks 8192,2048; } 2>&1 | grep prompt | tee gpurun_out/${TAG}_prefill_kernel.log
# the code predictor: per call (chain of launches / one-launch stack), its kernel table + call timeline, the sampler
{ for b in 32 128 512; do python tools/bench_code_predictor.py --batch $b --skip-reprefill 2>&1 | grep "code predictor"; done
  echo "VLLM_OMNI_AMD_CP_ONE_LAUNCH=1:"; for b in 32 512; do VLLM_OMNI_AMD_CP_ONE_LAUNCH=1 python tools/bench_code_predictor.py --batch $b --skip-reprefill 2>&1 | grep "code predictor"; done
  python tools/bench_sampler.py 2>&1 | grep rows; } | tee gpurun_out/${TAG}_code_predictor.log
for b in 32 512; do
  rm -rf /tmp/prof_cp$b
  rocprofv3 --kernel-trace --stats -d /tmp/prof_cp$b -o cp -- python tools/bench_code_predictor.py --batch $b --skip-reprefill > /dev/null 2>&1 || { echo "rocprof predictor $b failed"; continue; }
  db=$(find /tmp/prof_cp$b -name "*results.db" | head -1)
  python tools/summarize_rocprof.py "$db" --top 20 > gpurun_out/${TAG}_code_predictor_b${b}_kernel_stats.md
  python tools/step_timeline.py "$db" --anchor block_attn --layers 150 > gpurun_out/${TAG}_code_predictor_b${b}_timeline.md 2>&1 || true
done

set -e
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_block_attention.py -x -q 2>&1 | tail -15
timeout -k 10 600 python -m pytest tests/test_gpu_code_predictor.py -x -q 2>&1 | tail -5
echo "== predictor B512 new"; python tools/bench_code_predictor.py --batch 512 --skip-reprefill 2>&1 | tail -1
echo "== predictor B32 new"; python tools/bench_code_predictor.py --batch 32 --skip-reprefill 2>&1 | tail -1
rm -rf /tmp/prof_cp
rocprofv3 --kernel-trace --stats -d /tmp/prof_cp -o cp -- python tools/bench_code_predictor.py --batch 512 --skip-reprefill > /dev/null 2> gpurun_out/r03n_cp_rocprof.err
f=$(find /tmp/prof_cp -name "*kernel_stats.csv" | head -1)
if [ -z "$f" ]; then f=$(find /tmp/prof_cp -name "*results.db" | head -1); fi
python tools/summarize_rocprof.py "$f" --top 12 > gpurun_out/r03n_code_predictor_b512_kernel_stats.md
cat gpurun_out/r03n_code_predictor_b512_kernel_stats.md

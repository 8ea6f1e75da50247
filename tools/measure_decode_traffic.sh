#!/bin/bash
# HBM traffic per launch of the paged-attention decode kernel from PMC counters, as MI355X_MICROARCH.md's HBM section
# prescribes: two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE do not fit one pass), no trace domains.
# Writes profiles/pmc_decode_bytes_per_launch.json (read by bench.py when it cannot measure in the run).
set -eu
OUT=${1:-profiles/pmc_decode_bytes_per_launch.json}
export TMPDIR=/tmp
rm -rf /tmp/pmc_fetch /tmp/pmc_write
t0=$(date +%s)
rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_fetch -o run --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --roofline-steps 0 --traffic none > /tmp/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_write -o run --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --roofline-steps 0 --traffic none > /tmp/pmc_write.log 2>&1
t1=$(date +%s)
python tools/summarize_pmc.py --json "$OUT" "$(find /tmp/pmc_fetch -name '*counter_collection.csv' | head -1)" "$(find /tmp/pmc_write -name '*counter_collection.csv' | head -1)" decode_mfma_kernel
echo "two PMC passes took $((t1 - t0)) s"

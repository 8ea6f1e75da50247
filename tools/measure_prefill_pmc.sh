#!/bin/bash
# PMC readings of the prefill attention kernel: three separate rocprofv3 --pmc passes (no trace domains), mean per dispatch
# of the kernel over the bench tool's launches.  Usage: tools/measure_prefill_pmc.sh <out.log> [bench_prefill_kernel.py args]
# (default: the 8192-token prompt of the headline talker in one chunk; pass e.g. --nq 16 --nkv 2 --d 64 for the Qwen3 shape)
set -u
OUT=$1; shift
export TMPDIR=/tmp
ARGS=${*:-"--chunks 8192 --iters 2"}
: > "$OUT"
pass() {
  local n=$1 ctrs=$2
  rm -rf /tmp/pmc_pf_$n
  rocprofv3 --pmc $ctrs -d /tmp/pmc_pf_$n -o run --output-format csv -- python tools/bench_prefill_kernel.py $ARGS > /tmp/pmc_pf_$n.log 2>&1 || { echo "pass $n failed" >> "$OUT"; tail -3 /tmp/pmc_pf_$n.log >> "$OUT"; return; }
  echo "== pass $n ($ARGS)" >> "$OUT"
  python tools/summarize_pmc.py "$(find /tmp/pmc_pf_$n -name '*counter_collection.csv' | head -1)" prefill_mfma >> "$OUT"
}
pass 1 "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES"
pass 2 "SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
pass 3 "SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
cat "$OUT"

#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of
# the benchmark workload. Summaries land in gpurun_out/prof/; copy what should be
# kept into profiles/.
#   tools/profile_gpu.sh [extra bench.py args]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
case "$*" in *"--config c5"*) export PROF_CONFIG=c5;; *) export PROF_CONFIG=c3;; esac
ARGS="--steps 4 --warmup 2 --no-cpu-baseline --no-fused-extra --no-partitioned-extra --no-verify $*"
PARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-fused-extra --no-partitioned-extra --no-verify $*"
run() { name=$1; shift; echo "== $name"; timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $BARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }; }
BARGS=$ARGS  run stats --kernel-trace --stats || exit 1
if [ "${PROF_ONLY:-}" = "stats" ]; then python3 "$ROOT/tools/summarize_prof.py" "$OUT" | tee "$OUT/summary.txt"; exit 0; fi  # kernel trace only
BARGS=$PARGS run pmc_fetch --pmc FETCH_SIZE || exit 1
BARGS=$PARGS run pmc_write --pmc WRITE_SIZE || exit 1
BARGS=$PARGS run pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY || exit 1
BARGS=$PARGS run pmc_lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU || exit 1
BARGS=$PARGS run pmc_tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum || exit 1
find "$OUT" -name "*.csv" | head -50
python3 "$ROOT/tools/summarize_prof.py" "$OUT" | tee "$OUT/summary.txt"

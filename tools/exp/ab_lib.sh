#!/bin/bash
# A/B of library builds on the bench workload, interleaved, one process per run (on the GPU box):
#   tools/exp/ab_lib.sh OUT_PREFIX ROUNDS "bench args" LIB1 LIB2 ...      (LIB = path of a libwpdsp.so, or "tree")
# Writes gpurun_out/OUT_PREFIX.txt: per run the step and the three kernel times.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/$1.txt; ROUNDS=$2; BARGS=$3; shift; shift; shift
mkdir -p "$ROOT/gpurun_out"; : > "$OUT"
for r in $(seq 1 "$ROUNDS"); do
  for lib in "$@"; do
    if [ "$lib" = "tree" ]; then L=""; else L="--lib $ROOT/$lib"; fi
    echo "running: $lib"   # (a run that faults the GPU takes the box with it: this line names it)
    timeout -k 10 150 python3 "$ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-fused-extra --no-partitioned-extra --no-verify $BARGS $L 2>/dev/null \
      | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=r['roofline']['kernel_ms']
print('%-28s step %.3f ms  ' % ('$lib', r['ms_per_step']) + '  '.join('%s %.3f' % (a,b) for a,b in k.items()))" | tee -a "$OUT" || { echo "run failed: $lib" | tee -a "$OUT"; exit 1; }
  done
done

#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the other
# configurations (4-bar stereo reverb, 4096-stem master reduce, per-GPU share of
# the 2048-track render with shared and per-track IRs).  Output: gpurun_out/prof_cfg/.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_cfg
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in c2 c4 c5 c5ir; do
  echo "== $cfg"
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$cfg" -- python3 "$ROOT/tools/bench_configs.py" $cfg > "$OUT/$cfg.log" 2>&1 || { echo "$cfg failed"; tail -5 "$OUT/$cfg.log"; exit 1; }
  grep '^{' "$OUT/$cfg.log"
  f=$(find "$OUT/$cfg" -name "*kernel_stats.csv" | head -1)
  echo "-- $cfg kernel stats (name, calls, total ns, avg ns, %)"; cut -d, -f1-5 "$f" | head -12
done | tee "$OUT/summary.txt"

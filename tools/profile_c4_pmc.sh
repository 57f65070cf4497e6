#!/bin/bash
# Runs on the GPU box: HBM traffic counters of the 4096-stem master reduce (k_mix), one
# counter group per pass.  Output: gpurun_out/prof_c4/.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_c4
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/bench_configs.py" c4 > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_mix<" in row["Kernel_Name"]:
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
print("k_mix<2, 4>, 4096 stereo stems x 2 880 000 frames, per launch:")
for k in sorted(acc):
    print("  %-24s %16.6g  (launches %d)" % (k, acc[k][0] / acc[k][1], acc[k][1]))
if "FETCH_SIZE" in acc and "WRITE_SIZE" in acc:
    b = 2 * acc["FETCH_SIZE"][0] / acc["FETCH_SIZE"][1] * 1024 + acc["WRITE_SIZE"][0] / acc["WRITE_SIZE"][1] * 1024
    print("  HBM bytes (2*FETCH_SIZE + WRITE_SIZE, KiB units): %.4g   algorithmic: %.4g" % (b, 2880000 * (4 * 2 * 4096 + 16)))
PY

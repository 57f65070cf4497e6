#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace stats + PMC passes) into one text
summary: per kernel average duration, and per kernel per-launch counter values."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    n = name.split("(")[0]
    for k in ("k_lf_fwd", "k_lf_mid", "k_lf_inv", "k_fft_fwd", "k_fft_inv", "k_mac", "k_mix_prep", "k_mix", "k_hpack", "k_synth_noise"):
        if k in n:
            t = n[n.index(k):]
            return t[:52]
    return n[:40]


print("# kernel-trace stats")
for f in sorted(glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print("%-52s calls %6s  avg %10.1f us  total %10.1f us  %5s%%" % (
                short(row["Name"]), row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e3, row["Percentage"]))

# The stats average above includes the warm-up launch(es), which run cold (first touch of the
# workspace).  bench.py times only the steps after warm-up, so the figure to compare with its
# HIP-event timing is the average over the timed launches: every launch after the first WARM.
WARM = int(os.environ.get("PROF_WARMUP", "2"))
print("# kernel-trace, per launch (us) and average after the %d warm-up launch(es)" % WARM)
for f in sorted(glob.glob(os.path.join(root, "stats", "**", "*kernel_trace.csv"), recursive=True)):
    per = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            per[short(row["Kernel_Name"])].append((int(row["Start_Timestamp"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3))
    for k, v in per.items():
        if not any(t in k for t in ("k_lf_", "k_fft_fwd<", "k_fft_fwd_pair<", "k_fft_inv", "k_mac", "k_mix<")) or len(v) <= WARM:
            continue
        d = [x[1] for x in sorted(v)]
        print("%-52s %s  -> steady avg %10.1f us" % (k, " ".join("%.0f" % x for x in d), sum(d[WARM:]) / len(d[WARM:])))
    # With several runs of frames a kernel is launched once per run and step (and k_lf_mid_jobs<8> lists a long and a
    # short launch alternately): per step, the launches of each kernel family added up.  A step starts at a launch of
    # the forward kernel with the longest average duration.
    fam = lambda k: "k_lf_inv" if k.startswith("k_lf_inv") else "k_lf_mid" if k.startswith("k_lf_mid") else k.split("<")[0]
    firsts = [k for k in per if fam(k) in ("k_lf_fwd", "k_fft_fwd", "k_fft_fwd_pair") and len(per[k]) > WARM]
    if firsts:
        lead = max(firsts, key=lambda k: sum(x[1] for x in per[k]) / len(per[k]))
        starts = sorted(x[0] for x in per[lead])
        if len(starts) > WARM:
            t0, nst = starts[WARM], len(starts) - WARM
            print("# per step after the warm-up steps: launches of one kernel family added up (us); inverse passes of different runs of frames")
            print("# run on different streams and overlap: their durations add up to more than the pass takes (bench.py's roofline.kernel_ms has the pass)")
            sums = defaultdict(float)
            for k, v in per.items():
                if any(t in k for t in ("k_lf_", "k_fft_fwd", "k_fft_inv", "k_mac", "k_mix<")):
                    sums[fam(k)] += sum(x[1] for x in v if x[0] >= t0)
            for k in sorted(sums):
                print("%-52s %10.1f us per step" % (k, sums[k] / nst))

print("# PMC (per launch averages)")
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
peak = defaultdict(lambda: defaultdict(float))  # largest single launch: the bench launch where a kernel also runs once on the impulse response
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row["Kernel_Name"])
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
            peak[k][row["Counter_Name"]] = max(peak[k][row["Counter_Name"]], float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        print("    %-24s %16.4g   (launches %d)" % (c, s / n, n))

# HBM-side bytes per launch for the bench kernels: FETCH_SIZE and WRITE_SIZE are in KiB and
# come from separate passes; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so it is
# doubled (checked here against k_mix and k_mac, whose read bytes are known exactly).
import json
traffic = {}
tot = defaultdict(lambda: [0.0, 0.0])
for k in acc:
    base = k.split("<")[0]
    if base == "k_lf_mid_jobs":  # the row pass of batches
        base = "k_lf_mid"
    if base == "k_fft_fwd_pair":  # the forward transform of the bench path (two segments per workgroup)
        base = "k_fft_fwd"
    if base == "k_fft_inv_mix":
        base = "k_fft_inv_mix_w" if "true" in k else "k_fft_inv_mix"
    if base == "k_lf_inv":  # k_lf_inv<R, C, COLS, MIX, WRITE>
        flags = k.replace(" ", "").split(",")[3:5]
        base = "k_lf_inv" + ("_mix" if flags and flags[0].startswith("true") else "") + ("_w" if len(flags) > 1 and flags[0].startswith("true") and flags[1].startswith("true") else "")
    if k.startswith("k_lf_mid<true"):
        continue  # the one-off kernel-spectrum launch
    if base in ("k_lf_fwd", "k_lf_mid", "k_lf_inv", "k_lf_inv_mix", "k_lf_inv_mix_w", "k_fft_fwd", "k_mac", "k_fft_inv", "k_fft_inv_mix", "k_fft_inv_mix_w", "k_mix") and "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
        # the PMC passes run ONE bench step: every launch of the family added up (the runs of frames of the long-frame
        # form are separate launches; the launches on the impulse response add a few megabytes)
        tot[base][0] += acc[k]["FETCH_SIZE"][0]
        tot[base][1] += acc[k]["WRITE_SIZE"][0]
for base, (f, w) in tot.items():
    traffic[base] = int(2 * f * 1024 + w * 1024)
print("# traffic bytes per step (2*FETCH_SIZE + WRITE_SIZE), all launches of a kernel family:", traffic)
# the profiled command is a bench workload (tools/profile_gpu.sh [--config c5]): PROF_CONFIG names it
cfg = os.environ.get("PROF_CONFIG", "c3")
traffic["workload"] = {"tracks_per_gpu": 256, "frames": 2880000, "ir_len": 192000 if cfg == "c5" else 96000, "path": "onepass"}
with open(os.path.join(root, "traffic.json"), "w") as fh:
    json.dump(traffic, fh)

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
    for k in ("k_fft_fwd", "k_fft_inv", "k_mac", "k_mix_prep", "k_mix", "k_hpack", "k_synth_noise"):
        if k in n:
            t = n[n.index(k):]
            return t[:40]
    return n[:40]


print("# kernel-trace stats")
for f in sorted(glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            print("%-42s calls %6s  avg %10.1f us  total %10.1f us  %5s%%" % (
                short(row["Name"]), row["Calls"], float(row["AverageNs"]) / 1e3, float(row["TotalDurationNs"]) / 1e3, row["Percentage"]))

print("# PMC (per launch averages)")
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row["Kernel_Name"])
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        print("    %-24s %16.4g   (launches %d)" % (c, s / n, n))

#!/usr/bin/env python3
"""Largest loop of a kernel in a gfx950 assembly listing: instruction histogram and the registers that are
live into it (read before written inside the body)."""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2]
m = [m for m in re.finditer(r"^(_Z[^\n:]*):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M) if pat in m.group(1)][0]
body = m.group(2).splitlines()
labels = {}
for i, l in enumerate(body):
    mm = re.match(r"^(\.LBB\d+_\d+):", l)
    if mm:
        labels[mm.group(1)] = i
best = None
for i, l in enumerate(body):
    mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        if best is None or i - labels[mm.group(1)] > best[1] - best[0]:
            best = (labels[mm.group(1)], i)
a, b = best
c = collections.Counter()
written, livein = set(), set()


def regs(tok):
    out = []
    for kind, lo, hi in re.findall(r"\b([va])\[(\d+):(\d+)\]", tok):
        out += ["%s%d" % (kind, k) for k in range(int(lo), int(hi) + 1)]
    for kind, n in re.findall(r"\b([va])(\d+)\b", tok):
        out.append("%s%s" % (kind, n))
    return out


for l in body[a:b]:
    l = l.split(";")[0].strip()
    if not l or l.startswith(".") or l.endswith(":"):
        continue
    op = l.split()[0]
    c[op] += 1
    ops = l[len(op):].split(",")
    if not ops:
        continue
    stores = op.startswith(("global_store", "ds_write", "scratch_store", "buffer_store", "s_", "v_cmp"))
    dst = [] if stores else regs(ops[0])
    srcs = regs(",".join(ops if stores else ops[1:]))
    if op.startswith(("v_fmac", "v_mac")):
        srcs += dst
    for r in srcs:
        if r not in written:
            livein.add(r)
    for r in dst:
        written.add(r)
print(m.group(1)[:60], "loop lines", b - a, "instr", sum(c.values()), "valu", sum(v for k, v in c.items() if k.startswith("v_")),
      "salu", sum(v for k, v in c.items() if k.startswith("s_")))
print("live-in VGPRs", len([r for r in livein if r[0] == "v"]), "AGPRs", len([r for r in livein if r[0] == "a"]))
print(", ".join("%s %d" % kv for kv in c.most_common(24)))

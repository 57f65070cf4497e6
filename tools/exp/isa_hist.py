#!/usr/bin/env python3
"""Instruction histogram per kernel of a gfx950 assembly listing (hipcc --cuda-device-only -S)."""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z[^\n:]*):[^\n]*\n(.*?)s_endpgm", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    c = collections.Counter()
    for line in body.splitlines():
        line = line.strip()
        if not line or line.startswith((".", ";")) or line.endswith(":"):
            continue
        c[line.split()[0]] += 1
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print(name[:70], "total", sum(c.values()), "valu", valu)
    print("   ", ", ".join("%s %d" % kv for kv in c.most_common(16)))

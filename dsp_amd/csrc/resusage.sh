#!/bin/bash
# Prints VGPR / SGPR / scratch / LDS / occupancy per kernel of one .hip file.
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -o /dev/null "$f" "$@" -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re,subprocess
cur={}
rows=[]
for line in sys.stdin:
    m=re.search(r"remark: ([A-Za-z ]+?): *(.+?) \[-Rpass", line)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2).strip()
    if k=="Function Name":
        cur={"name":v}; rows.append(cur)
    else: cur[k]=v
for r in rows:
    n=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    n=re.sub(r"\(.*","",n)
    print("%-58s vgpr %-4s agpr %-3s sgpr %-4s scratch %-5s lds %-6s occ %s" % (n[:58], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
'

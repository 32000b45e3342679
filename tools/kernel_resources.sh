#!/bin/bash
# VGPR / SGPR / spill / LDS / occupancy of every kernel, from the compiler's resource-usage remarks (gfx950 code objects).
# usage: tools/kernel_resources.sh > profiles/rNN_resources.txt
set -e
cd "$(dirname "$0")/../stm-multifrontal-qr-factorization-empowered-by-gcn_amd/csrc"
for f in *.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=200000 -c -o /dev/null \
     -Rpass-analysis=kernel-resource-usage "$f" 2>&1 | python3 -c '
import sys, re
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m: cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark: .*?\s{2,}([A-Za-z ]+?)(?: \[.*?\])?: (\S+)", line)
    if m and cur is not None: cur[m.group(1).strip()] = m.group(2)
for r in rows:
    print("%-28s VGPRs %-4s AGPRs %-3s SGPRs %-4s spillV %-4s spillS %-4s scratch %-6s occupancy %-2s LDS %s" % (
        r["name"], r.get("VGPRs","?"), r.get("AGPRs","?"), r.get("SGPRs","?"), r.get("VGPRs Spill", r.get("VGPR Spill","?")),
        r.get("SGPRs Spill", r.get("SGPR Spill","?")), r.get("ScratchSize","?"), r.get("Occupancy","?"), r.get("LDS Size","?")))
'
done

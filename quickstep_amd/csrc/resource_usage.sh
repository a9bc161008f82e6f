#!/bin/bash
# Prints per-kernel VGPR/SGPR/scratch/LDS/occupancy for one .hip file (hipcc -Rpass-analysis).
f=$1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -munsafe-fp-atomics \
  -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 | python3 -c '
import sys,re
cur={}
for line in sys.stdin:
    m=re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|LDS Size \[bytes/block\]): (\S+)",line)
    if not m:
        m2=re.search(r"Name: (\S+)",line)
        if m2:
            if cur: print(cur)
            cur={"name":m2.group(1)[:70]}
        continue
    k,v=m.group(1),m.group(2)
    cur[k.split(" ")[0]]=v
if cur: print(cur)
'

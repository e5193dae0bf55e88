#!/bin/bash
# usage: tools/kres.sh <file.hip> [extra flags]  -- VGPR / spill / scratch / LDS per kernel (compiler's resource remarks)
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize "$@" -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 |
 python3 -c '
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur=m.group(1); print(); print(cur[:110],end="  ")
    for k in ("VGPRs:","VGPR Spill","SGPRs:","ScratchSize","LDS Size","Occupancy"):
        m=re.search(k+r"[^0-9]*(\d+)",l)
        if m and k in l: print(k.split()[0].strip(":")+("Spill" if "Spill" in k and "VGPR" in k else "")+"="+m.group(1),end=" ")
print()'

#!/bin/bash
# usage: scripts/kres.sh <file.hip> <regex>  — VGPRs / spills / occupancy per kernel (gfx950)
f=$1; re=$2
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -ffp-contract=off -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c "
import sys,re
cur=None; rows={}
for l in sys.stdin:
    if 'error' in l: print(l.strip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}
    for key in ('VGPRs','VGPRs Spill','Occupancy [waves/SIMD]','LDS Size [bytes/block]','SGPRs Spill'):
        m=re.search(r'remark:\s+'+re.escape(key)+r': (\d+)',l)
        if m and cur: rows[cur][key]=m.group(1)
for k,v in rows.items():
    if re.search(sys.argv[1],k): print(k[:70], 'vgpr',v.get('VGPRs'),'spill',v.get('VGPRs Spill'),'occ',v.get('Occupancy [waves/SIMD]'),'lds',v.get('LDS Size [bytes/block]'))
" "$re"

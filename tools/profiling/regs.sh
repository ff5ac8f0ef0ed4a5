#!/bin/bash
# compile-only register report for the berg kernels; extra flags pass through ($@)
cd /root/repo/icebergs_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" -c -o /dev/null kid_hip.hip 2>&1 | python3 -c "
import re,sys,subprocess
cur=None; rows={}
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)',line)
    if m and cur: rows[cur][m.group(1).strip()]=int(m.group(2))
for k,v in rows.items():
    if 'rocprim' in k: continue
    n=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip().replace('(anonymous namespace)::','')
    if ('14u' in n or ', 2u' in n or '12u' in n) and ('<true, true' in n or '<false, false' in n):
        print('%-60s VGPR %3d AGPR %3d scratch %5d occ %d'%(n[:60],v.get('VGPRs',0),v.get('AGPRs',0),v.get('ScratchSize',0),v.get('Occupancy',0)))
"

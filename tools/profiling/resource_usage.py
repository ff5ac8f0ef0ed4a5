#!/usr/bin/env python3
"""Prints VGPR/SGPR/scratch/occupancy per kernel from hipcc -Rpass-analysis=kernel-resource-usage."""
import re, subprocess, sys
out = subprocess.run(["make", "-s", "-C", "icebergs_amd/csrc", "resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if "rocprim" in k:
        continue
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name)[:70]
    print(f"{name:70s} VGPR {v.get('VGPRs',0):4d} AGPR {v.get('AGPRs',0):3d} SGPR {v.get('TotalSGPRs',0):4d} scratch {v.get('ScratchSize',0):5d} occ {v.get('Occupancy',0)}")

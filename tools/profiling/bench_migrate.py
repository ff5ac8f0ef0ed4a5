#!/usr/bin/env python3
"""Times the migration calls (kid_pack_emigrants x 4 directions, kid_unpack_immigrants; send_bergs_to_other_pes
FW:2997-3247) on a config-2 population: each step a few hundred bergs of a million leave a sub-domain.  Every call ends
with a host read (the count, then the records), so these are latencies."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icebergs_amd import synthetic as S  # noqa: E402
from icebergs_amd import types as T  # noqa: E402
from icebergs_amd.framework import Icebergs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bergs", type=int, default=1_000_000)
ap.add_argument("--leavers", type=int, default=500)
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
grid, p, b = S.config_c2(n=a.bergs, seed=4)
d = grid["desc"]
ib = Icebergs(grid, p, capacity=a.bergs + 8 * a.leavers * (a.iters + 3))
ib.upload_bergs(b)
ib.run(1)
rng = np.random.default_rng(0)
# records of bergs that arrive: taken from this population, positions kept (they land in the cells they came from)
dirs = [T.ENUMS[k] for k in ("KID_DIR_E", "KID_DIR_W", "KID_DIR_N", "KID_DIR_S")]
t_pack, t_unpack, packed = 0.0, 0.0, 0
t_pack2, t_unpack2 = 0.0, 0.0
for it in range(a.iters + 2):
    # push some live bergs over the edges by hand (the step would do this to a few hundred per step at 1e6 bergs / 64 ranks)
    cur = ib.download_bergs()
    rows = rng.choice(np.flatnonzero(cur["alive"] != 0), size=4 * a.leavers, replace=False)
    keep_i, keep_j = cur["ine"][rows].copy(), cur["jne"][rows].copy()
    cur["ine"][rows[0 * a.leavers:1 * a.leavers]] = d.iec + 1
    cur["ine"][rows[1 * a.leavers:2 * a.leavers]] = d.isc - 1
    cur["jne"][rows[2 * a.leavers:3 * a.leavers]] = d.jec + 1
    cur["jne"][rows[3 * a.leavers:4 * a.leavers]] = d.jsc - 1
    ib.upload_bergs(cur)
    ib.sync()
    t0 = time.perf_counter()
    bufs = [ib.pack_emigrants(k) for k in dirs]
    t1 = time.perf_counter()
    for buf in bufs:
        ib.unpack_immigrants(buf)          # check_and_find_cell puts each back into the cell that holds its position
    t2 = time.perf_counter()
    if it >= 2:
        t_pack += t1 - t0; t_unpack += t2 - t1; packed += sum(len(x) for x in bufs)
    # the same exchange with the pair calls (east + west, north + south in one launch each way)
    cur = ib.download_bergs()
    rows = rng.choice(np.flatnonzero(cur["alive"] != 0), size=4 * a.leavers, replace=False)
    cur["ine"][rows[0 * a.leavers:1 * a.leavers]] = d.iec + 1
    cur["ine"][rows[1 * a.leavers:2 * a.leavers]] = d.isc - 1
    cur["jne"][rows[2 * a.leavers:3 * a.leavers]] = d.jec + 1
    cur["jne"][rows[3 * a.leavers:4 * a.leavers]] = d.jsc - 1
    ib.upload_bergs(cur)
    ib.sync()
    t0 = time.perf_counter()
    ew = ib.pack_emigrants_pair(0)
    ns = ib.pack_emigrants_pair(1)
    t1 = time.perf_counter()
    ib.unpack_immigrants_pair(*ew)
    ib.unpack_immigrants_pair(*ns)
    t2 = time.perf_counter()
    assert sum(len(x) for x in ew + ns) == 4 * a.leavers
    if it >= 2:
        t_pack2 += t1 - t0; t_unpack2 += t2 - t1
print(json.dumps({"what": "kid_pack_emigrants x4 + kid_unpack_immigrants x4", "bergs": a.bergs, "records_per_step": packed // a.iters,
                  "pack_us_per_step": round(t_pack * 1e6 / a.iters, 1), "unpack_us_per_step": round(t_unpack * 1e6 / a.iters, 1),
                  "pair_pack_us_per_step": round(t_pack2 * 1e6 / a.iters, 1), "pair_unpack_us_per_step": round(t_unpack2 * 1e6 / a.iters, 1)}))
ib.close()

#!/usr/bin/env python3
"""Times kid_calving (calving block + accumulate_calving + calve_icebergs on the device, icebergs.F90:5203-5231, 6153-6402)
with the coupler arrays resident in HBM.  The call ends with the one host read the path needs (how many bergs were
appended), so the figure is a latency, not a bandwidth."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icebergs_amd import synthetic as S  # noqa: E402
from icebergs_amd import types as T  # noqa: E402
from icebergs_amd.framework import Icebergs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ni", type=int, default=1440)
ap.add_argument("--nj", type=int, default=1080)
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--frac", type=float, default=0.01)
a = ap.parse_args()
grid = S.c2_forcing(S.latlon_grid(ni=a.ni, nj=a.nj, dlon=360.0 / a.ni, dlat=160.0 / a.nj))
p = S.default_params()
cp = S.calving_params(p)
calv, hflx = S.coupler_calving(grid, seed=1, frac=a.frac, buckets=0.6)
dev = [torch.from_numpy(np.ascontiguousarray(v)).cuda() for v in (calv, hflx)]
stream = torch.cuda.Stream()
cap = 4_000_000
ib = Icebergs(grid, p, capacity=cap)
ib.set_stream(stream.cuda_stream)
ib.set_forcing(grid["forcing"])
ib.set_calving_params(cp)
ib.upload_bergs(S.place_bergs(grid, 1000, 1, (3, a.ni - 3), (3, a.nj - 3)))
for _ in range(5):
    ib.calving(dev[0].data_ptr(), dev[1].data_ptr(), on_device=True)
n0 = ib.num_bergs()[0]
t0 = time.perf_counter()
for _ in range(a.iters):
    ib.calving(dev[0].data_ptr(), dev[1].data_ptr(), on_device=True)
us = (time.perf_counter() - t0) * 1e6 / a.iters
n1 = ib.num_bergs()[0]
ncell = ib.ni * ib.nj
nbytes = 8 * ncell * (2 + 3 + 2 * 2 + 2 * T.ENUMS["KID_NCLASSES"] + 2 * T.ENUMS["KID_NCLASSES"])   # inputs, static, planes r+w, buckets r+w twice
print(json.dumps({"what": "kid_calving", "grid": [a.ni, a.nj], "calving_cells": int((calv > 0).sum()), "us_per_call": round(us, 1),
                  "bergs_calved_per_call": round((n1 - n0) / a.iters, 1), "plane_traffic_MB": round(nbytes / 1e6, 1)}))
ib.close()

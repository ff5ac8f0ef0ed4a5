#!/usr/bin/env python3
"""Times kid_ingest_forcing (forcing ingest on the device, icebergs.F90:5236-5383) with the coupler arrays resident in HBM.
Algorithmic bytes per call: every input array read once + the 11 planes written once (+ ua/va and the wrapped halo
columns read back), reported against the HBM roofline."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icebergs_amd import synthetic as S  # noqa: E402
from icebergs_amd import types as T  # noqa: E402
from icebergs_amd.framework import Icebergs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ni", type=int, default=1440)
ap.add_argument("--nj", type=int, default=1080)
ap.add_argument("--vel", default="C")
ap.add_argument("--stress", default="A")
ap.add_argument("--iters", type=int, default=200)
a = ap.parse_args()
grid = S.latlon_grid(ni=a.ni, nj=a.nj, dlon=360.0 / a.ni, dlat=160.0 / a.nj)
args = S.coupler_forcing(grid, seed=1, vel_stagger=a.vel, stress_stagger=a.stress, kelvin=True)
dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in args.items()}
stream = torch.cuda.Stream()
ib = Icebergs(grid, S.default_params(), capacity=16)
ib.set_stream(stream.cuda_stream)
ptrs = {k: (t.data_ptr(), tuple(t.shape)) for k, t in dev.items()}
kw = dict(vel_stagger=a.vel, stress_stagger=a.stress, cyclic_x=True, on_device=True)
with torch.cuda.stream(stream):
    for _ in range(10):
        ib.ingest_forcing(ptrs, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(a.iters):
        ib.ingest_forcing(ptrs, **kw)
    e1.record(stream)
    e1.synchronize()
us = e0.elapsed_time(e1) * 1e3 / a.iters
ncell = ib.ni * ib.nj
nbytes = 8 * (sum(int(np.prod(v.shape)) for v in args.values()) + ncell           # inputs + the mask
              + T.ENUMS["KID_NFORCING"] * ncell * 3 + 2 * ncell)                  # planes: written, scrubbed (read + write); ua/va read
print(json.dumps({"what": "kid_ingest_forcing + per-cell record pack", "grid": [a.ni, a.nj], "vel": a.vel, "stress": a.stress,
                  "us_per_call": round(us, 2), "algorithmic_MB": round(nbytes / 1e6, 2), "GB_per_s": round(nbytes / us / 1e3, 1)}))
ib.close()

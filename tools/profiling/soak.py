#!/usr/bin/env python3
"""Soak run: many steps of the slow-lane schedule on the global config-2 grid with coasts, periodic re-entry, calving
between steps and trajectory sampling; checks invariants (finite state, no error counts, berg counts add up)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icebergs_amd import synthetic as S  # noqa: E402
from icebergs_amd import types as T  # noqa: E402
from icebergs_amd.distributed import PipelinedStepper  # noqa: E402
from icebergs_amd.framework import Icebergs  # noqa: E402

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
plain = len(sys.argv) > 2 and sys.argv[2] == "plain"   # the serial single-stream schedule, for comparison
n0 = 200_000
grid, p, b = S.config_c2(n=n0, seed=99, continents=True)
p.periodic_reentry = 1
p.current_year = 1
cp = S.calving_params(p)
cap = 2_000_000
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ib = Icebergs(grid, p, capacity=cap)
ib.set_stream(stream.cuda_stream)
ib.set_forcing(grid["forcing"])
ib.set_calving_params(cp)
tp = T.TrajParams()
tp.traj_area_thres, tp.traj_area_thres_fl, tp.save_all_traj_year, tp.save_short_traj, tp.save_fl_traj = 1.0, 1e9, 1e30, 1, 1
ib.set_traj_params(tp)
ib.upload_bergs(b)
dev = torch.device("cuda", 0)
forcing = [torch.from_numpy(np.ascontiguousarray(grid["forcing"][k])).to(dev) for k in T.FORCING_NAMES]
ptrs = [t.data_ptr() for t in forcing]
if plain:
    from icebergs_amd.distributed import ShardedStepper
    _, count = ib.accum_device_ptr()
    acc_t = torch.zeros(count, dtype=torch.float64, device=dev)
    ib.bind_accum_buffer(acc_t.data_ptr(), count)
    st = ShardedStepper(ib, acc_t, ib.ncell, p.diag_mask, None, params=p, resort_interval=12)
else:
    st = PipelinedStepper(ib, p, None, slow_lane=True, resort_interval=12)
calv, hflx = S.coupler_calving(grid, seed=5, frac=0.002, buckets=0.05)
t0 = time.time()
calved = 0
for s in range(nsteps):
    p.current_yearday = s * p.dt / 86400.0
    ib.set_params(p)
    if s % 8 == 0:
        st.flush()
        calved += int(ib.calving(calv, hflx)[T.ENUMS["KID_CS_NBERGS_CALVED"]])
    if s % 48 == 0:
        st.flush()
        ib.record_posn()
    st.set_forcing_device(ptrs)
    st.step()
    if s % 250 == 0:
        st.flush()
        slots, alive = ib.num_bergs()
        print("step %d: slots %d alive %d calved %d traj %d  (%.1f s)" % (s, slots, alive, calved, ib.num_traj_records(), time.time() - t0), flush=True)
st.flush()
torch.cuda.synchronize()
acc, out, scal = ib.fetch()
got = ib.download_bergs()
live = got["alive"] != 0
for name in ("lon", "lat", "uvel", "vvel", "mass", "thickness", "xi", "yj"):
    assert np.isfinite(got[name][live]).all(), name
assert scal[T.ENUMS["KID_S_ERROR_COUNT"]] == 0, scal
melted = int(scal[T.ENUMS["KID_S_NBERGS_MELTED"]])
left = n0 + calved - int(live.sum()) - melted      # bergs that left through the northern / southern edge are removed, not melted
assert 0 <= left, (int(live.sum()), melted, calved)
d = grid["desc"]
assert (got["ine"][live] >= d.isc).all() and (got["ine"][live] <= d.iec).all()
assert np.isfinite(out).all() and np.isfinite(acc).all()
print("soak ok (%s): %d steps, %d alive, %d calved, %d melted, %d left the domain, checksum %.6e, %.1f s" % ("plain" if plain else "slow lane", nsteps, int(live.sum()), calved, melted, left, float(np.sort(got["mass"][live]).sum()), time.time() - t0))
ib.close()

"""Time config 3 at BASELINE size: 1e7 bergs, footloose profile, 2000x1000 Cartesian 1 km grid, dt = 10 s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from icebergs_amd import synthetic as S, types as T
from icebergs_amd.framework import Icebergs
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
t0 = time.time()
grid, p, b = S.config_c3(n=n, seed=3, ni=2000, nj=1000, fl_style="fl_bits", capacity_factor=1.3, dt=10.0, spread=True, displace=True, periodic=True)   # the profile's own namelist: displaced children, periodic channel
print("generated %d bergs in %.1f s" % (n, time.time() - t0), flush=True)
cap = len(b["lon"])
ib = Icebergs(grid, p, capacity=cap, device=0)
ib.upload_bergs(b)
if os.environ.get("KID_C3_NO_ENV_STORE"):   # the fused step does not read the stored environment back: 104 B per berg-step less to write
    ib.set_store_environment(False)
ib.set_resort_interval(int(sys.argv[3]) if len(sys.argv) > 3 else 24)   # re-binning interval (the library's default is 16; 24 is bench.py's choice for this config)
ib.run(2); ib.sync()
t0 = time.time()
ib.run(steps); ib.sync()
dt = time.time() - t0
print("steps %d: %.3f ms/step, %.3e berg-steps/s" % (steps, 1e3 * dt / steps, n * steps / dt), flush=True)
acc, out, scal = ib.fetch()
print("scalars", scal, "n_slots", ib.num_bergs())
ib.close()

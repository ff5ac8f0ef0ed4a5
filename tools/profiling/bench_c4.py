"""Time config 4 at BASELINE size: 224x224 square-packed DEM elements, 90 sub-steps per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from icebergs_amd import synthetic as S
from icebergs_amd.framework import Icebergs
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 224
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
t0 = time.time()
grid, p, b, bd = S.config_c4(nx=nx, ny=nx, hexagonal=False, radius=1500.0, ni=60, nj=60, gridres=20000.0, sub_steps=90,
                             origin=(100137.0, 100211.0), bump=(900.0e3, 440.0e3))
print("generated", len(b["lon"]), "elements", bd["count"].sum(), "bond sides in %.1f s" % (time.time() - t0), flush=True)
ib = Icebergs(grid, p, capacity=len(b["lon"]), device=0)
ib.upload_bergs(b); ib.upload_bonds(bd)
ib.run(1); ib.sync()
t0 = time.time()
ib.run(steps); ib.sync()
dt = time.time() - t0
n = len(b["lon"])
print("steps %d: %.2f ms/step, %.3e element-sub-steps/s, %.3e berg-steps/s" % (steps, 1e3 * dt / steps, n * p.mts_sub_steps * steps / dt, n * steps / dt), flush=True)
acc, out, scal = ib.fetch()
print("scalars", scal)
if len(sys.argv) > 3:
    import oracle_lib
    o = oracle_lib.Oracle(grid, p)
    t0 = time.time(); o.run_step_mts(b, bd, 1); print("oracle 1 step: %.2f s" % (time.time() - t0))
ib.close()
if os.environ.get("KID_DUMP_MAPS"):   # which libraries sit where (to read a native stack trace of the exit path)
    open(os.environ["KID_DUMP_MAPS"], "w").write("".join(l for l in open("/proc/self/maps") if " r-xp " in l))
from icebergs_amd import lib as _kl
_kl.device_reset()   # see lib.device_reset: exit order under rocprofv3 after a cooperative launch

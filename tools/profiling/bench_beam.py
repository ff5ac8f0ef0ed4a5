"""Time the reference's DEM beam tests through the HIP library: bench_beam.py cantilever|supported nsteps [sub_steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icebergs_amd import synthetic as S
from icebergs_amd.framework import Icebergs
kind = sys.argv[1] if len(sys.argv) > 1 else "cantilever"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
grid, p, b, bd = S.config_beam(kind)
if len(sys.argv) > 3: p.mts_sub_steps = int(sys.argv[3])
ib = Icebergs(grid, p, capacity=len(b["lon"]), device=0)
ib.upload_bergs(b); ib.upload_bonds(bd)
ib.run(1); ib.sync()
t0 = time.time(); ib.run(nsteps); ib.sync(); dt = time.time() - t0
print("%s: %d steps x %d sub-steps: %.1f ms/step, %.2f us per sub-step" % (kind, nsteps, p.mts_sub_steps, 1e3 * dt / nsteps, 1e6 * dt / nsteps / p.mts_sub_steps))
ib.close()

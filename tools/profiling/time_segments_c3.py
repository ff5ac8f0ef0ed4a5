"""The segment clock (see time_segments.py) on config 3 at 1e7 bergs: KID_HIP_SO=build_exp/timing.so python tools/profiling/time_segments_c3.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icebergs_amd import synthetic as S, lib as L
from icebergs_amd.framework import Icebergs
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
grid, p, b = S.config_c3(n=n, seed=3, ni=2000, nj=1000, fl_style="fl_bits", capacity_factor=1.3, dt=10.0, spread=True, displace=True, periodic=True)
ib = Icebergs(grid, p, capacity=len(b["lon"]), device=0)
ib.upload_bergs(b)
ib.run(3); ib.sync()
lib = L.load()
out = (C.c_ulonglong * 16)()
lib.kid_exp_timing.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.kid_exp_timing(out, 1)
ib.run(6); ib.sync()
lib.kid_exp_timing(out, 0)
names = ["Verlet: entry (parking reload ...)", "Verlet: position update + adjust_index", "prologue (loads, runs, packets) + first interp", "Verlet: lat terms + accel_pre", "Verlet: accel", "-", "evolve stores", "footloose + second interp", "thermodynamics", "spreading + flush", "final stores (incl. environment)"]
tot = sum(out[1 + q] for q in range(11))
print("waves %d, cycles per wave %.0f" % (out[0], tot / max(out[0], 1)))
for q, nm in enumerate(names):
    if out[1 + q]: print("%-80s %5.1f %%" % (nm, 100.0 * out[1 + q] / tot))
ib.close()

"""The segment clock (see time_segments.py) on the fused sub-step kernel of config 4 (224x224 elements, 90 sub-steps):
KID_HIP_SO=build_exp/timing.so python tools/profiling/time_segments_c4.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icebergs_amd import synthetic as S, lib as L
from icebergs_amd.framework import Icebergs
grid, p, b, bd = S.config_c4(nx=224, ny=224, hexagonal=False, radius=1500.0, ni=60, nj=60, gridres=20000.0, sub_steps=90,
                             origin=(100137.0, 100211.0), bump=(900.0e3, 440.0e3))
ib = Icebergs(grid, p, capacity=len(b["lon"]), device=0)
ib.upload_bergs(b); ib.upload_bonds(bd)
ib.run(3); ib.sync()
lib = L.load()
out = (C.c_ulonglong * 16)()
lib.kid_exp_timing.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.kid_exp_timing(out, 1)
nsteps = 10
ib.run(nsteps); ib.sync()
lib.kid_exp_timing(out, 0)
names = ["waiting for the neighbours' records (poll)", "pair evaluations + sums", "own row (velocity, position)", "drain + record store"]
tot = sum(out[1 + q] for q in range(4))
print("waves %d, cycles per wave and sub-step %.0f" % (out[0], tot / max(out[0], 1) / p.mts_sub_steps))
for q, nm in enumerate(names):
    print("%-46s %5.1f %%" % (nm, 100.0 * out[1 + q] / tot))
ib.close()
from icebergs_amd import lib as _kl
_kl.device_reset()   # see lib.device_reset: exit order under rocprofv3 after a cooperative launch

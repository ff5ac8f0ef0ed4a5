"""Where a wave of the hot build spends its lifetime (config 2 at the bench's population), from a -DKID_EXPERIMENTS
-DKID_EXP_TIMING build of the library:  KID_HIP_SO=build_exp/timing.so python tools/profiling/time_segments.py [bergs]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icebergs_amd import synthetic as S, lib as L
from icebergs_amd.framework import Icebergs
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
grid, p, b = S.config_c2(n=n, seed=2)
ib = Icebergs(grid, p, capacity=len(b["lon"]), device=0)
ib.upload_bergs(b); ib.set_store_environment(False)
age = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # steps without re-binning before the clock starts (how the profile changes as the cell order decays)
if age: ib.set_resort_interval(0)
ib.run(3 + age); ib.sync()
lib = L.load()
out = (C.c_ulonglong * 16)()
lib.kid_exp_timing.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.kid_exp_timing(out, 1)
ib.run(8); ib.sync()
lib.kid_exp_timing(out, 0)
names = ["prologue: rest (parking, accel_pre, first stage setup)", "adjust_index (stages 2-4)", "lat terms", "interp_flds", "accel", "stage tail / sums", "after the loop + stores",
         "second interp", "thermodynamics", "spreading + flush", "final stores",
         "prologue: issue loads, wait for alive / stamp / cell", "prologue: runs, packet loads issued", "prologue: wait for packets and fields"]
tot = sum(out[1 + q] for q in range(14))
print("waves %d (hot + general builds), cycles per wave %.0f" % (out[0], tot / max(out[0], 1)))
for q, nm in sorted(enumerate(names), key=lambda x: (x[0] + 3) % 14 if x[0] in (11, 12, 13) else x[0] + 3 if x[0] > 0 else 3.5):
    print("%-34s %5.1f %%" % (nm, 100.0 * out[1 + q] / tot))
ib.close()

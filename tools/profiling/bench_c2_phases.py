import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from icebergs_amd import synthetic as S
from icebergs_amd.framework import Icebergs
grid, p, b = S.config_c2(n=1000000, seed=2)
ib = Icebergs(grid, p, capacity=len(b["lon"]), device=0)
ib.upload_bergs(b); ib.set_store_environment(False)
ib.run_phases(3); ib.sync()
t0 = time.time(); ib.run_phases(20); ib.sync(); print("phases: %.3f ms/step" % (1e3 * (time.time() - t0) / 20))
t0 = time.time(); ib.run(20); ib.sync(); print("fused: %.3f ms/step" % (1e3 * (time.time() - t0) / 20))
ib.close()

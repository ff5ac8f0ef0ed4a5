#!/usr/bin/env python3
"""Long run of config 3 (footloose profile, displaced children, periodic channel) at 1e6 bergs: many calving events, several
re-binnings; checks at the end that no error was counted, ids are unique, every child drew exactly one counter value and
nobody left the periodic channel.  soak_c3.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icebergs_amd import synthetic as S, types as T
from icebergs_amd.framework import Icebergs
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = 1_000_000
grid, p, b = S.config_c3(n=n, seed=11, ni=640, nj=320, fl_style="new_bergs", capacity_factor=1.6, dt=10.0, spread=True, displace=True, periodic=True)
ib = Icebergs(grid, p, capacity=len(b["lon"]))
ib.upload_bergs(b)
ib.set_store_environment(False)
c0 = ib.get_iceberg_counter().astype(np.int64)
t0 = time.time()
for s in range(0, steps, 50):
    ib.run(min(50, steps - s)); ib.sync()
    print("step", s + 50, "slots/alive", ib.num_bergs(), flush=True)
acc, out, scal = ib.fetch()
a = ib.download_bergs()
alive = a["alive"] != 0
ids = a["id"][alive]
children = int((ids >= (1 << 32)).sum())
c1 = ib.get_iceberg_counter().astype(np.int64)
calved = int(round(scal[T.SCALAR_NAMES["nbergs_calved_fl"]]))
melted = int(round(scal[T.SCALAR_NAMES["nbergs_melted"]]))
err = scal[T.SCALAR_NAMES["error_count"]]
print("children alive", children, "counter diff", int((c1 - c0).sum()), "calved", calved, "melted", melted, "errors", err, "%.1f s" % (time.time() - t0))
assert err == 0.0
assert len(np.unique(ids)) == len(ids)
assert int((c1 - c0).sum()) == calved
assert int(alive.sum()) == n + calved - melted
assert np.all(np.isfinite(a["lon"][alive])) and np.all(np.isfinite(a["mass"][alive]))
print("soak_c3 ok")
ib.close()

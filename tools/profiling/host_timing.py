import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from icebergs_amd import synthetic as S, types as T
from icebergs_amd.framework import Icebergs
grid, p, b = S.config_c2(n=1000000, seed=2)
ib = Icebergs(grid, p, capacity=len(b["lon"]), device=0)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev); torch.cuda.set_stream(st); ib.set_stream(st.cuda_stream)
ib.upload_bergs(b); ib.set_store_environment(False)
forcing_dev = [torch.from_numpy(np.ascontiguousarray(grid["forcing"][name])).to(dev) for name in T.FORCING_NAMES]
ptrs = [t.data_ptr() for t in forcing_dev]
def timeit(name, fn, n=100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-28s host %7.1f us/call   (+%7.1f us/call to drain)" % (name, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t1) / n), flush=True)
for prof in (False, True):
    ib.profile(prof)
    print("profile", prof)
    timeit("set_forcing_device", lambda: ib.set_forcing_device(ptrs))
    timeit("zero_accumulators", lambda: ib._check(ib.lib.kid_zero_accumulators(ib.h), "z"))
    timeit("step_local", ib.step_local)
    timeit("step_gather", ib.step_gather)
    timeit("num_bergs(no alive)", lambda: ib.lib.kid_version())
ev = torch.cuda.Event()
timeit("torch event record", lambda: ev.record(st))
timeit("torch wait_event", lambda: st.wait_event(ev))
s2 = torch.cuda.Stream(dev)
def ctx():
    with torch.cuda.stream(s2): pass
timeit("torch stream ctx", ctx)
ib.close()

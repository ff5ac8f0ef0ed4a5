"""Long run of config 4 (224x224 bonded DEM elements onto a seamount) with the fused sub-step kernel and with the three-launch
graph path (KID_MTS_NO_FUSED=1 in a child process): error counters must stay 0 and the two runs must agree.
  python tools/profiling/soak_c4.py [steps] [nx]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 224
if os.environ.get("KID_SOAK_CHILD"):
    from icebergs_amd import synthetic as S
    from icebergs_amd.framework import Icebergs
    grid, p, b, bd = S.config_c4(nx=nx, ny=nx, hexagonal=False, radius=1500.0, ni=60, nj=60, gridres=20000.0, sub_steps=90,
                                 origin=(100137.0, 100211.0), bump=(450.0e3, 440.0e3), bump_depth=50.0, frac=(float(os.environ.get("KID_SOAK_FRAC_N", "1850")), float(os.environ.get("KID_SOAK_FRAC_T", "1000"))))   # the seamount in the berg's path
    ib = Icebergs(grid, p, capacity=len(b["lon"]), device=0)
    ib.upload_bergs(b); ib.upload_bonds(bd)
    ib.run(steps); ib.sync()
    acc, out, scal = ib.fetch()
    g = ib.download_bergs(); gb = ib.download_bonds(bd["max_bonds"])
    o = np.argsort(g["id"])
    np.savez(os.environ["KID_SOAK_CHILD"], lon=g["lon"][o], lat=g["lat"][o], uvel=g["uvel"][o], rot=g["rot"][o], conglom=g["conglom_id"][o], scal=scal,
             broken=int((gb["broken"] != 0).sum()), nbonds=int(gb["count"].sum()))
    ib.close()
    sys.exit(0)
res = {}
for name, env in (("fused", {}), ("graph", {"KID_MTS_NO_FUSED": "1"})):
    out = "/tmp/soak_c4_%s.npz" % name
    e = dict(os.environ); e.update(env); e["KID_SOAK_CHILD"] = out
    subprocess.check_call([sys.executable, os.path.abspath(__file__), str(steps), str(nx)], env=e)
    res[name] = np.load(out)
f, g = res["fused"], res["graph"]
print("scalars fused", f["scal"], "graph", g["scal"])
print("bond sides %d / %d, broken %d / %d" % (f["nbonds"], g["nbonds"], f["broken"], g["broken"]))
for k in ("lon", "lat", "uvel", "rot"):
    d = np.abs(f[k] - g[k]).max() / max(np.abs(g[k]).max(), 1e-300)
    print("%-5s max rel diff fused vs graph %.2e" % (k, d))
print("conglomerate ids equal:", bool(np.array_equal(f["conglom"], g["conglom"])))

#!/usr/bin/env python3
"""Regression vectors for the two producers in front of the evolve loop (forcing ingest, calving source), frozen from the
CPU oracle like the ones of make_golden.py (NOT reference outputs; see that file's header).

    python tests/golden/make_golden_frontend.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from icebergs_amd import synthetic as S  # noqa: E402
from icebergs_amd import types as T  # noqa: E402

CALVING_BERG_FIELDS = ["lon", "lat", "xi", "yj", "mass", "thickness", "width", "length", "start_lon", "start_lat", "start_day", "start_mass",
                       "mass_scaling", "heat_density", "ine", "jne", "start_year", "id"]


def ingest_case():
    """C-grid velocities with symmetric memory, A-grid stress, Kelvin sst, cyclic, land: two consecutive calls"""
    grid = S.latlon_grid(ni=24, nj=20, dlon=15.0)
    grid["forcing"] = {k: S.zeros(grid["desc"]) for k in T.FORCING_NAMES}
    d = grid["desc"]
    i = (np.arange(d.isd, d.ied + 1)[None, :] - 1) % 24 + 1
    j = np.arange(d.jsd, d.jed + 1)[:, None]
    grid["static"]["msk"][((i >= 5) & (i <= 9) & (j >= 6) & (j <= 10)) | (((i >= 23) | (i <= 1)) & (j >= 13) & (j <= 16))] = 0.0
    kw = dict(vel_stagger="C", stress_stagger="A", cyclic_x=True)
    calls = [S.coupler_forcing(grid, seed=70 + k, vel_stagger="C", stress_stagger="A", symmetric=True, kelvin=(k == 0)) for k in range(2)]
    return grid, calls, kw


def calving_case():
    grid = S.c2_forcing(S.latlon_grid(ni=24, nj=40, dlon=15.0, dlat=4.0))
    p = S.default_params()
    p.current_year, p.current_yearday = 5, 77.5
    cp = S.calving_params(p, tau_calving=3.0e6)
    b = S.place_bergs(grid, 40, 8, (3, 21), (3, 37))
    calls = [S.coupler_calving(grid, seed=k % 2, frac=0.1) for k in range(4)]
    return grid, p, cp, b, calls


def main():
    import oracle_lib as O
    grid, calls, kw = ingest_case()
    orc = O.Oracle(grid, S.default_params())
    planes = None
    for a in calls:
        planes = orc.ingest_forcing(a, planes=planes, **kw)
    path = os.path.join(HERE, "ingest_CA_cyclic.npz")
    np.savez_compressed(path, **planes)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))

    grid, p, cp, b, calls = calving_case()
    orc = O.Oracle(grid, p)
    st = orc.new_calving_state()
    cap = 4000
    bergs = S.empty_bergs(cap)
    n = len(b["lon"])
    for k, v in b.items():
        bergs[k][:n] = v
    bergs["alive"][n:] = 0
    bergs["_n"] = n
    scal = None
    for calv, hflx in calls:
        rc, scal = orc.calving(cp, calv, hflx, st, bergs, cap)
        assert rc == 0
    m = bergs["_n"]
    o = np.argsort(bergs["id"][:m])
    res = {"scalars": scal, "n": np.int64(m)}
    for name in ("calving", "calving_hflx", "stored_ice", "stored_heat", "real_calving", "rmean_calving", "rmean_calving_hflx"):
        res["st_" + name] = st[name]
    for f in CALVING_BERG_FIELDS:
        res["b_" + f] = bergs[f][:m][o]
    path = os.path.join(HERE, "calving_4calls.npz")
    np.savez_compressed(path, **res)
    print("wrote", path, "%d bergs, %.1f KB" % (m, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()

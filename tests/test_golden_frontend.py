"""Frozen vectors for the forcing ingest and the calving source (tests/golden/make_golden_frontend.py): the oracle must still
produce them (CPU), and the HIP library must hit them (GPU)."""
import os
import sys

import numpy as np
import pytest

from icebergs_amd import synthetic as S
from icebergs_amd import types as T
from oracle import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_frontend as G  # noqa: E402

E = T.ENUMS


def _check_calving(res, state, scal, bergs, m):
    for name in ("calving", "calving_hflx", "stored_ice", "stored_heat", "real_calving", "rmean_calving", "rmean_calving_hflx"):
        assert np.array_equal(state[name], res["st_" + name]), name
    assert np.allclose(scal, res["scalars"], rtol=1e-12, atol=0)
    assert m == int(res["n"])
    o = np.argsort(bergs["id"][:m])
    for f in G.CALVING_BERG_FIELDS:
        got, want = bergs[f][:m][o], res["b_" + f]
        if f in ("xi", "yj"):
            assert np.allclose(got, want, rtol=0, atol=1e-12), f
        else:
            assert np.array_equal(got, want), f


def test_oracle_reproduces_frontend_golden():
    grid, calls, kw = G.ingest_case()
    orc = O.Oracle(grid, S.default_params())
    planes = None
    for a in calls:
        planes = orc.ingest_forcing(a, planes=planes, **kw)
    want = np.load(os.path.join(HERE, "golden", "ingest_CA_cyclic.npz"))
    for name in T.FORCING_NAMES:
        assert np.array_equal(planes[name], want[name]), name
    grid, p, cp, b, calls = G.calving_case()
    orc = O.Oracle(grid, p)
    st = orc.new_calving_state()
    cap = 4000
    bergs = S.empty_bergs(cap)
    n = len(b["lon"])
    for k, v in b.items():
        bergs[k][:n] = v
    bergs["alive"][n:] = 0
    bergs["_n"] = n
    for calv, hflx in calls:
        rc, scal = orc.calving(cp, calv, hflx, st, bergs, cap)
    _check_calving(np.load(os.path.join(HERE, "golden", "calving_4calls.npz")), st, scal, bergs, bergs["_n"])


@pytest.mark.gpu
def test_hip_hits_frontend_golden():
    from icebergs_amd.framework import Icebergs
    grid, calls, kw = G.ingest_case()
    ib = Icebergs(grid, S.default_params(), capacity=16)
    for a in calls:
        ib.ingest_forcing(a, **kw)
    got = ib.get_forcing()
    want = np.load(os.path.join(HERE, "golden", "ingest_CA_cyclic.npz"))
    for name in T.FORCING_NAMES:
        assert np.array_equal(got[name], want[name]), name
    ib.close()
    grid, p, cp, b, calls = G.calving_case()
    ib = Icebergs(grid, p, capacity=4000)
    ib.set_forcing(grid["forcing"])
    ib.set_calving_params(cp)
    ib.upload_bergs(b)
    for calv, hflx in calls:
        scal = ib.calving(calv, hflx)
    bergs = ib.download_bergs()
    _check_calving(np.load(os.path.join(HERE, "golden", "calving_4calls.npz")), ib.get_calving_state(), scal, bergs, len(bergs["lon"]))
    ib.close()

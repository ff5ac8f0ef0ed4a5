"""Forcing ingest (icebergs.F90:5236-5383): oracle self-checks on the CPU, HIP-vs-oracle parity on the GPU.

No recorded vector exists for this block (PARITY UNPINNED, see oracle/kid_oracle_ingest.c); the CPU tests pin the
restatement against closed forms written straight from the reference lines they cite."""
import numpy as np
import pytest

from icebergs_amd import synthetic as S
from icebergs_amd import types as T
from oracle import oracle_lib as O

CASES = [  # vel stagger, stress stagger, symmetric memory, stress is a velocity, Kelvin sst, sss present, cyclic, continents
    ("B", "B", False, False, False, True, False, True),
    ("B", "B", False, True, True, False, True, True),
    ("C", "C", False, False, False, True, False, True),
    ("C", "C", True, False, True, True, True, True),
    ("C", "A", False, False, False, False, True, False),
    ("B", "A", False, True, False, True, False, True),
    ("B", "C", True, False, True, True, True, True),
]


def _grid(continents, ni=48, nj=30):
    grid = S.latlon_grid(ni=ni, nj=nj, dlon=360.0 / ni)
    if continents:   # two blocks of land, one across the zonal seam; the halo mask repeats with the period
        d = grid["desc"]
        i = (np.arange(d.isd, d.ied + 1)[None, :] - 1) % ni + 1
        j = np.arange(d.jsd, d.jed + 1)[:, None]
        land = ((i >= 10) & (i <= 20) & (j >= 8) & (j <= 14)) | (((i >= 45) | (i <= 2)) & (j >= 18) & (j <= 24))
        grid["static"]["msk"][land] = 0.0
    grid["forcing"] = {k: S.zeros(grid["desc"]) for k in T.FORCING_NAMES}
    return grid


def _case(c, seed=5):
    vs, ss, sym, tiv, kel, sss, cyc, cont = c
    grid = _grid(cont)
    args = S.coupler_forcing(grid, seed=seed, vel_stagger=vs, stress_stagger=ss, symmetric=sym, kelvin=kel, sss=sss)
    kw = dict(vel_stagger=vs, stress_stagger=ss, tau_is_velocity=tiv, cyclic_x=cyc)
    return grid, args, kw


def test_oracle_bgrid_closed_form():
    """B-grid everything: plain copies into (isc-1:iec+1, jsc-1:jec+1) / (isc:iec, jsc:jec), invert_tau_for_du, scrub"""
    grid, args, kw = _case(CASES[0])
    d = grid["desc"]
    out = O.Oracle(grid, S.default_params()).ingest_forcing(args, **kw)
    msk = grid["static"]["msk"]
    j0, j1, i0, i1 = d.jsc - d.jsd, d.jec - d.jsd + 1, d.isc - d.isd, d.iec - d.isd + 1
    exp = np.zeros_like(msk)
    exp[j0 - 1:j1 + 1, i0 - 1:i1 + 1] = args["uo"]
    exp = np.where((msk < 0.5) | np.isnan(exp), 0.0, exp)
    assert np.array_equal(out["uo"], exp)
    tx, ty = args["tauxa"], args["tauya"]
    with np.errstate(invalid="ignore", divide="ignore"):
        cdd = np.sqrt(0.0015 * np.sqrt(tx * tx + ty * ty))
        ua = np.where(cdd != 0.0, tx / cdd, 0.0)
    ua = np.where((msk[j0:j1, i0:i1] < 0.5) | np.isnan(ua), 0.0, ua)
    assert np.array_equal(out["ua"][j0:j1, i0:i1], ua)
    assert np.all(out["ua"][:j0] == 0) and np.all(out["ua"][:, :i0] == 0)      # a closed domain keeps its (zero) halo
    assert np.array_equal(out["ssh"][j0 - 1:j1 + 1, i0 - 1:i1 + 1], args["ssh"])   # ssh is not masked
    sst = np.where(msk[j0:j1, i0:i1] < 0.5, 0.0, args["sst"])
    assert np.array_equal(out["sst"][j0:j1, i0:i1], sst)
    assert np.count_nonzero(msk < 0.5) > 0 and np.isfinite(np.stack([out[k] for k in T.FORCING_NAMES])).all()


def test_oracle_kelvin_absent_sss_and_wrap():
    grid, args, kw = _case(CASES[1])
    d = grid["desc"]
    out = O.Oracle(grid, S.default_params()).ingest_forcing(args, **kw)
    msk = grid["static"]["msk"]
    j0, j1, i0, i1 = d.jsc - d.jsd, d.jec - d.jsd + 1, d.isc - d.isd, d.iec - d.isd + 1
    wet = msk[j0:j1, i0:i1] > 0.5
    assert np.array_equal(out["sst"][j0:j1, i0:i1][wet], (args["sst"] - 273.15)[wet])
    assert np.all(out["sss"][j0:j1, i0:i1][wet] == -1.0) and np.all(out["sss"][:, :i0] == 0)   # sss is never wrapped
    nic = i1 - i0
    for name in ("uo", "ua", "ssh", "sst", "cn"):   # halo columns hold the column one period away (where the halo cell is wet)
        halo_wet = msk[:, :i0] > 0.5 if name != "ssh" else np.ones_like(msk[:, :i0], dtype=bool)
        src = out[name][:, nic:nic + i0]
        src_ok = (msk[:, nic:nic + i0] > 0.5) | (name == "ssh")
        sel = halo_wet & src_ok
        assert np.array_equal(out[name][:, :i0][sel], src[sel]), name
    assert np.array_equal(out["ua"][j0:j1, i0:i1][wet], np.nan_to_num(args["tauxa"])[wet])   # tau_is_velocity: no inversion


def test_oracle_cgrid_and_agrid_interpolation():
    grid, args, kw = _case(CASES[4])   # C velocities, A stress, cyclic, all ocean
    d = grid["desc"]
    out = O.Oracle(grid, S.default_params()).ingest_forcing(args, **kw)
    j0, j1, i0, i1 = d.jsc - d.jsd, d.jec - d.jsd + 1, d.isc - d.isd, d.iec - d.isd + 1
    nic, njc = i1 - i0, j1 - j0
    # IB:5246-5255 with size(uo) = (nic+2, njc+2): Iu = i - isc + 2 (1-based) -> the array carries one halo cell
    u = args["uo"]
    exp = 0.5 * (u[1:njc + 1, 1:nic + 1] + u[2:njc + 2, 1:nic + 1])   # cells (isc:iec, jsc:jec)
    got = out["uo"][j0:j1, i0:i1]
    ok = ~np.isnan(exp)
    assert np.array_equal(got[ok], exp[ok]) and np.all(got[~ok] == 0)
    v = args["vo"]
    expv = 0.5 * (v[1:njc + 1, 1:nic + 1] + v[1:njc + 1, 2:nic + 2])
    assert np.array_equal(out["vo"][j0:j1, i0:i1], expv)
    # A-grid stress: mean of the four surrounding tracer points of the wrapped, zero-haloed copy (IB:5296-5311)
    t = np.zeros((njc + 2, nic + 2))
    t[1:-1, 1:-1] = args["tauxa"]
    t[1:-1, 0], t[1:-1, -1] = args["tauxa"][:, -1], args["tauxa"][:, 0]
    s = np.zeros_like(t)
    s[1:-1, 1:-1] = args["tauya"]
    s[1:-1, 0], s[1:-1, -1] = args["tauya"][:, -1], args["tauya"][:, 0]
    ax = 0.25 * ((t[1:-1, 1:-1] + t[2:, 2:]) + (t[1:-1, 2:] + t[2:, 1:-1]))
    ay = 0.25 * ((s[1:-1, 1:-1] + s[2:, 2:]) + (s[1:-1, 2:] + s[2:, 1:-1]))
    with np.errstate(invalid="ignore", divide="ignore"):
        cdd = np.sqrt(0.0015 * np.sqrt(ax * ax + ay * ay))
        ua = np.where(cdd != 0.0, ax / cdd, 0.0)
    ua = np.where(np.isnan(ua), 0.0, ua)
    assert np.array_equal(out["ua"][j0:j1, i0:i1], ua)


def test_oracle_refuses_bad_extents_and_keeps_unwritten_cells():
    grid, args, kw = _case(CASES[0])
    orc = O.Oracle(grid, S.default_params())
    bad = dict(args)
    bad["uo"] = args["uo"][:, :-1]
    bad["ui"] = args["ui"][:, :-1]
    assert orc.ingest_forcing(bad, **kw) is None
    before = {k: np.full((orc.nj, orc.ni), 7.0) for k in T.FORCING_NAMES}
    out = orc.ingest_forcing(args, planes=before, **dict(kw, tau_is_velocity=True))
    d = grid["desc"]
    wet_corner = grid["static"]["msk"][0, 0] > 0.5
    assert out["uo"][0, 0] == (7.0 if wet_corner else 0.0)   # outside (isc-1:iec+1, jsc-1:jec+1): previous content, scrubbed by the mask
    assert out["ssh"][0, 0] == 7.0


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s%s%s%s%s%s%s%s" % (c[0], c[1], "s" if c[2] else "", "v" if c[3] else "", "K" if c[4] else "", "S" if c[5] else "", "x" if c[6] else "", "L" if c[7] else ""))
def test_ingest_parity(case):
    """bit-exact: the block is copies, two- and four-point means, one division by a square root and comparisons"""
    from icebergs_amd.framework import Icebergs
    grid, args, kw = _case(case)
    p = S.default_params()
    orc = O.Oracle(grid, p)
    ib = Icebergs(grid, p, capacity=16)
    planes = None
    for step in range(2):   # the second call sees what the first left behind (stale halo content, re-inverted stress)
        a = S.coupler_forcing(grid, seed=11 + step, vel_stagger=case[0], stress_stagger=case[1], symmetric=case[2], kelvin=case[4], sss=case[5])
        planes = orc.ingest_forcing(a, planes=planes, **kw)
        ib.ingest_forcing(a, **kw)
        got = ib.get_forcing()
        for name in T.FORCING_NAMES:
            assert np.array_equal(got[name], planes[name]), (step, name, np.argwhere(got[name] != planes[name])[:4])
    ib.close()


@pytest.mark.gpu
def test_ingest_feeds_the_step():
    """a step after kid_ingest_forcing equals a step after kid_set_forcing with the oracle's ingested planes"""
    from icebergs_amd.framework import Icebergs
    grid, p, b = S.config_c2(n=2000, seed=9)
    args = S.coupler_forcing(grid, seed=3, vel_stagger="C", stress_stagger="A", kelvin=True)
    kw = dict(vel_stagger="C", stress_stagger="A", cyclic_x=True)
    planes = O.Oracle(grid, p).ingest_forcing(args, **kw)
    res = []
    for mode in ("planes", "ingest"):
        ib = Icebergs(grid, p, capacity=len(b["lon"]))
        if mode == "planes":
            ib.set_forcing(planes)
        else:
            ib.ingest_forcing(args, **kw)
        ib.upload_bergs(S.copy_bergs(b))
        ib.run(2)
        acc, out, _ = ib.fetch()
        res.append((ib.download_bergs(), acc.copy(), out.copy()))
        ib.close()
    for name in ("lon", "lat", "uvel", "vvel", "mass"):
        assert np.array_equal(res[0][0][name], res[1][0][name]), name
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert np.abs(res[0][0]["uvel"]).max() > 0


@pytest.mark.gpu
def test_ingest_device_arrays_and_errors():
    import torch
    from icebergs_amd import lib as L
    from icebergs_amd.framework import Icebergs
    grid, args, kw = _case(CASES[3])
    p = S.default_params()
    want = O.Oracle(grid, p).ingest_forcing(args, **kw)
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in args.items()}
    ib = Icebergs(grid, p, capacity=16)
    torch.cuda.synchronize()
    ib.ingest_forcing({k: (t.data_ptr(), tuple(t.shape)) for k, t in dev.items()}, on_device=True, **kw)
    got = ib.get_forcing()
    for name in T.FORCING_NAMES:
        assert np.array_equal(got[name], want[name]), name
    bad = dict(args)
    bad["tauxa"] = args["tauxa"][:-3]
    with pytest.raises(L.KidError):
        ib.ingest_forcing(bad, **kw)
    ib.close()

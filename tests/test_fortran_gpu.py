"""The Fortran host path: the stand-alone Fortran driver (icebergs_amd/fortran/kid_replay.F90) drives the HIP library
through the ISO_C_BINDING module exactly as icebergs_run would (same call sites), and must reproduce the oracle."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from icebergs_amd import synthetic as S
from icebergs_amd import types as T
import parity as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPLAY = os.path.join(ROOT, "icebergs_amd", "fortran", "kid_replay")
MAGIC = 1263093761


def write_case(path, grid, p, b, nsteps, mode):
    n = len(b["lon"])
    with open(path, "wb") as f:
        f.write(struct.pack("<i", MAGIC))
        f.write(bytes(grid["desc"]))
        f.write(bytes(p))
        f.write(struct.pack("<ii", nsteps, mode))
        f.write(struct.pack("<q", n))
        for name in T.GRID_STATIC_NAMES:
            f.write(np.ascontiguousarray(grid["static"][name], dtype=np.float64).tobytes())
        for name in T.FORCING_NAMES:
            f.write(np.ascontiguousarray(grid["forcing"][name], dtype=np.float64).tobytes())
        for name in T.BERG_F64_NAMES:
            f.write(b[name].tobytes())
        for name in T.BERG_I32_NAMES:
            f.write(b[name].tobytes())
        f.write(b["id"].tobytes())


def read_result(path, grid):
    d = grid["desc"]
    ni, nj = d.ied - d.isd + 1, d.jed - d.jsd + 1
    with open(path, "rb") as f:
        n = struct.unpack("<q", f.read(8))[0]
        b = {}
        for name in T.BERG_F64_NAMES:
            b[name] = np.frombuffer(f.read(8 * n), dtype=np.float64).copy()
        for name in T.BERG_I32_NAMES:
            b[name] = np.frombuffer(f.read(4 * n), dtype=np.int32).copy()
        b["id"] = np.frombuffer(f.read(8 * n), dtype=np.int64).copy()
        acc = np.frombuffer(f.read(8 * T.NACC * ni * nj), dtype=np.float64).reshape(T.NACC, nj, ni).copy()
        out = np.frombuffer(f.read(8 * T.NOUT * ni * nj), dtype=np.float64).reshape(T.NOUT, nj, ni).copy()
        scal = np.frombuffer(f.read(8 * T.NSCALAR), dtype=np.float64).copy()
    return b, acc, out, scal


def test_fortran_binding_matches_header():
    """CPU check: the generated Fortran types are up to date with include/kid_types.h."""
    inc = os.path.join(ROOT, "icebergs_amd", "fortran", "kid_types_gen.inc")
    before = open(inc).read()
    subprocess.run(["python3", os.path.join(ROOT, "tools", "gen_fortran_types.py")], check=True, capture_output=True)
    assert open(inc).read() == before, "run tools/gen_fortran_types.py and commit the result"


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_fortran_driver_config1(oracle, tmp_path, mode):
    if not os.path.exists(REPLAY):
        subprocess.run(["make", "-s", "-C", os.path.dirname(REPLAY)], check=True)
    grid, p, b = S.config_c1()
    S.set_diag_all(p)
    nsteps = 72
    case, res = str(tmp_path / "case.bin"), str(tmp_path / "res.bin")
    write_case(case, grid, p, b, nsteps, mode)
    r = subprocess.run([REPLAY, case, res], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    got = read_result(res, grid)
    ref = P.run_oracle(grid, p, b, nsteps)
    P.compare(ref, got, "fortran/C1/mode%d" % mode, params=p)


@pytest.mark.gpu
def test_fortran_driver_config2(oracle, tmp_path):
    if not os.path.exists(REPLAY):
        subprocess.run(["make", "-s", "-C", os.path.dirname(REPLAY)], check=True)
    grid, p, b = S.config_c2(n=20000, seed=31, continents=True)
    case, res = str(tmp_path / "case.bin"), str(tmp_path / "res.bin")
    write_case(case, grid, p, b, 6, 0)
    r = subprocess.run([REPLAY, case, res], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    got = read_result(res, grid)
    ref = P.run_oracle(grid, p, b, 6)
    P.compare(ref, got, "fortran/C2", params=p)


@pytest.mark.gpu
def test_fortran_driver_reports_errors(tmp_path):
    """A bad request must abort with the library's message (the reference's FATAL convention), not continue."""
    grid, p, b = S.config_c1()
    p.tidal_drift = 1.0  # needs FMS's random stream -> KID_EUNSUPPORTED from kid_create
    case, res = str(tmp_path / "case.bin"), str(tmp_path / "res.bin")
    write_case(case, grid, p, b, 1, 0)
    r = subprocess.run([REPLAY, case, res], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "tidal_drift" in (r.stderr + r.stdout)


COUPLE = os.path.join(ROOT, "icebergs_amd", "fortran", "kid_couple")
MAGIC2 = 1263093762


@pytest.mark.gpu
@pytest.mark.parametrize("staggers", [("B", "B"), ("C", "A")])
def test_fortran_coupling_front_end(oracle, tmp_path, staggers):
    """forcing ingest + calving + step through the Fortran module (kid_couple.F90 stands where the coupler stands) must
    reproduce the oracle's ko_ingest_forcing / ko_calving / ko_run_step sequence"""
    import oracle_lib as O
    subprocess.run(["make", "-s", "-C", os.path.dirname(COUPLE)], check=True)
    vs, ss = staggers
    grid = S.c2_forcing(S.latlon_grid(ni=60, nj=200, dlon=6.0))
    p = S.default_params()
    p.current_year, p.current_yearday = 3, 41.5
    b = S.place_bergs(grid, 300, 5, (3, 57), (3, 197))
    d0, st0 = grid["desc"], grid["static"]
    for k in range(3):       # three bergs start in the first halo column east of the domain: the first step deletes them as leavers (FW:3028)
        b["ine"][k] = d0.iec + 1
        b["lon"][k] = b["lon_old"][k] = st0["lon"][b["jne"][k] - d0.jsd, d0.iec - d0.isd] + 3.0
        b["xi"][k] = 0.5
    cp = S.calving_params(p)
    ncalls, cap = 4, 12000
    st_code = {"B": T.ENUMS["KID_BGRID_NE"], "C": T.ENUMS["KID_CGRID_NE"], "A": T.ENUMS["KID_AGRID"]}
    calls = []
    for k in range(ncalls):
        a = S.coupler_forcing(grid, seed=30 + k, vel_stagger=vs, stress_stagger=ss, kelvin=(k % 2 == 0), sss=True)
        a["calving"], a["calving_hflx"] = S.coupler_calving(grid, seed=k % 2, frac=0.04)
        calls.append(a)
    case, res = str(tmp_path / "couple.bin"), str(tmp_path / "couple_res.bin")
    n = len(b["lon"])
    with open(case, "wb") as f:
        f.write(struct.pack("<i", MAGIC2))
        f.write(bytes(grid["desc"])); f.write(bytes(p)); f.write(bytes(cp))
        f.write(struct.pack("<6i", st_code[vs], st_code[ss], 0, 1, 1, ncalls))
        a0 = calls[0]
        f.write(struct.pack("<8i", a0["uo"].shape[1], a0["uo"].shape[0], a0["vo"].shape[1], a0["vo"].shape[0],
                            a0["tauxa"].shape[1], a0["tauxa"].shape[0], a0["tauya"].shape[1], a0["tauya"].shape[0]))
        f.write(struct.pack("<qq", n, cap))
        for name in T.GRID_STATIC_NAMES:
            f.write(np.ascontiguousarray(grid["static"][name], dtype=np.float64).tobytes())
        for name in T.BERG_F64_NAMES:
            f.write(b[name].tobytes())
        for name in T.BERG_I32_NAMES:
            f.write(b[name].tobytes())
        f.write(b["id"].tobytes())
        for a in calls:
            for name in ("uo", "ui", "vo", "vi", "tauxa", "tauya", "ssh", "cn", "hi", "sst", "sss", "calving", "calving_hflx"):
                f.write(np.ascontiguousarray(a[name], dtype=np.float64).tobytes())
    rdir = tmp_path / "RESTART"
    rdir.mkdir()
    r = subprocess.run([COUPLE, case, res, str(rdir)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    # the same sequence on the oracle
    orc = O.Oracle(grid, p)
    stc = orc.new_calving_state()
    bergs = S.empty_bergs(cap)
    for k_, v in b.items():
        bergs[k_][:n] = v
    bergs["alive"][n:] = 0
    bergs["_n"] = n
    planes = None
    for a in calls:
        planes = orc.ingest_forcing(a, vel_stagger=vs, stress_stagger=ss, cyclic_x=True, planes=planes)
        orc.set_forcing(planes)
        rc, rscal = orc.calving(cp, a["calving"], a["calving_hflx"], stc, bergs, cap)
        assert rc == 0
        orc.run_step(bergs, 1)
    d = grid["desc"]
    ni, nj, nk = d.ied - d.isd + 1, d.jed - d.jsd + 1, T.ENUMS["KID_NCLASSES"]
    with open(res, "rb") as f:
        rd = lambda cnt: np.frombuffer(f.read(8 * cnt), dtype=np.float64).copy()
        gplanes = rd(T.ENUMS["KID_NFORCING"] * ni * nj).reshape(-1, nj, ni)
        g_ice = rd(nk * ni * nj).reshape(nk, nj, ni)
        g_heat = rd(ni * nj).reshape(nj, ni)
        g_real = rd(nk * ni * nj).reshape(nk, nj, ni)
        g_calv, g_hflx = rd(ni * nj).reshape(nj, ni), rd(ni * nj).reshape(nj, ni)
        gscal = rd(T.ENUMS["KID_NCALV_SCALARS"])
        nslots = struct.unpack("<q", f.read(8))[0]
        gb = {name: rd(nslots) for name in T.BERG_F64_NAMES}
        for name in T.BERG_I32_NAMES:
            gb[name] = np.frombuffer(f.read(4 * nslots), dtype=np.int32).copy()
        gb["id"] = np.frombuffer(f.read(8 * nslots), dtype=np.int64).copy()
    for k_, name in enumerate(T.FORCING_NAMES):
        assert np.array_equal(gplanes[k_], planes[name]), name
    assert np.array_equal(g_ice, stc["stored_ice"]) and np.array_equal(g_heat, stc["stored_heat"]) and np.array_equal(g_real, stc["real_calving"])
    assert np.array_equal(g_calv, stc["calving"]) and np.array_equal(g_hflx, stc["calving_hflx"])
    assert np.allclose(gscal, rscal, rtol=1e-12, atol=0)
    assert bergs["_n"] > n + 200 and nslots >= int((bergs["alive"][:bergs["_n"]] != 0).sum())
    # surviving bergs, matched by id
    ra = bergs["alive"][:bergs["_n"]] != 0
    ga = gb["alive"] != 0
    ro, go = np.argsort(bergs["id"][:bergs["_n"]][ra]), np.argsort(gb["id"][ga])
    assert np.array_equal(bergs["id"][:bergs["_n"]][ra][ro], gb["id"][ga][go])
    for name in ("lon", "lat", "uvel", "vvel", "mass", "thickness", "heat_density", "start_day"):
        rv, gv = bergs[name][:bergs["_n"]][ra][ro], gb[name][ga][go]
        assert np.allclose(gv, rv, rtol=1e-10, atol=1e-12), (name, float(np.abs(gv - rv).max()))
    # the migration entry points through the Fortran module: the driver packed east + west after the download
    import re
    m = re.search(r"migration: width (\d+) east (\d+) west (\d+)", r.stdout)
    assert m, r.stdout
    out_e = (gb["ine"] > d.iec) & (gb["halo_berg"] < 0.5)
    out_w = (gb["ine"] < d.isc) & (gb["halo_berg"] < 0.5)
    assert (int(m.group(1)), int(m.group(2)), int(m.group(3))) == (34, int(out_e.sum()), int(out_w.sum())), (m.groups(), int(out_e.sum()), int(out_w.sum()))
    assert int(m.group(2)) >= 3                              # the three planted leavers at least
    m2 = re.search(r"first east record ine (\d+) id_ij (-?\d+)", r.stdout)
    if out_e.any():
        assert m2 and int(m2.group(1)) == d.iec + 1 and int(m2.group(2)) in set((gb["id"][out_e] & 0xffffffff).astype(np.int64).tolist() + ((gb["id"][out_e] & 0xffffffff) - (1 << 32)).tolist())
    # the files the Fortran driver wrote from the resident state (write_restart_bergs, write_trajectory), read independently
    from scipy.io import netcdf_file
    with netcdf_file(str(rdir / "icebergs.res.nc"), "r", mmap=False) as f:
        ident = (f.variables["id_cnt"][:].astype(np.int64) << 32) + f.variables["id_ij"][:].astype(np.int64)
        of = np.argsort(ident)
        assert np.array_equal(ident[of], gb["id"][ga][go])
        for name in ("lon", "lat", "mass", "thickness", "heat_density"):
            assert np.array_equal(f.variables[name][:][of], gb[name][ga][go]), name
    with netcdf_file(str(rdir / "calving.res.nc"), "r", mmap=False) as f:
        nic, njc = d.iec - d.isc + 1, d.jec - d.jsc + 1
        sl = (slice(d.jsc - d.jsd, d.jec - d.jsd + 1), slice(d.isc - d.isd, d.iec - d.isd + 1))
        assert np.array_equal(f.variables["stored_ice"][0], g_ice[:, sl[0], sl[1]]) and np.array_equal(f.variables["stored_heat"][0], g_heat[sl])
    with netcdf_file(str(rdir / "iceberg_trajectories.nc"), "r", mmap=False) as f:
        assert f.variables["lon"].shape == (int(ga.sum()),) and list(f.variables)[:6] == ["lon", "lat", "year", "day", "id_cnt", "id_ij"]


GLUE = os.path.join(ROOT, "icebergs_amd", "fortran", "kid_glue_test")
MAGIC3 = 1263093764


@pytest.mark.gpu
@pytest.mark.parametrize("staggers", [("B", "B"), ("C", "A")])
def test_glue_lists_and_icebergs_run(oracle, tmp_path, staggers):
    """icebergs_amd/fortran/kid_icebergs_glue.F90 through its self-contained driver: linked lists of `iceberg` nodes per
    cell are built by sorted insertion (`inorder`, FW:4318-4359) from a SHUFFLED population, flattened in the reference's
    traversal order (cells j outer / i inner, list order: SURVEY A13), stepped four times through the argument list of
    icebergs_run (forcing ingest + calving + hot path + the coupler return of IB:5654-5679), and rebuilt from the device.
    Checked: the flattening order is the reference's; the calving / calving_hflx / mass_berg handed back to the coupler and
    the bergs of the rebuilt lists equal the oracle's; the rebuilt lists are sorted (the driver checks with `inorder`)."""
    import oracle_lib as O
    subprocess.run(["make", "-s", "-C", os.path.dirname(GLUE)], check=True)
    vs, ss = staggers
    grid = S.c2_forcing(S.latlon_grid(ni=60, nj=200, dlon=6.0))
    p = S.default_params()
    p.current_year, p.current_yearday = 3, 41.5
    b = S.place_bergs(grid, 400, 6, (3, 57), (3, 197))
    rng = np.random.default_rng(12)
    n = len(b["lon"])
    b["ine"][:150] = 30                              # long lists: 150 bergs share four cells
    b["jne"][:150] = rng.integers(100, 104, 150)
    d, st = grid["desc"], grid["static"]
    for k in range(150):                             # (keep them inside the cells they were moved to)
        jj, ii = b["jne"][k] - d.jsd, b["ine"][k] - d.isd
        b["xi"][k], b["yj"][k] = rng.uniform(0.2, 0.8, 2)
        b["lon"][k] = st["lon"][jj, ii - 1] + b["xi"][k] * (st["lon"][jj, ii] - st["lon"][jj, ii - 1])
        b["lat"][k] = st["lat"][jj - 1, ii] + b["yj"][k] * (st["lat"][jj, ii] - st["lat"][jj - 1, ii])
        b["lon_old"][k], b["lat_old"][k], b["start_lon"][k], b["start_lat"][k] = b["lon"][k], b["lat"][k], b["lon"][k], b["lat"][k]
    b["start_year"][:] = rng.integers(1, 3, n)
    b["start_day"][:] = np.round(rng.uniform(0, 50, n))       # ties on the second key: the later keys decide
    perm = rng.permutation(n)
    shuffled = {k: (v[perm].copy() if hasattr(v, "dtype") and len(v) == n else v) for k, v in b.items()}
    cp = S.calving_params(p)
    ncalls, cap = 4, 12000
    st_code = {"B": T.ENUMS["KID_BGRID_NE"], "C": T.ENUMS["KID_CGRID_NE"], "A": T.ENUMS["KID_AGRID"]}
    calls = []
    for k in range(ncalls):
        a = S.coupler_forcing(grid, seed=40 + k, vel_stagger=vs, stress_stagger=ss, kelvin=(k % 2 == 1), sss=True)
        a["calving"], a["calving_hflx"] = S.coupler_calving(grid, seed=k % 2, frac=0.04)
        calls.append(a)
    case, res = str(tmp_path / "glue.bin"), str(tmp_path / "glue_res.bin")
    with open(case, "wb") as f:
        f.write(struct.pack("<i", MAGIC3))
        f.write(bytes(grid["desc"])); f.write(bytes(p)); f.write(bytes(cp))
        f.write(struct.pack("<6i", st_code[vs], st_code[ss], 0, 1, 1, ncalls))
        a0 = calls[0]
        f.write(struct.pack("<8i", a0["uo"].shape[1], a0["uo"].shape[0], a0["vo"].shape[1], a0["vo"].shape[0],
                            a0["tauxa"].shape[1], a0["tauxa"].shape[0], a0["tauya"].shape[1], a0["tauya"].shape[0]))
        f.write(struct.pack("<qq", n, cap))
        for name in T.GRID_STATIC_NAMES:
            f.write(np.ascontiguousarray(grid["static"][name], dtype=np.float64).tobytes())
        for name in T.BERG_F64_NAMES:
            f.write(shuffled[name].tobytes())
        for name in T.BERG_I32_NAMES:
            f.write(shuffled[name].tobytes())
        f.write(shuffled["id"].tobytes())
        for a in calls:
            for name in ("uo", "ui", "vo", "vi", "tauxa", "tauya", "ssh", "cn", "hi", "sst", "sss", "calving", "calving_hflx"):
                f.write(np.ascontiguousarray(a[name], dtype=np.float64).tobytes())
    r = subprocess.run([GLUE, case, res], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    # the same sequence on the oracle
    orc = O.Oracle(grid, p)
    stc = orc.new_calving_state()
    bergs = S.empty_bergs(cap)
    for k_, v in b.items():
        bergs[k_][:n] = v
    bergs["alive"][n:] = 0
    bergs["_n"] = n
    planes, want_calv, want_hflx = None, [], []
    nic, njc = d.iec - d.isc + 1, d.jec - d.jsc + 1
    sl = (slice(d.jsc - d.jsd, d.jec - d.jsd + 1), slice(d.isc - d.isd, d.iec - d.isd + 1))
    area = st["area"][sl]
    for a in calls:
        planes = orc.ingest_forcing(a, vel_stagger=vs, stress_stagger=ss, cyclic_x=True, planes=planes)
        orc.set_forcing(planes)
        rc, _ = orc.calving(cp, a["calving"], a["calving_hflx"], stc, bergs, cap)
        assert rc == 0
        orc.run_step(bergs, 1)
        fm = orc.acc[T.ACC_NAMES["floating_melt"]][sl]
        want_calv.append(np.where(area > 0, stc["calving"][sl] / np.where(area > 0, area, 1.0) + fm, 0.0))      # IB:5654-5660
        want_hflx.append(stc["calving_hflx"][sl] + orc.acc[T.ACC_NAMES["calving_hflx"]][sl])                    # IB:3129, 5661
    want_mass = orc.out[T.OUT_NAMES["spread_mass"]][sl]
    with open(res, "rb") as f:
        m0 = struct.unpack("<q", f.read(8))[0]
        flat_ids = np.frombuffer(f.read(8 * m0), dtype=np.int64).copy()
        got_calv, got_hflx = [], []
        for _ in range(ncalls):
            got_calv.append(np.frombuffer(f.read(8 * nic * njc), dtype=np.float64).reshape(njc, nic).copy())
            got_hflx.append(np.frombuffer(f.read(8 * nic * njc), dtype=np.float64).reshape(njc, nic).copy())
        got_mass = np.frombuffer(f.read(8 * nic * njc), dtype=np.float64).reshape(njc, nic).copy()
        m = struct.unpack("<q", f.read(8))[0]
        gb = {name: np.frombuffer(f.read(8 * m), dtype=np.float64).copy() for name in T.BERG_F64_NAMES}
        for name in T.BERG_I32_NAMES:
            gb[name] = np.frombuffer(f.read(4 * m), dtype=np.int32).copy()
        gb["id"] = np.frombuffer(f.read(8 * m), dtype=np.int64).copy()
    # A13: the order the glue flattened the lists in = cells j outer / i inner, `inorder` inside a cell
    want_order = sorted(range(n), key=lambda k: (b["jne"][k], b["ine"][k], b["start_year"][k], b["start_day"][k], b["start_mass"][k], b["start_lon"][k], b["start_lat"][k]))
    assert m0 == n and np.array_equal(flat_ids, b["id"][want_order])
    # ... and the oracle's own notion of that order
    perm2 = np.zeros(n, dtype=np.int64)
    soa = orc.soa({k: (v[:n] if hasattr(v, "dtype") else v) for k, v in b.items() if k != "_n"})
    orc.lib.ko_reference_order(C.byref(soa), perm2.ctypes.data_as(C.POINTER(C.c_int64)))
    assert np.array_equal(b["id"][perm2], flat_ids)
    # what went back to the coupler
    for k in range(ncalls):
        assert np.allclose(got_calv[k], want_calv[k], rtol=1e-9, atol=1e-9 * np.abs(want_calv[k]).max()), k
        assert np.allclose(got_hflx[k], want_hflx[k], rtol=1e-9, atol=1e-9 * max(np.abs(want_hflx[k]).max(), 1e-300)), k
    assert np.allclose(got_mass, want_mass, rtol=1e-9, atol=1e-9 * np.abs(want_mass).max())
    # the rebuilt lists hold the oracle's survivors
    ra = bergs["alive"][:bergs["_n"]] != 0
    assert m == int(ra.sum()) and m > n + 200
    ro, go = np.argsort(bergs["id"][:bergs["_n"]][ra]), np.argsort(gb["id"])
    assert np.array_equal(bergs["id"][:bergs["_n"]][ra][ro], gb["id"][go])
    for name in ("lon", "lat", "uvel", "vvel", "mass", "thickness", "width", "length", "heat_density", "start_day", "xi", "yj"):
        rv, gv = bergs[name][:bergs["_n"]][ra][ro], gb[name][go]
        assert np.allclose(gv, rv, rtol=1e-10, atol=1e-12), (name, float(np.abs(gv - rv).max()))
    # and come out of the lists in traversal order again
    key = list(zip(gb["jne"].tolist(), gb["ine"].tolist(), gb["start_year"].tolist(), gb["start_day"].tolist(), gb["start_mass"].tolist(),
                   gb["start_lon"].tolist(), gb["start_lat"].tolist()))
    assert key == sorted(key)


def write_case_bonded(path, grid, p, b, bd, nsteps):
    """the case file of kid_replay with a bonds section (magic + 1)"""
    write_case(path, grid, p, b, nsteps, 0)
    with open(path, "r+b") as f:
        f.write(struct.pack("<i", MAGIC + 1))
        f.seek(0, 2)
        f.write(struct.pack("<i", bd["max_bonds"]))
        f.write(bd["count"].astype(np.int32).tobytes())
        f.write(bd["other_id"].astype(np.int64).tobytes())
        f.write(bd["broken"].astype(np.int32).tobytes())
        for name in T.BOND_F64_NAMES:
            f.write(np.ascontiguousarray(bd[name], dtype=np.float64).tobytes())


@pytest.mark.gpu
def test_fortran_driver_bonded_cantilever_beam(oracle, tmp_path):
    """The bonded path from the host language: the Fortran driver uploads the bergs and the bond lists of the reference's
    cantilever-beam test (tests/dem_cbeam_test restated, 90 elements, tests/test_beam.py), steps it through kid_run_step --
    which dispatches to the MTS / DEM path -- and downloads bergs and bonds.  40 of the test's 300 steps here (the whole run
    against the analytic lines is test_beam.py's): the beam is bending, no bond has broken, and the state is bit for bit
    what the same calls give from Python."""
    grid, p, b, bd = S.config_beam("cantilever")
    nsteps = 40
    case, res = str(tmp_path / "beam.bin"), str(tmp_path / "beam.out")
    write_case_bonded(case, grid, p, b, bd, nsteps)
    r = subprocess.run([REPLAY, case, res], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bergs=90 alive=90" in r.stdout
    gb, acc, out, scal = read_result(res, grid)
    (pb, _, _, _), pbd = P.run_hip_mts(grid, p, b, bd, nsteps)
    for f in ("lon", "lat", "uvel", "vvel", "rot", "ang_vel"):
        assert np.array_equal(gb[f], pb[f]), f
    tip = np.argmax(b["lon"] + 1e-3 * b["lat"])
    assert gb["lat"][tip] - b["lat"][tip] < -20.0e3           # ~ -36 km after 40 steps, on its way to -48.7 km
    n, mb = 90, bd["max_bonds"]
    with open(res, "rb") as f:
        d = grid["desc"]
        ni, nj = d.ied - d.isd + 1, d.jed - d.jsd + 1
        f.seek(8 + n * (8 * len(T.BERG_F64_NAMES) + 4 * len(T.BERG_I32_NAMES) + 8) + 8 * (T.NACC + T.NOUT) * ni * nj + 8 * T.NSCALAR)
        count = np.frombuffer(f.read(4 * n), dtype=np.int32)
        other = np.frombuffer(f.read(8 * n * mb), dtype=np.int64)
        broken = np.frombuffer(f.read(4 * n * mb), dtype=np.int32)
        length = np.frombuffer(f.read(8 * n * mb), dtype=np.float64)
    assert np.array_equal(count, bd["count"]) and int(count.sum()) == 294
    assert not broken.any()
    live = (np.arange(mb)[:, None] < count[None, :]).ravel()
    assert np.array_equal(other[live], bd["other_id"][live])
    assert np.array_equal(length[live], pbd["length"][live])


# ------------------------------------------------------------------------------------------------------------------------------
# icebergs_init's argument list, the namelist group, bond lists; the multi-GPU pattern from Fortran
# ------------------------------------------------------------------------------------------------------------------------------
INIT = os.path.join(ROOT, "icebergs_amd", "fortran", "kid_init_test")
MULTI = os.path.join(ROOT, "icebergs_amd", "fortran", "kid_multi_test")
MAGIC4 = 1263093765


def icebergs_nml_text(p, desc, halo, **extra):
    """&icebergs_nml as a maintainer would write it, from a kid_params: every variable of the group (tools/icebergs_nml_table.py)
    that kid_params carries under the same name, the two enumerations as their strings, and the grid switches."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from icebergs_nml_table import ICEBERGS_NML
    fields = {f[0].lower(): f[0] for f in T.Params._fields_}
    lines = []
    for name, typ, dim, _ in ICEBERGS_NML:
        key = name.lower()
        if key in extra or key in ("halo", "lx", "grid_is_latlon", "grid_is_regular", "fl_style", "fracture_criterion", "initial_mass", "initial_mass_n"):
            continue
        if key not in fields:
            continue
        v = getattr(p, fields[key])
        if typ == "logical":
            lines.append("  %s = %s" % (name, ".true." if v else ".false."))
        elif typ == "integer":
            lines.append("  %s = %d" % (name, v))
        elif typ == "real" and dim is None:
            lines.append("  %s = %r" % (name, float(v)))
    lines.append("  initial_mass = " + ", ".join(repr(float(x)) for x in p.initial_mass_s))
    lines.append("  initial_mass_n = " + ", ".join(repr(float(x)) for x in p.initial_mass_n))
    lines.append("  separate_distrib_for_n_hemisphere = .true.")
    lines.append("  fl_style = '%s'" % ("fl_bits" if p.fl_style == T.ENUMS["KID_FL_STYLE_FL_BITS"] else "new_bergs"))
    lines.append("  fracture_criterion = '%s'" % ("stress" if p.fracture_criterion_stress else "none"))
    lines.append("  halo = %d" % halo)
    lines.append("  Lx = %r" % float(desc.Lx))
    lines.append("  grid_is_latlon = %s" % (".true." if desc.grid_is_latlon else ".false."))
    lines.append("  grid_is_regular = %s" % (".true." if desc.grid_is_regular else ".false."))
    for k, v in extra.items():
        lines.append("  %s = %s" % (k, v))
    return "&some_other_nml\n  x = 1\n/\n\n&icebergs_nml\n" + "\n".join(lines) + "\n/\n"


def write_init_case(path, gni, gnj, cyclic, nsteps, gridres, dt, sst, sss, cap, b, pairs):
    n = len(b["lon"])
    with open(path, "wb") as f:
        f.write(struct.pack("<5i", MAGIC4, gni, gnj, 2 if cyclic else 0, nsteps))
        f.write(struct.pack("<4d", gridres, dt, sst, sss))
        f.write(struct.pack("<qq", cap, n))
        for name in T.BERG_F64_NAMES:
            f.write(np.ascontiguousarray(b[name], dtype=np.float64).tobytes())
        for name in T.BERG_I32_NAMES:
            f.write(np.ascontiguousarray(b[name], dtype=np.int32).tobytes())
        f.write(np.ascontiguousarray(b["id"], dtype=np.int64).tobytes())
        f.write(struct.pack("<q", len(pairs)))
        for a, o in pairs:
            f.write(struct.pack("<qq", int(a), int(o)))


def read_init_result(path):
    with open(path, "rb") as f:
        d = T.GridDesc.from_buffer_copy(f.read(C.sizeof(T.GridDesc)))
        p = T.Params.from_buffer_copy(f.read(C.sizeof(T.Params)))
        ni, nj = d.ied - d.isd + 1, d.jed - d.jsd + 1
        st = {name: np.frombuffer(f.read(8 * ni * nj), dtype=np.float64).reshape(nj, ni).copy() for name in T.GRID_STATIC_NAMES}
        m = struct.unpack("<q", f.read(8))[0]
        gb = {name: np.frombuffer(f.read(8 * m), dtype=np.float64).copy() for name in T.BERG_F64_NAMES}
        for name in T.BERG_I32_NAMES:
            gb[name] = np.frombuffer(f.read(4 * m), dtype=np.int32).copy()
        gb["id"] = np.frombuffer(f.read(8 * m), dtype=np.int64).copy()
        bonds = []
        for _ in range(m):
            cnt = struct.unpack("<i", f.read(4))[0]
            lst = []
            for _ in range(cnt):
                other_id, other_berg_id, broken, other_ine = struct.unpack("<qqii", f.read(24))
                vals = struct.unpack("<12d", f.read(96))
                lst.append({"other_id": other_id, "other_berg_id": other_berg_id, "broken": broken, "other_berg_ine": other_ine, "f64": vals})
            bonds.append(lst)
        assert f.read() == b""
    return d, p, st, gb, bonds


@pytest.mark.gpu
def test_icebergs_init_and_bond_lists_cantilever(tmp_path):
    """the reference's cantilever test (tests/dem_cbeam_test restated: 90 elements, 294 bond sides) enters
    through icebergs_init's own argument list + &icebergs_nml, lives in per-cell lists of `iceberg` nodes whose bonds are
    `bond` lists made by form_a_bond, is flattened (bond slots in list order), stepped 40 times by kid_icebergs_run and
    rebuilt.  The parameters the namelist reader derives are the ones this package's Python host uses for the same test, the
    state is bit for bit what the Python host gets from the library, every bond list comes back in its slot order and
    connected to its partner node."""
    grid, p, b, bd = S.config_beam("cantilever")
    n, nsteps, mb = len(b["lon"]), 40, bd["max_bonds"]
    (tmp_path / "input.nml").write_text(icebergs_nml_text(p, grid["desc"], halo=3, manually_initialize_bonds_from_radii=".true.", debug=".false."))
    rng = np.random.default_rng(5)
    perm = rng.permutation(n)                                                   # file order is not list order
    sh = {k: (v[perm].copy() if hasattr(v, "dtype") and len(v) == n else v) for k, v in b.items()}
    sh["n_bonds"] = np.zeros(n, dtype=np.int32)                                 # assign_n_bonds is the Fortran side's job
    sh["start_lon"] = np.zeros(n); sh["start_lat"] = np.zeros(n)                # ... and so is dem_tests_init
    # form_a_bond puts a new bond at the head: forming a berg's bonds last slot first leaves the list in slot order
    todo = {k: [int(bd["other_id"][s * n + k]) for s in reversed(range(bd["count"][k]))] for k in range(n)}
    pairs = []
    while todo:                                                                 # ... in any interleaving between bergs
        k = list(todo)[int(rng.integers(len(todo)))]
        pairs.append((int(b["id"][k]), todo[k].pop(0)))
        if not todo[k]:
            del todo[k]
    case, res = str(tmp_path / "init.bin"), str(tmp_path / "init.out")
    write_init_case(case, 20, 20, False, nsteps, 15000.0, p.dt, -1.0, 34.0, n, sh, pairs)
    r = subprocess.run([INIT, case, res], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    d, fp, st, gb, bonds = read_init_result(res)
    # the namelist -> kid_params: every member, but for the clock (the caller's), and the element size the Python generator fills
    # in although constant_interaction_LW is off (the library reads it only when it is on)
    skip = {"current_year", "current_yearday", "pad0", "constant_length", "constant_width"}
    for name, _ in [(f[0], f[1]) for f in T.Params._fields_]:
        if name in skip:
            continue
        a, e = getattr(fp, name), getattr(p, name)
        if hasattr(a, "__len__"):
            assert list(a) == list(e), name
        else:
            assert a == e, (name, a, e)
    assert fp.contact_cells_lon == 1 and fp.contact_cells_lat == 1 and fp.mts_sub_steps == 2000 and fp.explicit_inner_mts == 1
    assert fp.dem_tests_start_lon == b["lon"].min() and fp.dem_tests_end_lon == b["lon"].max()
    # the grid: halo 3 from the namelist, the driver's values on the computational domain, extrapolated corners around it,
    # and -- a closed domain on one PE -- nothing but the ice model's one-cell ring in the halo of the cell metrics
    assert (d.isd, d.ied, d.jsd, d.jed, d.isc, d.iec, d.jsc, d.jec) == (-2, 23, -2, 23, 1, 20, 1, 20) and d.Lx == -1.0
    ii, jj = np.meshgrid(np.arange(d.isd, d.ied + 1), np.arange(d.jsd, d.jed + 1))
    assert np.array_equal(st["lon"], 15000.0 * ii) and np.array_equal(st["lat"], 15000.0 * jj)
    assert np.array_equal(st["lonc"][1:, 1:], 15000.0 * ii[1:, 1:] - 7500.0) and np.array_equal(st["latc"][1:, 1:], 15000.0 * jj[1:, 1:] - 7500.0)
    ring = (ii >= 0) & (ii <= 21) & (jj >= 0) & (jj <= 21)
    comp = (ii >= 1) & (ii <= 20) & (jj >= 1) & (jj <= 20)
    for name, val in (("dx", 15000.0), ("dy", 15000.0), ("msk", 1.0)):
        assert np.array_equal(st[name], np.where(ring, val, 0.0)), name
    assert np.array_equal(st["area"], np.where(comp, 15000.0 ** 2, 0.0)) and np.array_equal(st["ocean_depth"], np.where(comp, 1000.0, 0.0))
    assert np.array_equal(st["cos"], np.ones_like(st["cos"])) and not st["sin"].any()
    # the state: what the Python host gets from the same library for the same test
    (pb, _, _, _), pbd = P.run_hip_mts(grid, p, b, bd, nsteps)
    assert len(gb["id"]) == n
    go, po = np.argsort(gb["id"]), np.argsort(pb["id"])
    for f in ("lon", "lat", "uvel", "vvel", "rot", "ang_vel", "axn", "ayn", "bxn", "byn", "start_lon", "start_lat"):
        assert np.array_equal(gb[f][go], pb[f][po]), f
    assert np.array_equal(gb["n_bonds"][go], b["n_bonds"][np.argsort(b["id"])])
    tip = np.argmax(b["lon"] + 1e-3 * b["lat"])
    assert gb["lat"][go][np.argsort(np.argsort(b["id"]))[tip]] - b["lat"][tip] < -20.0e3
    # the bond lists: slot order, partners connected (connect_all_bonds), the DEM members of every bond
    row_of = {int(i): k for k, i in enumerate(pb["id"])}
    nside = 0
    for k in range(n):
        row = row_of[int(gb["id"][k])]
        assert len(bonds[k]) == bd["count"][row]
        for s, bnd in enumerate(bonds[k]):
            nside += 1
            assert bnd["other_id"] == pbd["other_id"][s * n + row] == bnd["other_berg_id"]
            assert bnd["broken"] == pbd["broken"][s * n + row] == 0
            assert bnd["other_berg_ine"] == gb["ine"][list(gb["id"]).index(bnd["other_id"])]
            for q, name in enumerate(T.BOND_F64_NAMES):
                assert bnd["f64"][q] == pbd[name][s * n + row], (name, k, s)
    assert nside == 294                                                         # 'Total number of bonds is: 294', dem_cbeam_test/input.nml:9


@pytest.mark.gpu
def test_icebergs_init_cyclic_grid_and_defaults(tmp_path):
    """kid_icebergs_init on a zonally periodic channel (dom_x_flags = CYCLIC_GLOBAL_DOMAIN, Lx = ni * gridres, no input.nml
    at all): the namelist defaults of FW:686-822 reach kid_params, the x halo holds the other side's cells moved by whole
    periods (the periodicity fix of FW:1127-1148), the closed y direction keeps what FW:950-960 put there plus the ice
    model's one-cell ring."""
    gni, gnj, gridres = 20, 20, 1000.0
    b = S.empty_bergs(0)
    (tmp_path / "input.nml").write_text("&icebergs_nml\n  Lx = 20000.\n  grid_is_latlon = .false.\n/\n")
    case, res = str(tmp_path / "init.bin"), str(tmp_path / "init.out")
    write_init_case(case, gni, gnj, True, 0, gridres, 1800.0, 0.0, -1.0, 64, b, [])
    r = subprocess.run([INIT, case, res], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    d, fp, st, gb, bonds = read_init_result(res)
    dp = S.default_params()
    for name in [f[0] for f in T.Params._fields_]:
        if name in ("current_year", "current_yearday", "pad0", "periodic_reentry", "max_bonds", "dt", "initial_mass_n"):
            continue
        a, e = getattr(fp, name), getattr(dp, name)
        assert (list(a) == list(e)) if hasattr(a, "__len__") else (a == e), (name, a, e)
    assert fp.periodic_reentry == 1 and fp.max_bonds == 0 and fp.dt == 1800.0   # max_bonds: FW:1263 (bonds off)
    assert list(fp.initial_mass_n) == list(fp.initial_mass_s)                  # FW:1183-1186: no separate northern distribution
    assert (d.isd, d.ied, d.jsd, d.jed) == (-3, 24, -3, 24) and d.Lx == 20000.0 and d.grid_is_latlon == 0   # halo = 4, FW:686
    ii, jj = np.meshgrid(np.arange(d.isd, d.ied + 1), np.arange(d.jsd, d.jed + 1))
    assert np.array_equal(st["lon"], gridres * ii) and np.array_equal(st["lat"], gridres * jj)
    rows = (jj >= 1) & (jj <= 20)
    ring = (jj >= 0) & (jj <= 21) & (ii >= 0) & (ii <= 21)
    for name in ("dx", "dy", "msk"):
        assert np.array_equal(st[name], np.where(rows | ring, 1.0 if name == "msk" else gridres, 0.0)), name
    assert np.array_equal(st["area"], np.where(rows, gridres ** 2, 0.0)) and np.array_equal(st["ocean_depth"], np.where(rows, 1000.0, 0.0))


@pytest.mark.gpu
def test_fortran_two_handles_summed_by_hand(oracle, tmp_path):
    """INTEGRATION.md section 6 from the reference's host language: two handles on one GPU play two MPI ranks that share a
    grid (kid_icebergs_run_local -> sum of the live prefix of the accumulator block, done by hand through the host ->
    kid_icebergs_run_finish).  Both ranks hand the coupler the same fields, bit for bit; those fields and the union of the two
    ranks' bergs are what the oracle gives for the whole population on one grid."""
    import oracle_lib as O
    vs, ss = "B", "B"
    grid, p, _ = S.config_c2(n=10, seed=5)
    p.add_weight_to_ocean, p.pass_fields_to_ocean_model = 1, 1
    b = S.place_bergs(grid, 600, 9, (3, 357), (3, 197))
    n = len(b["lon"])
    rng = np.random.default_rng(3)
    perm = rng.permutation(n)
    shuffled = {k: (v[perm].copy() if hasattr(v, "dtype") and len(v) == n else v) for k, v in b.items()}
    cp = S.calving_params(p)
    ncalls, cap = 5, 1024
    st_code = {"B": T.ENUMS["KID_BGRID_NE"]}
    calls = [S.coupler_forcing(grid, seed=60 + k, vel_stagger=vs, stress_stagger=ss, kelvin=False, sss=True) for k in range(ncalls)]
    d, st = grid["desc"], grid["static"]
    nic, njc = d.iec - d.isc + 1, d.jec - d.jsc + 1
    case, res = str(tmp_path / "multi.bin"), str(tmp_path / "multi.out")
    with open(case, "wb") as f:
        f.write(struct.pack("<i", MAGIC3))
        f.write(bytes(d)); f.write(bytes(p)); f.write(bytes(cp))
        f.write(struct.pack("<6i", st_code[vs], st_code[ss], 0, 1, 1, ncalls))
        a0 = calls[0]
        f.write(struct.pack("<8i", a0["uo"].shape[1], a0["uo"].shape[0], a0["vo"].shape[1], a0["vo"].shape[0],
                            a0["tauxa"].shape[1], a0["tauxa"].shape[0], a0["tauya"].shape[1], a0["tauya"].shape[0]))
        f.write(struct.pack("<qq", n, cap))
        for name in T.GRID_STATIC_NAMES:
            f.write(np.ascontiguousarray(st[name], dtype=np.float64).tobytes())
        for name in T.BERG_F64_NAMES:
            f.write(shuffled[name].tobytes())
        for name in T.BERG_I32_NAMES:
            f.write(shuffled[name].tobytes())
        f.write(shuffled["id"].tobytes())
        zero = np.zeros((njc, nic))
        for a in calls:
            for name in ("uo", "ui", "vo", "vi", "tauxa", "tauya", "ssh", "cn", "hi", "sst", "sss"):
                f.write(np.ascontiguousarray(a[name], dtype=np.float64).tobytes())
            f.write(zero.tobytes()); f.write(zero.tobytes())
    r = subprocess.run([MULTI, case, res], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    # the whole population on the oracle
    orc = O.Oracle(grid, p)
    bergs = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in b.items()}
    sl = (slice(d.jsc - d.jsd, d.jec - d.jsd + 1), slice(d.isc - d.isd, d.iec - d.isd + 1))
    planes, want_calv, want_hflx = None, [], []
    for a in calls:
        planes = orc.ingest_forcing(a, vel_stagger=vs, stress_stagger=ss, cyclic_x=True, planes=planes)
        orc.set_forcing(planes)
        orc.run_step(bergs, 1)
        want_calv.append(orc.acc[T.ACC_NAMES["floating_melt"]][sl].copy())
        want_hflx.append(orc.acc[T.ACC_NAMES["calving_hflx"]][sl].copy())
    want_mass = orc.out[T.OUT_NAMES["spread_mass"]][sl]
    with open(res, "rb") as f:
        live, na, nb_ = struct.unpack("<qqq", f.read(24))
        got = []
        for _ in range(ncalls):
            got.append([np.frombuffer(f.read(8 * nic * njc), dtype=np.float64).reshape(njc, nic).copy() for _ in range(4)])
        mass_a = np.frombuffer(f.read(8 * nic * njc), dtype=np.float64).reshape(njc, nic).copy()
        mass_b = np.frombuffer(f.read(8 * nic * njc), dtype=np.float64).reshape(njc, nic).copy()
        ranks = []
        for _ in range(2):
            m = struct.unpack("<q", f.read(8))[0]
            gb = {name: np.frombuffer(f.read(8 * m), dtype=np.float64).copy() for name in T.BERG_F64_NAMES}
            for name in T.BERG_I32_NAMES:
                gb[name] = np.frombuffer(f.read(4 * m), dtype=np.int32).copy()
            gb["id"] = np.frombuffer(f.read(8 * m), dtype=np.int64).copy()
            ranks.append(gb)
        assert f.read() == b""
    assert na + nb_ == n and abs(na - nb_) <= 1
    ncell = (d.ied - d.isd + 1) * (d.jed - d.jsd + 1)
    from icebergs_amd.distributed import accumulator_views
    assert live == len(accumulator_views(np.zeros(T.NSCALAR + T.NACC * ncell), ncell, p.diag_mask, p)[0])   # Fortran and Python hosts sum the same prefix
    for k in range(ncalls):
        ca, ha, cb, hb = got[k]
        assert np.array_equal(ca, cb) and np.array_equal(ha, hb), k             # both ranks: the same coupler return
        assert np.abs(ca).max() > 0
        assert np.allclose(ca, want_calv[k], rtol=1e-9, atol=1e-9 * np.abs(want_calv[k]).max()), k
        assert np.allclose(ha, want_hflx[k], rtol=1e-9, atol=1e-9 * max(np.abs(want_hflx[k]).max(), 1e-300)), k
    assert np.array_equal(mass_a, mass_b) and np.abs(mass_a).max() > 0
    assert np.allclose(mass_a, want_mass, rtol=1e-9, atol=1e-9 * np.abs(want_mass).max())
    ids = np.concatenate([ranks[0]["id"], ranks[1]["id"]])
    alive = bergs["alive"] != 0
    assert len(set(ids.tolist())) == len(ids) == int(alive.sum())
    go, ro = np.argsort(ids), np.argsort(bergs["id"][alive])
    for name in ("lon", "lat", "uvel", "vvel", "mass", "thickness", "width", "length"):
        gv = np.concatenate([ranks[0][name], ranks[1][name]])[go]
        rv = bergs[name][alive][ro]
        assert np.allclose(gv, rv, rtol=1e-10, atol=1e-12), (name, float(np.abs(gv - rv).max()))


@pytest.mark.gpu
def test_icebergs_init_derived_parameters(tmp_path):
    """What ice_bergs_framework_init derives from the namelist, through kid_icebergs_init: the MTS sub-step count from the spring
    constant (FW:1296-1301: ceiling(dt / (0.3 / sqrt(spring_coef)))), contact_cells_lon / lat from contact_distance on the grid
    (FW:1492-1519), contact_spring_coef defaulting to spring_coef (FW:1313), Verlet forced by mts (FW:1305-1308), the halo raised
    for rotating bonded bergs (FW:1243-1246), the scaled fracture thresholds (FW:1355-1356), explicit_inner_mts forced by dem
    (FW:1436), old_interp_flds_order off with mts (FW:1483)."""
    gni, gnj, gridres = 20, 20, 1000.0
    nml = """&icebergs_nml
  grid_is_latlon = .false.
  Lx = -1.
  halo = 1
  mts = .true.
  dem = .true.
  mts_sub_steps = -1
  spring_coef = 1.e-4
  iceberg_bonds_on = .true.
  interactive_icebergs_on = .true.
  max_bonds = 4
  contact_distance = 2500.
  frac_thres_scaling = 2.5
  frac_thres_n = 4.
  frac_thres_t = 8.
  fracture_criterion = 'stress'
  Runge_not_Verlet = .true.
  set_melt_rates_to_zero = .true.
/
"""
    (tmp_path / "input.nml").write_text(nml)
    case, res = str(tmp_path / "init.bin"), str(tmp_path / "init.out")
    write_init_case(case, gni, gnj, False, 0, gridres, 1800.0, 0.0, -1.0, 64, S.empty_bergs(0), [])
    r = subprocess.run([INIT, case, res], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    d, fp, st, gb, bonds = read_init_result(res)
    assert fp.mts == 1 and fp.dem == 1 and fp.Runge_not_Verlet == 0 and fp.old_interp_flds_order == 0 and fp.explicit_inner_mts == 1
    assert fp.mts_sub_steps == int(np.ceil(1800.0 / (0.3 / np.sqrt(1.0e-4)))) == 60
    assert fp.contact_spring_coef == fp.spring_coef == 1.0e-4
    assert fp.contact_cells_lon == 3 and fp.contact_cells_lat == 3          # 2500 m on 1 km cells
    assert fp.max_bonds == 4 and fp.iceberg_bonds_on == 1 and fp.fracture_criterion_stress == 1
    assert fp.frac_thres_n == 10.0 and fp.frac_thres_t == 20.0
    assert (d.isd, d.ied) == (-2, 23)                                        # halo 1 -> 3: rotate_icebergs_for_mass_spreading with bonds

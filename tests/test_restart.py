"""Restart files straight from the structure of arrays (icebergs_fms2io.F90:124-631, 663-1049).

The independent checker for the file format is scipy.io.netcdf_file (its own implementation of netCDF classic): what
the library writes must read back there with the reference's variable names, types and attributes, and a file written
there (as FMS/libnetcdf would in classic mode) must load through the library."""
import ctypes as C
import os

import numpy as np
import pytest
from scipy.io import netcdf_file

from icebergs_amd import lib as L
from icebergs_amd import synthetic as S
from icebergs_amd import types as T

INT_VARS = ("ine", "jne", "start_year", "id_cnt", "id_ij", "i")
BASE_VARS = ["lon", "lat", "uvel", "vvel", "mass", "ine", "jne", "thickness", "width", "length", "start_lon", "start_lat", "start_year",
             "id_cnt", "id_ij", "start_day", "start_mass", "mass_scaling", "mass_of_bits", "heat_density"]


def _soa(b, n):
    s = T.BergSoA()
    s.n = n
    for k, name in enumerate(T.BERG_F64_NAMES):
        s.f64[k] = b[name].ctypes.data_as(C.POINTER(C.c_double))
    for k, name in enumerate(T.BERG_I32_NAMES):
        s.i32[k] = b[name].ctypes.data_as(C.POINTER(C.c_int32))
    s.id = b["id"].ctypes.data_as(C.POINTER(C.c_int64))
    return s


def _bergs(n=257, seed=3):
    grid, p, b = S.config_c2(n=n, seed=seed)
    rng = np.random.default_rng(seed)
    for name in ("uvel", "vvel", "axn", "ayn", "bxn", "byn", "heat_density", "mass_of_bits", "start_lon", "start_lat", "start_day", "start_mass", "fl_k"):
        b[name][:] = rng.normal(0, 1, n)
    b["start_year"][:] = rng.integers(1900, 2100, n)
    b["id"][:] = (rng.integers(1, 1 << 20, n).astype(np.int64) << 32) + rng.integers(1, 72000, n)
    b["alive"][::17] = 0          # dead bergs do not reach the file
    return grid, p, b


def test_written_file_reads_in_scipy(tmp_path):
    lib = L.load()
    grid, p, b = _bergs()
    p.Runge_not_Verlet = 0        # Verlet: the four accelerations are part of the file (IO2:326-335)
    path = str(tmp_path / "icebergs.res.nc")
    assert lib.kid_restart_write_bergs(path.encode(), C.byref(p), C.byref(_soa(b, len(b["lon"])))) == 0
    live = b["alive"] != 0
    with netcdf_file(path, "r", mmap=False) as f:
        assert f.version_byte == 2 and f.dimensions["i"] is None                 # 64-bit offset, "i" unlimited
        assert f.file_format_major_version == 0 and f.file_format_minor_version == 1 and f.time_axis == 0
        want = BASE_VARS[:5] + ["axn", "ayn", "bxn", "byn"] + BASE_VARS[5:] + ["i"]
        assert list(f.variables) == want                                            # the reference's order of registration
        for name in want:
            v = f.variables[name]
            assert v.dimensions == ("i",) and v.shape == (int(live.sum()),)
            assert v.typecode() == ("i" if name in INT_VARS else "d"), name
        assert f.variables["lon"].long_name == b"longitude" and f.variables["lon"].units == b"degrees_E"
        assert f.variables["id_ij"].long_name == b"position component of iceberg id"
        for name in ("lon", "lat", "uvel", "mass", "axn", "byn", "heat_density", "start_day", "mass_scaling"):
            assert np.array_equal(f.variables[name][:], b[name][live]), name
        assert np.array_equal(f.variables["ine"][:], b["ine"][live]) and np.array_equal(f.variables["start_year"][:], b["start_year"][live])
        ident = (f.variables["id_cnt"][:].astype(np.int64) << 32) + f.variables["id_ij"][:].astype(np.int64)
        assert np.array_equal(ident, b["id"][live])
        assert np.array_equal(f.variables["i"][:], np.arange(1, live.sum() + 1))
    p.Runge_not_Verlet, p.footloose, p.mts, p.dem = 1, 1, 1, 1   # optional groups (IO2:366-385); no accelerations for RK4
    b["static_berg"][5] = 1.0
    assert lib.kid_restart_write_bergs(path.encode(), C.byref(p), C.byref(_soa(b, len(b["lon"])))) == 0
    with netcdf_file(path, "r", mmap=False) as f:
        names = list(f.variables)
        assert "axn" not in names and names[-2:] == ["static_berg", "i"]
        for name in ("fl_k", "mass_of_fl_bits", "mass_of_fl_bergy_bits", "axn_fast", "byn_fast", "ang_vel", "ang_accel", "rot"):
            assert name in names
        assert f.variables["static_berg"][:].sum() == 1.0


@pytest.mark.parametrize("version", [1, 2])
def test_file_written_by_scipy_loads(tmp_path, version):
    """a classic file from another writer (variables in another order, a float variable, an extra one): CDF-1 and CDF-2"""
    lib = L.load()
    grid, p, b = _bergs(n=100, seed=9)
    n = 100
    path = str(tmp_path / "in.nc")
    with netcdf_file(path, "w", version=version) as f:
        f.createDimension("i", None)
        order = ["mass", "lat", "lon", "id_ij", "id_cnt", "jne", "ine", "uvel", "vvel", "thickness", "width", "length", "start_lon", "start_lat",
                 "start_year", "start_day", "start_mass", "mass_scaling", "mass_of_bits", "heat_density", "axn", "bxn"]
        for name in order:
            if name == "id_cnt":
                v = f.createVariable(name, "i", ("i",)); v[:] = (b["id"] >> 32).astype(np.int32)
            elif name == "id_ij":
                v = f.createVariable(name, "i", ("i",)); v[:] = (b["id"] & 0xFFFFFFFF).astype(np.int32)
            elif name in INT_VARS:
                v = f.createVariable(name, "i", ("i",)); v[:] = b[name]
            elif name == "width":
                v = f.createVariable(name, "f", ("i",)); v[:] = b[name].astype(np.float32)   # a single-precision writer
            else:
                v = f.createVariable(name, "d", ("i",)); v[:] = b[name]
        v = f.createVariable("unrelated", "d", ("i",)); v[:] = np.ones(n)
    cnt = C.c_int64()
    assert lib.kid_restart_count_bergs(path.encode(), C.byref(cnt)) == 0 and cnt.value == n
    out = S.empty_bergs(n)
    out["alive"][:] = 0
    soa = _soa(out, 0)
    assert lib.kid_restart_read_bergs(path.encode(), C.byref(soa), n - 1) == -4          # KID_ECAPACITY
    assert lib.kid_restart_read_bergs(path.encode(), C.byref(soa), n) == 0 and soa.n == n
    for name in ("lon", "lat", "uvel", "vvel", "mass", "thickness", "length", "start_lon", "start_day", "mass_scaling", "heat_density", "axn", "bxn"):
        assert np.array_equal(out[name], b[name]), name
    assert np.array_equal(out["width"], b["width"].astype(np.float32).astype(np.float64))
    assert np.array_equal(out["id"], b["id"]) and np.array_equal(out["ine"], b["ine"]) and np.array_equal(out["start_year"], b["start_year"])
    assert np.array_equal(out["uvel_old"], b["uvel"]) and np.array_equal(out["lon_old"], b["lon"]) and np.all(out["alive"] == 1)   # IO2:901-904
    assert not out["ayn"].any() and not out["halo_berg"].any()                          # absent from the file -> zero
    open(str(tmp_path / "hdf5.nc"), "wb").write(b"\\x89HDF\\r\\n\\x1a\\n" + b"\\0" * 64)
    assert lib.kid_restart_count_bergs(str(tmp_path / "hdf5.nc").encode(), C.byref(cnt)) == -1


def test_round_trip_on_the_host(tmp_path):
    lib = L.load()
    grid, p, b = _bergs(n=333, seed=5)
    path = str(tmp_path / "rt.nc")
    assert lib.kid_restart_write_bergs(path.encode(), C.byref(p), C.byref(_soa(b, 333))) == 0
    live = b["alive"] != 0
    out = S.empty_bergs(333)
    soa = _soa(out, 0)
    assert lib.kid_restart_read_bergs(path.encode(), C.byref(soa), 333) == 0 and soa.n == live.sum()
    m = int(live.sum())
    for name in ("lon", "lat", "uvel", "vvel", "mass", "thickness", "width", "length", "heat_density", "mass_scaling", "start_mass"):
        assert np.array_equal(out[name][:m], b[name][live]), name
    assert np.array_equal(out["id"][:m], b["id"][live])


@pytest.mark.gpu
def test_restart_round_trip_continues_the_run(tmp_path):
    """run, write the restart files, load them into a fresh handle: the restored run continues like the original
    (xi, yj are recomputed from lon / lat, so to rounding), the calving buckets and id counters come back bit for bit"""
    from icebergs_amd.framework import Icebergs
    grid, p, b = S.config_c2(n=3000, seed=12)
    grid = S.c2_forcing(S.latlon_grid(ni=60, nj=200, dlon=6.0))
    b = S.place_bergs(grid, 3000, 12, (3, 57), (3, 197))
    p.current_year, p.current_yearday = 11, 200.5
    cp = S.calving_params(p, tau_calving=3.0e6)
    cap = 30000

    def fresh():
        ib = Icebergs(grid, p, capacity=cap)
        ib.set_forcing(grid["forcing"])
        ib.set_calving_params(cp)
        return ib
    a = fresh()
    a.upload_bergs(b)
    for k in range(5):
        calv, hflx = S.coupler_calving(grid, seed=k % 2, frac=0.04)
        a.calving(calv, hflx)
        a.run(1)
    a.write_restart(tmp_path)
    for name in ("icebergs.res.nc", "calving.res.nc"):
        assert os.path.exists(tmp_path / name)
    with netcdf_file(str(tmp_path / "calving.res.nc"), "r", mmap=False) as f:
        d = grid["desc"]
        nic, njc = d.iec - d.isc + 1, d.jec - d.jsc + 1
        assert f.variables["stored_ice"].shape == (1, 10, njc, nic) and f.variables["stored_heat"].shape == (1, njc, nic)
        assert f.variables["stored_ice"].dimensions == ("Time", "zaxis_1", "yaxis_1", "xaxis_1")
        st = a.get_calving_state()
        sl = (slice(d.jsc - d.jsd, d.jec - d.jsd + 1), slice(d.isc - d.isd, d.iec - d.isd + 1))
        assert np.array_equal(f.variables["stored_ice"][0], st["stored_ice"][:, sl[0], sl[1]])
        assert np.array_equal(f.variables["rmean_calving"][0], st["rmean_calving"][sl])
        assert np.array_equal(f.variables["iceberg_counter_grd"][0], a.get_iceberg_counter()[sl])
    r = fresh()
    r.read_restart(tmp_path)
    ba, br = a.download_bergs(), r.download_bergs()
    la = ba["alive"] != 0
    oa, orr = np.argsort(ba["id"][la]), np.argsort(br["id"])
    assert np.array_equal(ba["id"][la][oa], br["id"][orr]) and len(br["id"]) > 3500
    for name in ("lon", "lat", "uvel", "vvel", "mass", "thickness", "width", "length", "heat_density", "mass_scaling", "start_day", "mass_of_bits"):
        assert np.array_equal(ba[name][la][oa], br[name][orr]), name
    for name in ("xi", "yj"):
        assert np.allclose(ba[name][la][oa], br[name][orr], rtol=0, atol=1e-9), name
    sa, sr = a.get_calving_state(), r.get_calving_state()
    for name in ("stored_ice", "stored_heat", "rmean_calving", "rmean_calving_hflx"):
        assert np.array_equal(sa[name], sr[name]), name
    assert np.array_equal(a.get_iceberg_counter(), r.get_iceberg_counter())
    for ib in (a, r):   # both go on for three more steps
        for k in range(3):
            calv, hflx = S.coupler_calving(grid, seed=(k + 1) % 2, frac=0.04)
            ib.calving(calv, hflx)
            ib.run(1)
    ba, br = a.download_bergs(), r.download_bergs()
    la, lr = ba["alive"] != 0, br["alive"] != 0
    oa, orr = np.argsort(ba["id"][la]), np.argsort(br["id"][lr])
    assert np.array_equal(ba["id"][la][oa], br["id"][lr][orr])
    for name in ("lon", "lat", "uvel", "vvel", "mass", "thickness"):
        assert np.allclose(ba[name][la][oa], br[name][lr][orr], rtol=1e-9, atol=1e-9), name
    a.close()
    r.close()


def _bond_soa(bd, n):
    s = T.BondSoA()
    s.n, s.max_bonds = n, int(bd["max_bonds"])
    s.count = bd["count"].ctypes.data_as(C.POINTER(C.c_int32))
    s.other_id = bd["other_id"].ctypes.data_as(C.POINTER(C.c_int64))
    s.broken = bd["broken"].ctypes.data_as(C.POINTER(C.c_int32))
    for k, name in enumerate(T.BOND_F64_NAMES):
        s.f64[k] = bd[name].ctypes.data_as(C.POINTER(C.c_double))
    return s


def test_bonds_file_on_the_host(tmp_path):
    """bonds_iceberg.res.nc (IO2:466-583): one record per bond side, readable by scipy; reading puts every bond at the
    head of its berg's list (form_a_bond), i.e. the slots come back in reverse"""
    lib = L.load()
    grid, p, b, bd = S.config_c4(nx=4, ny=5)
    n = len(b["lon"])
    rng = np.random.default_rng(1)
    for name in ("tangd1", "tangd2", "nstress", "sstress", "rel_rotation"):
        bd[name][:] = rng.normal(0, 1, bd[name].shape)
    bd["broken"][::7] = 1
    path = str(tmp_path / "bonds_iceberg.res.nc")
    assert lib.kid_restart_write_bonds(path.encode(), C.byref(p), C.byref(_soa(b, n)), C.byref(_bond_soa(bd, n))) == 0
    nb = int(bd["count"].sum())
    with netcdf_file(path, "r", mmap=False) as f:
        assert list(f.variables)[:8] == ["first_berg_ine", "first_berg_jne", "first_id_cnt", "first_id_ij", "other_berg_ine", "other_berg_jne", "other_id_cnt", "other_id_ij"]
        assert list(f.variables)[8:] == ["tangd1", "tangd2", "nstress", "sstress", "rel_rotation", "broken"]
        assert f.variables["first_id_ij"].shape == (nb,) and f.variables["broken"].typecode() == "d"
        first = (f.variables["first_id_cnt"][:].astype(np.int64) << 32) + f.variables["first_id_ij"][:]
        other = (f.variables["other_id_cnt"][:].astype(np.int64) << 32) + f.variables["other_id_ij"][:]
        row = {int(i): k for k, i in enumerate(b["id"])}
        assert all(b["ine"][row[int(o)]] == v for o, v in zip(other, f.variables["other_berg_ine"][:]))
        k0 = int(np.argmax(bd["count"]))       # a berg with the most bonds: its records follow its slots
        mine = np.nonzero(first == b["id"][k0])[0]
        assert list(other[mine]) == [int(bd["other_id"][s * n + k0]) for s in range(bd["count"][k0])]
        assert list(f.variables["tangd1"][:][mine]) == [bd["tangd1"][s * n + k0] for s in range(bd["count"][k0])]
    back = S.empty_bonds(n, int(bd["max_bonds"]))
    assert lib.kid_restart_read_bonds(path.encode(), C.byref(_soa(b, n)), C.byref(_bond_soa(back, n))) == 0
    assert np.array_equal(back["count"], bd["count"])
    for k in range(n):
        c = int(bd["count"][k])
        for s in range(c):
            r = c - 1 - s
            assert back["other_id"][s * n + k] == bd["other_id"][r * n + k]
            assert back["broken"][s * n + k] == bd["broken"][r * n + k]
            assert back["sstress"][s * n + k] == bd["sstress"][r * n + k]
    small = S.empty_bonds(n, 1)
    assert lib.kid_restart_read_bonds(path.encode(), C.byref(_soa(b, n)), C.byref(_bond_soa(small, n))) == -4


@pytest.mark.gpu
def test_bonded_restart_round_trip(tmp_path):
    """config-4 family: a bonded DEM conglomerate is written mid-run and restored; same bonds (as a set, with their state),
    and the restored run goes on like the original"""
    import parity as P
    from icebergs_amd.framework import Icebergs
    grid, p, b, bd = S.config_c4(nx=5, ny=7)
    n = len(b["lon"])
    a = Icebergs(grid, p, capacity=n)
    a.upload_bergs(b)
    a.upload_bonds(bd)
    a.run(3)
    a.write_restart(tmp_path)
    assert os.path.exists(tmp_path / "bonds_iceberg.res.nc")
    r = Icebergs(grid, p, capacity=n)
    r.set_forcing(grid["forcing"])
    r.read_restart(tmp_path)
    ba, br = a.download_bergs(), r.download_bergs()
    bda, bdr = a.download_bonds(bd["max_bonds"]), r.download_bonds(bd["max_bonds"])
    assert np.array_equal(ba["id"], br["id"])        # rows of a bonded population keep their order
    sa, sr = P.bond_set(ba, bda), P.bond_set(br, bdr)
    assert set(sa) == set(sr) and len(sa) > 50
    for key in sa:
        ka, kr = sa[key], sr[key]
        for name in ("tangd1", "tangd2", "nstress", "sstress", "rel_rotation"):
            assert bda[name][ka] == bdr[name][kr], (key, name)
        assert bda["broken"][ka] == bdr["broken"][kr]
    for name in ("lon", "lat", "uvel", "vvel", "ang_vel", "rot", "axn_fast"):
        assert np.array_equal(ba[name], br[name]), name
    a.run(2)
    r.run(2)
    ba, br = a.download_bergs(), r.download_bergs()
    for name in ("lon", "lat", "uvel", "vvel"):
        assert np.allclose(ba[name], br[name], rtol=1e-6, atol=1e-9), (name, float(np.abs(ba[name] - br[name]).max()))
    a.close()
    r.close()


@pytest.mark.gpu
def test_legacy_iceberg_num_file(tmp_path):
    """a restart file of the 32-bit era (iceberg_num, no id_cnt / id_ij): the ids are made by generate_id in file order
    (IO2:743, 917-918): counter 1, 2, ... per cell, the cell hash in the low word"""
    from icebergs_amd.framework import Icebergs
    grid, p, b = S.config_c2(n=400, seed=77)
    b["ine"][:40] = b["ine"][0]; b["jne"][:40] = b["jne"][0]      # forty bergs share a cell
    b["lon"][:40] = b["lon"][0]; b["lat"][:40] = b["lat"][0]
    with netcdf_file(str(tmp_path / "icebergs.res.nc"), "w", version=2) as f:
        f.createDimension("i", None)
        for name in ("lon", "lat", "uvel", "vvel", "mass", "thickness", "width", "length", "start_lon", "start_lat", "start_day", "start_mass",
                     "mass_scaling", "mass_of_bits", "heat_density"):
            v = f.createVariable(name, "d", ("i",)); v[:] = b[name]
        for name in ("ine", "jne", "start_year"):
            v = f.createVariable(name, "i", ("i",)); v[:] = b[name]
        v = f.createVariable("iceberg_num", "i", ("i",)); v[:] = np.arange(1, 401, dtype=np.int32)
    ib = Icebergs(grid, p, capacity=400)
    ib.set_forcing(grid["forcing"])
    ib.read_restart(tmp_path)
    got = ib.download_bergs()
    d = grid["desc"]
    nic = d.iec - d.isc + 1
    assert len(got["id"]) == 400 and len(np.unique(got["id"])) == 400
    assert np.array_equal(got["id"] & 0xFFFFFFFF, got["ine"] + nic * (got["jne"] - 1))
    assert list(got["id"][:40] >> 32) == list(range(1, 41))       # file order within the shared cell
    cnt = ib.get_iceberg_counter()
    assert cnt[b["jne"][0] - d.jsd, b["ine"][0] - d.isd] == 40 and cnt.sum() == 400
    ib.close()

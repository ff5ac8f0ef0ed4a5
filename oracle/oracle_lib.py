"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE -- never imported by icebergs_amd/).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, ".."))
from icebergs_amd import types as T  # noqa: E402  (interface types only)


class KoCalvingState(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_double)) for n in ("calving", "calving_hflx", "stored_ice", "stored_heat", "real_calving",
                                                     "rmean_calving", "rmean_calving_hflx")] + \
               [(n, C.c_int32) for n in ("first_call", "rmean_calving_initialized", "rmean_calving_hflx_initialized", "pad")]


class KoGrid(C.Structure):
    _fields_ = [("d", T.GridDesc),
                ("stat", C.POINTER(C.c_double) * T.ENUMS["KID_NGRID_STATIC"]),
                ("forc", C.POINTER(C.c_double) * T.ENUMS["KID_NFORCING"]),
                ("iceberg_counter", C.POINTER(C.c_int32))]


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def load():
    so = os.environ.get("KID_ORACLE_SO") or os.path.join(_HERE, "libkid_oracle.so")  # KID_ORACLE_SO: e.g. a sanitizer build
    if not os.path.exists(so):
        build()
    lib = C.CDLL(so)
    d = C.c_double
    lib.ko_sizeof.restype = C.c_int64
    assert lib.ko_sizeof(0) == C.sizeof(T.Params), "kid_params layout mismatch"
    assert lib.ko_sizeof(1) == C.sizeof(T.GridDesc)
    assert lib.ko_sizeof(2) == C.sizeof(T.BergSoA)
    assert lib.ko_sizeof(3) == C.sizeof(KoGrid)
    lib.ko_modulo.restype = d; lib.ko_modulo.argtypes = [d, d]
    lib.ko_apply_modulo_around_point.restype = d; lib.ko_apply_modulo_around_point.argtypes = [d, d, d]
    lib.ko_bilin.restype = d
    lib.ko_bilin.argtypes = [C.POINTER(KoGrid), C.POINTER(T.Params), C.POINTER(d), C.c_int, C.c_int, d, d]
    lib.ko_is_point_in_cell.restype = C.c_int
    lib.ko_is_point_in_cell.argtypes = [C.POINTER(KoGrid), d, d, C.c_int, C.c_int]
    lib.ko_pos_within_cell.restype = C.c_int
    lib.ko_pos_within_cell.argtypes = [C.POINTER(KoGrid), C.POINTER(T.Params), d, d, C.c_int, C.c_int,
                                       C.POINTER(d), C.POINTER(d), C.POINTER(C.c_int)]
    lib.ko_calc_xiyj.restype = C.c_int
    lib.ko_calc_xiyj.argtypes = [d] * 10 + [C.POINTER(d), C.POINTER(d), d]
    lib.ko_interp_flds.restype = None
    lib.ko_interp_flds.argtypes = [C.POINTER(KoGrid), C.POINTER(T.Params), d, d, C.c_int, C.c_int, d, d, C.POINTER(d)]
    lib.ko_find_basal_melt.restype = d
    lib.ko_find_basal_melt.argtypes = [C.POINTER(T.GridDesc), C.POINTER(T.Params), d, d, d, d, C.c_int, d]
    lib.ko_hexagon_into_quadrants.restype = None
    lib.ko_hexagon_into_quadrants.argtypes = [d, d, d, d] + [C.POINTER(d)] * 5
    lib.ko_point_in_triangle.restype = C.c_int
    lib.ko_point_in_triangle.argtypes = [d] * 8
    lib.ko_rolling.restype = None
    lib.ko_rolling.argtypes = [C.POINTER(T.Params)] + [C.POINTER(d)] * 3
    lib.ko_spread_weights.restype = None
    lib.ko_spread_weights.argtypes = [C.POINTER(KoGrid), C.POINTER(T.Params), C.c_int, C.c_int, d, d, d, d,
                                      C.POINTER(d), C.POINTER(d)]
    lib.ko_default_params.restype = None
    lib.ko_default_params.argtypes = [C.POINTER(T.Params)]
    lib.ko_calving.restype = C.c_int
    lib.ko_calving.argtypes = [C.POINTER(KoGrid), C.POINTER(T.Params), C.POINTER(T.CalvingParams), C.POINTER(d), C.POINTER(d),
                               C.POINTER(KoCalvingState), C.POINTER(T.BergSoA), C.c_int64, C.POINTER(d)]
    lib.ko_ingest_forcing.restype = C.c_int
    lib.ko_ingest_forcing.argtypes = [C.POINTER(KoGrid), C.POINTER(T.ForcingIn), C.POINTER(C.POINTER(d))]
    for name in ("ko_interp_gridded_fields_to_bergs",):
        getattr(lib, name).restype = None
        getattr(lib, name).argtypes = [C.POINTER(KoGrid), C.POINTER(T.Params), C.POINTER(T.BergSoA)]
    lib.ko_evolve_icebergs.restype = None
    lib.ko_evolve_icebergs.argtypes = [C.POINTER(KoGrid), C.POINTER(T.Params), C.POINTER(T.BergSoA), C.POINTER(d)]
    lib.ko_thermodynamics.restype = None
    lib.ko_thermodynamics.argtypes = [C.POINTER(KoGrid), C.POINTER(T.Params), C.POINTER(T.BergSoA), C.POINTER(d), C.POINTER(d)]
    lib.ko_create_gridded_icebergs_fields.restype = None
    lib.ko_create_gridded_icebergs_fields.argtypes = [C.POINTER(KoGrid), C.POINTER(T.Params), C.POINTER(T.BergSoA), C.POINTER(d), C.POINTER(d)]
    lib.ko_set_spread_mass_buffer.restype = None
    lib.ko_set_spread_mass_buffer.argtypes = [C.POINTER(d)]
    lib.ko_bergs_chksum.restype = None
    lib.ko_bergs_chksum.argtypes = [C.POINTER(KoGrid), C.POINTER(T.BergSoA), C.POINTER(C.c_int64)]
    lib.ko_philox4x32_10.restype = None
    lib.ko_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.ko_fl_uniform.restype = d; lib.ko_fl_uniform.argtypes = [C.c_int32, C.c_int64, C.c_int64, C.c_int32]
    lib.ko_set_fl_step.restype = None; lib.ko_set_fl_step.argtypes = [C.c_int64]
    lib.ko_get_fl_step.restype = C.c_int64; lib.ko_get_fl_step.argtypes = []
    lib.ko_footloose_calving.restype = None
    lib.ko_footloose_calving.argtypes = [C.POINTER(KoGrid), C.POINTER(T.Params), C.POINTER(T.BergSoA), C.c_int64, C.POINTER(d), C.POINTER(d)]
    lib.ko_step_local.restype = None
    lib.ko_step_local.argtypes = [C.POINTER(KoGrid), C.POINTER(T.Params), C.POINTER(T.BergSoA), C.c_int64, C.POINTER(d), C.POINTER(d)]
    lib.ko_gather_fields.restype = None
    lib.ko_gather_fields.argtypes = [C.POINTER(KoGrid), C.POINTER(T.Params), C.POINTER(d), C.POINTER(d)]
    lib.ko_run_step.restype = None
    lib.ko_run_step.argtypes = [C.POINTER(KoGrid), C.POINTER(T.Params), C.POINTER(T.BergSoA), C.c_int64,
                                C.POINTER(d), C.POINTER(d), C.POINTER(d)]
    lib.ko_send_bergs.restype = C.c_long
    lib.ko_send_bergs.argtypes = [C.POINTER(KoGrid), C.POINTER(T.BergSoA), C.c_int, C.POINTER(d)]
    lib.ko_unpack_bergs.restype = C.c_long
    lib.ko_unpack_bergs.argtypes = [C.POINTER(KoGrid), C.POINTER(T.Params), C.POINTER(T.BergSoA), C.POINTER(d), C.c_long]
    lib.ko_reference_order.restype = None
    lib.ko_reference_order.argtypes = [C.POINTER(T.BergSoA), C.POINTER(C.c_int64)]
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Oracle:
    """Drives the oracle over numpy state: a grid dict (see icebergs_amd.synthetic) and a berg dict."""

    def __init__(self, grid, params):
        self.lib = load()
        self.grid = grid
        self.params = params
        self.kg = KoGrid()
        self.kg.d = grid["desc"]
        self._keep = []
        for k, name in enumerate(T.GRID_STATIC_NAMES):
            a = np.ascontiguousarray(grid["static"][name], dtype=np.float64)
            self._keep.append(a)
            self.kg.stat[k] = _dp(a)
        self.set_forcing(grid["forcing"])
        ni = grid["desc"].ied - grid["desc"].isd + 1
        nj = grid["desc"].jed - grid["desc"].jsd + 1
        self.ni, self.nj = ni, nj
        self.iceberg_counter = np.zeros((nj, ni), dtype=np.int32)
        self.kg.iceberg_counter = self.iceberg_counter.ctypes.data_as(C.POINTER(C.c_int32))
        self.acc = np.zeros((T.NACC, nj, ni))
        self.out = np.zeros((T.NOUT, nj, ni))
        self.scalars = np.zeros(T.NSCALAR)

    def set_forcing(self, forcing):
        self._fkeep = []
        for k, name in enumerate(T.FORCING_NAMES):
            a = np.ascontiguousarray(forcing[name], dtype=np.float64)
            self._fkeep.append(a)
            self.kg.forc[k] = _dp(a)

    def ingest_forcing(self, args, vel_stagger="B", stress_stagger="B", tau_is_velocity=False, cyclic_x=False, planes=None):
        """ko_ingest_forcing (icebergs.F90:5236-5383) on the coupler arguments `args` (numpy arrays (n2, n1) = Fortran a(n1, n2));
        `planes`: the grd%* planes before the call (default: zeros); returns the planes after it, or None on refused extents"""
        st = {"B": T.ENUMS["KID_BGRID_NE"], "C": T.ENUMS["KID_CGRID_NE"], "A": T.ENUMS["KID_AGRID"]}
        fin, keep = T.ForcingIn(), []
        for name in ("uo", "vo", "ui", "vi", "tauxa", "tauya", "ssh", "sst", "cn", "hi", "sss"):
            v = args.get(name)
            if v is None:
                continue
            a = np.ascontiguousarray(v, dtype=np.float64)
            keep.append(a)
            setattr(fin, name, _dp(a))
        fin.u_nj, fin.u_ni = args["uo"].shape
        fin.v_nj, fin.v_ni = args["vo"].shape
        fin.taux_nj, fin.taux_ni = args["tauxa"].shape
        fin.tauy_nj, fin.tauy_ni = args["tauya"].shape
        fin.vel_stagger, fin.stress_stagger = st[vel_stagger], st[stress_stagger]
        fin.tau_is_velocity, fin.cyclic_x = int(tau_is_velocity), int(cyclic_x)
        out = {name: (np.zeros((self.nj, self.ni)) if planes is None else np.array(planes[name], dtype=np.float64, order="C"))
               for name in T.FORCING_NAMES}
        arr = (C.POINTER(C.c_double) * T.ENUMS["KID_NFORCING"])(*[_dp(out[name]) for name in T.FORCING_NAMES])
        rc = self.lib.ko_ingest_forcing(C.byref(self.kg), C.byref(fin), arr)
        return out if rc == 0 else None

    def new_calving_state(self):
        """grd%calving .. grd%rmean_calving_hflx as ice_bergs_framework_init leaves them (zeros), first_call = T"""
        nk = T.ENUMS["KID_NCLASSES"]
        st = {"calving": np.zeros((self.nj, self.ni)), "calving_hflx": np.zeros((self.nj, self.ni)),
              "stored_ice": np.zeros((nk, self.nj, self.ni)), "stored_heat": np.zeros((self.nj, self.ni)),
              "real_calving": np.zeros((nk, self.nj, self.ni)), "rmean_calving": np.zeros((self.nj, self.ni)),
              "rmean_calving_hflx": np.zeros((self.nj, self.ni)), "flags": [1, 0, 0]}
        return st

    def calving(self, cp, calving, calving_hflx, state, bergs, capacity):
        """ko_calving (IB:5203-5231, 6153-6402): updates `state` and appends to `bergs` (arrays sized `capacity`, bergs["_n"] rows live)"""
        ks = KoCalvingState()
        for n in ("calving", "calving_hflx", "stored_ice", "stored_heat", "real_calving", "rmean_calving", "rmean_calving_hflx"):
            assert state[n].dtype == np.float64 and state[n].flags.c_contiguous
            setattr(ks, n, _dp(state[n]))
        ks.first_call, ks.rmean_calving_initialized, ks.rmean_calving_hflx_initialized = state["flags"]
        a = np.ascontiguousarray(calving, dtype=np.float64)
        b = np.ascontiguousarray(calving_hflx, dtype=np.float64)
        soa = self.soa(bergs)
        scal = np.zeros(T.ENUMS["KID_NCALV_SCALARS"])
        rc = self.lib.ko_calving(C.byref(self.kg), C.byref(self.params), C.byref(cp), _dp(a), _dp(b), C.byref(ks), C.byref(soa), int(capacity), _dp(scal))
        state["flags"] = [ks.first_call, ks.rmean_calving_initialized, ks.rmean_calving_hflx_initialized]
        bergs["_n"] = int(soa.n)
        return rc, scal

    @staticmethod
    def soa(bergs):
        """bergs["_n"] (optional) = number of live rows when the arrays carry spare capacity for footloose children"""
        s = T.BergSoA()
        n = int(bergs.get("_n", len(bergs["lon"])))
        s.n = n
        for k, name in enumerate(T.BERG_F64_NAMES):
            a = bergs[name]
            assert a.dtype == np.float64 and a.flags.c_contiguous and len(a) >= n
            s.f64[k] = _dp(a)
        for k, name in enumerate(T.BERG_I32_NAMES):
            a = bergs[name]
            assert a.dtype == np.int32 and a.flags.c_contiguous
            s.i32[k] = a.ctypes.data_as(C.POINTER(C.c_int32))
        s.id = bergs["id"].ctypes.data_as(C.POINTER(C.c_int64))
        return s

    @staticmethod
    def bond_soa(bonds, n):
        s = T.BondSoA()
        s.n, s.max_bonds = n, int(bonds["max_bonds"])
        assert len(bonds["count"]) == n
        s.count = bonds["count"].ctypes.data_as(C.POINTER(C.c_int32))
        s.other_id = bonds["other_id"].ctypes.data_as(C.POINTER(C.c_int64))
        s.broken = bonds["broken"].ctypes.data_as(C.POINTER(C.c_int32))
        for k, name in enumerate(T.BOND_F64_NAMES):
            a = bonds[name]
            assert a.dtype == np.float64 and len(a) == s.max_bonds * n
            s.f64[k] = _dp(a)
        return s

    def run_step_mts(self, bergs, bonds, nsteps=1):
        """icebergs_run with mts=.true.; the first call after construction does the `Visited` block (IB:5409-5420)"""
        s = self.soa(bergs)
        bs = self.bond_soa(bonds, len(bergs["lon"]))
        for _ in range(nsteps):
            first = 0 if getattr(self, "_visited", False) else 1
            self._visited = True
            self.lib.ko_run_step_mts(C.byref(self.kg), C.byref(self.params), C.byref(s), C.byref(bs), first,
                                     _dp(self.acc), _dp(self.out), _dp(self.scalars))
        return bergs, bonds

    def run_step_interactive(self, bergs, bonds, nsteps=1):
        """icebergs_run with interactive_icebergs_on under the single-time-step (Verlet) scheme"""
        s = self.soa(bergs)
        bs = self.bond_soa(bonds, len(bergs["lon"]))
        for _ in range(nsteps):
            first = 0 if getattr(self, "_visited", False) else 1
            self._visited = True
            self.lib.ko_run_step_interactive(C.byref(self.kg), C.byref(self.params), C.byref(s), C.byref(bs), first,
                                             _dp(self.acc), _dp(self.out), _dp(self.scalars))
        return bergs, bonds

    def send_bergs(self, bergs, direction):
        """the packing loop of send_bergs_to_other_pes for one direction (0 E, 1 W, 2 N, 3 S): rows of 34 reals"""
        s = self.soa(bergs)
        buf = np.zeros((max(int(s.n), 1), 34))
        n = self.lib.ko_send_bergs(C.byref(self.kg), C.byref(s), int(direction), _dp(buf))
        return buf[:n].copy()

    def unpack_bergs(self, bergs, buf):
        """unpack_berg_from_buffer2 into the spare rows of `bergs` (bergs["_n"] live rows); returns the bergs no cell took"""
        buf = np.ascontiguousarray(buf, dtype=np.float64)
        s = self.soa(bergs)
        assert int(s.n) + len(buf) <= len(bergs["lon"]), "no room for the arrivals"
        lost = self.lib.ko_unpack_bergs(C.byref(self.kg), C.byref(self.params), C.byref(s), _dp(buf), len(buf)) if len(buf) else 0
        bergs["_n"] = int(s.n)
        return int(lost)

    def bergs_chksum(self, bergs):
        """(chksum, chksum2, chksum3, chksum4, chksum5, #) of bergs_chksum (FW:6889-6987)"""
        s = self.soa(bergs)
        out = (C.c_int64 * 6)()
        self.lib.ko_bergs_chksum(C.byref(self.kg), C.byref(s), out)
        return tuple(int(v) for v in out)

    def set_spread_mass_buffer(self, two_planes):
        """sharded find_melt_using_spread_mass: the array the ranks sum between step_local and step_gather (None: off)"""
        self._spread_buf = two_planes
        self.lib.ko_set_spread_mass_buffer(_dp(two_planes) if two_planes is not None else None)

    def step_local(self, bergs):
        s = self.soa(bergs)
        self.lib.ko_set_fl_step(getattr(self, "fl_step", 0))
        self._after_fl = True
        self.lib.ko_step_local(C.byref(self.kg), C.byref(self.params), C.byref(s), len(bergs["lon"]), _dp(self.acc), _dp(self.scalars))
        self.fl_step = int(self.lib.ko_get_fl_step())

    def step_gather(self):
        self.lib.ko_gather_fields(C.byref(self.kg), C.byref(self.params), _dp(self.acc), _dp(self.out))

    def run_step(self, bergs, nsteps=1):
        s = self.soa(bergs)
        # the footloose step (third counter word of the child-placement generator) is kept per Oracle object: the C library
        # holds one global
        self.lib.ko_set_fl_step(getattr(self, "fl_step", 0))
        for _ in range(nsteps):
            self.lib.ko_run_step(C.byref(self.kg), C.byref(self.params), C.byref(s), len(bergs["lon"]),
                                 _dp(self.acc), _dp(self.out), _dp(self.scalars))
        self.fl_step = int(self.lib.ko_get_fl_step())
        if "_n" in bergs:
            bergs["_n"] = int(s.n)  # footloose calving appends children
        return bergs

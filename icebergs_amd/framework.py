"""Host-side mirror of the reference's `ice_bergs` surface for the evolve-loop path.

The reference host is Fortran (icebergs_init / icebergs_run, src/icebergs.F90:92-178, 5074-5887) and the
Fortran binding lives in icebergs_amd/fortran/.  This module is the same thin host layer in Python, used by
the tests, bench.py and the multi-GPU driver: it owns no numerics, it only moves arrays across the C ABI
(include/kid.h) and calls the phases in the order icebergs_run does (IB:5423-5512).
"""
import atexit
import ctypes as C
import weakref

import numpy as np

from . import lib as _lib
from . import types as T


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


_LIVE = weakref.WeakSet()


@atexit.register
def _close_live_handles():
    """Handles still open when the interpreter exits are destroyed here, while the HIP runtime (and a profiler attached to it)
    is still up: a handle left to `__del__` during interpreter shutdown -- or never finalised at all -- used to keep its
    streams, captured graphs and the cooperative launch's buffers alive into the runtime's own exit handlers."""
    for ib in list(_LIVE):
        try:
            ib.close()
        except Exception:
            pass


class Icebergs:
    """One handle == one `type(icebergs)` container bound to one GPU (and one HIP stream)."""

    def __init__(self, grid, params, capacity, device=0):
        """icebergs_init (IB:92-178): grid + parameters -> device-resident state."""
        self.lib = _lib.load()
        self.grid = grid
        self.params = params
        self.device = int(device)
        d = grid["desc"]
        self.ni, self.nj = d.ied - d.isd + 1, d.jed - d.jsd + 1
        self.ncell = self.ni * self.nj
        self.h = C.c_void_p()
        rc = self.lib.kid_create(C.byref(d), C.byref(params), int(capacity), int(device), C.byref(self.h))
        if rc != 0:
            msg = self.lib.kid_last_error(self.h).decode() if self.h else "kid_create failed"
            if self.h:
                self.lib.kid_destroy(self.h)
                self.h = None
            raise _lib.KidError("kid_create: rc=%d %s" % (rc, msg))
        arr = (C.POINTER(C.c_double) * T.ENUMS["KID_NGRID_STATIC"])()
        keep = []
        for k, name in enumerate(T.GRID_STATIC_NAMES):
            a = np.ascontiguousarray(grid["static"][name], dtype=np.float64)
            assert a.shape == (self.nj, self.ni), (name, a.shape)
            keep.append(a)
            arr[k] = _dp(a)
        self._check(self.lib.kid_set_static_grid(self.h, arr), "kid_set_static_grid")
        self.set_forcing(grid["forcing"])
        self.acc = np.zeros((T.NACC, self.nj, self.ni))
        self.out = np.zeros((T.NOUT, self.nj, self.ni))
        self.scalars = np.zeros(T.NSCALAR)
        _LIVE.add(self)

    def _check(self, rc, what):
        if rc != 0:
            raise _lib.KidError("%s: rc=%d %s" % (what, rc, self.lib.kid_last_error(self.h).decode()))

    def close(self):
        if getattr(self, "h", None):
            self.lib.kid_destroy(self.h)
            self.h = None
        _LIVE.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- forcing (what icebergs_run's ingest block leaves in grd%*, IB:5236-5383) ----
    def set_forcing(self, forcing):
        arr = (C.POINTER(C.c_double) * T.ENUMS["KID_NFORCING"])()
        keep = []
        for k, name in enumerate(T.FORCING_NAMES):
            a = np.ascontiguousarray(forcing[name], dtype=np.float64)
            assert a.shape == (self.nj, self.ni), (name, a.shape)
            keep.append(a)
            arr[k] = _dp(a)
        self._check(self.lib.kid_set_forcing(self.h, arr), "kid_set_forcing")

    def ingest_forcing(self, args, vel_stagger="B", stress_stagger="B", tau_is_velocity=False, cyclic_x=False, on_device=False):
        """icebergs_run's ingest block on the device (kid_ingest_forcing, IB:5236-5383).  `args`: dict with uo, vo, ui, vi,
        tauxa, tauya, ssh, sst, cn, hi and optionally sss -- numpy arrays of shape (n2, n1) holding the Fortran array
        a(n1, n2), or with on_device (device address, (n2, n1)) pairs."""
        if on_device:   # resident arrays: the argument block is rebuilt only when an address or a switch changes
            key = (tuple(sorted((k, int(v[0]), tuple(v[1])) for k, v in args.items() if v is not None)), vel_stagger, stress_stagger, bool(tau_is_velocity), bool(cyclic_x))
            if getattr(self, "_ingest_key", None) == key:
                self._check(self.lib.kid_ingest_forcing(self.h, self._ingest_ref), "kid_ingest_forcing")
                return
        st = {"B": T.ENUMS["KID_BGRID_NE"], "C": T.ENUMS["KID_CGRID_NE"], "A": T.ENUMS["KID_AGRID"]}
        fin, keep, shape = T.ForcingIn(), [], {}
        for name in ("uo", "vo", "ui", "vi", "tauxa", "tauya", "ssh", "sst", "cn", "hi", "sss"):
            v = args.get(name)
            if v is None:
                setattr(fin, name, None)
                continue
            if on_device:
                ptr, shp = v
                setattr(fin, name, C.cast(C.c_void_p(int(ptr)), C.POINTER(C.c_double)))
            else:
                a = np.ascontiguousarray(v, dtype=np.float64)
                keep.append(a)
                shp = a.shape
                setattr(fin, name, _dp(a))
            shape[name] = shp
        for name, other in (("uo", "ui"), ("vo", "vi")):
            assert shape[name] == shape[other], (name, other)
        d = self.grid["desc"]
        nic, njc = d.iec - d.isc + 1, d.jec - d.jsc + 1
        for name in ("ssh", "cn", "hi"):
            assert shape[name] == (njc + 2, nic + 2), (name, shape[name])
        for name in ("sst", "sss"):
            assert name not in shape or shape[name] == (njc, nic), (name, shape[name])
        fin.u_nj, fin.u_ni = shape["uo"]
        fin.v_nj, fin.v_ni = shape["vo"]
        fin.taux_nj, fin.taux_ni = shape["tauxa"]
        fin.tauy_nj, fin.tauy_ni = shape["tauya"]
        fin.vel_stagger, fin.stress_stagger = st[vel_stagger], st[stress_stagger]
        fin.tau_is_velocity, fin.cyclic_x, fin.on_device = int(tau_is_velocity), int(cyclic_x), int(on_device)
        self._check(self.lib.kid_ingest_forcing(self.h, C.byref(fin)), "kid_ingest_forcing")
        if on_device:
            self._ingest_key, self._ingest_fin, self._ingest_ref = key, fin, C.byref(fin)

    def get_forcing(self):
        """grd%uo .. grd%hi as the handle holds them (after set_forcing or ingest_forcing)"""
        out = {name: np.empty((self.nj, self.ni)) for name in T.FORCING_NAMES}
        arr = (C.POINTER(C.c_double) * T.ENUMS["KID_NFORCING"])(*[_dp(out[name]) for name in T.FORCING_NAMES])
        self._check(self.lib.kid_get_forcing(self.h, arr), "kid_get_forcing")
        return out

    # ---- trajectories (record_posn FW:5328-5498, write_trajectory IO2:1631-2103) ----
    def set_traj_params(self, tp):
        self._check(self.lib.kid_set_traj_params(self.h, C.byref(tp)), "kid_set_traj_params")

    def record_posn(self):
        self._check(self.lib.kid_record_posn(self.h), "kid_record_posn")

    def num_traj_records(self):
        n = C.c_int64()
        self._check(self.lib.kid_num_traj_records(self.h, C.byref(n)), "kid_num_traj_records")
        return n.value

    def write_trajectories(self, path):
        self._check(self.lib.kid_write_trajectories(self.h, str(path).encode()), "kid_write_trajectories")

    def num_bond_traj_records(self):
        n = C.c_int64()
        self._check(self.lib.kid_num_bond_traj_records(self.h, C.byref(n)), "kid_num_bond_traj_records")
        return n.value

    def write_bond_trajectories(self, path):
        """write_bond_trajectory, IO2:2106-2331 (needs TrajParams.save_bond_traj and uploaded bonds)"""
        self._check(self.lib.kid_write_bond_trajectories(self.h, str(path).encode()), "kid_write_bond_trajectories")

    # ---- migration between the handles of a decomposed domain (send_bergs_to_other_pes FW:2997-3247) ----
    def buffer_width(self):
        w = C.c_int32()
        self._check(self.lib.kid_buffer_width(self.h, C.byref(w)), "kid_buffer_width")
        return w.value

    def pack_emigrants(self, direction):
        """bergs that left through `direction` (types.ENUMS['KID_DIR_E'] ...) as rows of the reference's wire format
        (pack_berg_into_buffer2 FW:3250-3301); they are gone from the handle afterwards"""
        n = C.c_int64()
        if getattr(self, "_mig_buf", None) is None:
            self._mig_buf = np.empty((4096, self.buffer_width()))
        rc = self.lib.kid_pack_emigrants(self.h, direction, _dp(self._mig_buf), self._mig_buf.shape[0], C.byref(n))
        if rc == -4 and n.value > self._mig_buf.shape[0]:        # KID_ECAPACITY: nothing was packed, n says how many wait
            self._mig_buf = np.empty((2 * n.value, self._mig_buf.shape[1]))
            rc = self.lib.kid_pack_emigrants(self.h, direction, _dp(self._mig_buf), self._mig_buf.shape[0], C.byref(n))
        self._check(rc, "kid_pack_emigrants")
        return self._mig_buf[:n.value].copy()

    def pack_emigrants_pair(self, axis):
        """east + west (axis 0) or north + south (axis 1) in one launch; returns the two record arrays"""
        na, nb = C.c_int64(), C.c_int64()
        if getattr(self, "_mig_buf2", None) is None:
            self._mig_buf2 = [np.empty((4096, self.buffer_width())) for _ in range(2)]
        for attempt in range(2):
            a, b = self._mig_buf2
            rc = self.lib.kid_pack_emigrants_pair(self.h, axis, _dp(a), a.shape[0], C.byref(na), _dp(b), b.shape[0], C.byref(nb))
            if rc != -4 or attempt:
                break
            self._mig_buf2 = [np.empty((max(2 * n.value, x.shape[0]), x.shape[1])) for n, x in ((na, a), (nb, b))]   # KID_ECAPACITY: nothing was packed
        self._check(rc, "kid_pack_emigrants_pair")
        return self._mig_buf2[0][:na.value].copy(), self._mig_buf2[1][:nb.value].copy()

    def unpack_immigrants_pair(self, buf_a, buf_b):
        a, b = (np.ascontiguousarray(x, dtype=np.float64) for x in (buf_a, buf_b))
        if a.size or b.size:
            self._check(self.lib.kid_unpack_immigrants_pair(self.h, _dp(a) if a.size else None, a.shape[0] if a.size else 0,
                                                            _dp(b) if b.size else None, b.shape[0] if b.size else 0), "kid_unpack_immigrants_pair")

    def unpack_immigrants(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.float64)
        if buf.size:
            self._check(self.lib.kid_unpack_immigrants(self.h, _dp(buf), buf.shape[0]), "kid_unpack_immigrants")

    # ---- restart files (icebergs_fms2io.F90:124-631, 663-1049) ----
    def write_restart(self, directory):
        self._check(self.lib.kid_write_restart(self.h, str(directory).encode()), "kid_write_restart")

    def read_restart(self, directory):
        self._check(self.lib.kid_read_restart(self.h, str(directory).encode()), "kid_read_restart")

    # ---- calving source (IB:5203-5231, accumulate_calving IB:6153, calve_icebergs IB:6225) ----
    def set_calving_params(self, cp):
        self._calv_params = cp
        self._check(self.lib.kid_set_calving_params(self.h, C.byref(cp)), "kid_set_calving_params")

    def set_calving_state(self, stored_ice=None, stored_heat=None, rmean_calving=None, rmean_calving_hflx=None):
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (stored_ice, stored_heat, rmean_calving, rmean_calving_hflx)]
        if arrs[0] is not None:
            assert arrs[0].shape == (T.ENUMS["KID_NCLASSES"], self.nj, self.ni)
        self._check(self.lib.kid_set_calving_state(self.h, *[None if a is None else _dp(a) for a in arrs]), "kid_set_calving_state")

    def get_calving_state(self):
        nk = T.ENUMS["KID_NCLASSES"]
        out = {"stored_ice": np.empty((nk, self.nj, self.ni)), "stored_heat": np.empty((self.nj, self.ni)),
               "rmean_calving": np.empty((self.nj, self.ni)), "rmean_calving_hflx": np.empty((self.nj, self.ni)),
               "real_calving": np.empty((nk, self.nj, self.ni)), "calving": np.empty((self.nj, self.ni)), "calving_hflx": np.empty((self.nj, self.ni))}
        self._check(self.lib.kid_get_calving_state(self.h, _dp(out["stored_ice"]), _dp(out["stored_heat"]), _dp(out["rmean_calving"]),
                                                   _dp(out["rmean_calving_hflx"]), _dp(out["real_calving"])), "kid_get_calving_state")
        self._check(self.lib.kid_get_calving(self.h, _dp(out["calving"]), _dp(out["calving_hflx"])), "kid_get_calving")
        return out

    def calving(self, calving, calving_hflx, on_device=False):
        """kid_calving; arrays cover (isc:iec, jsc:jec) as (njc, nic) numpy, or device addresses with on_device.
        Returns the KID_CS_* scalar increments."""
        cin = T.CalvingIn()
        if on_device:
            cin.calving = C.cast(C.c_void_p(int(calving)), C.POINTER(C.c_double))
            cin.calving_hflx = C.cast(C.c_void_p(int(calving_hflx)), C.POINTER(C.c_double))
        else:
            d = self.grid["desc"]
            a = np.ascontiguousarray(calving, dtype=np.float64)
            b = np.ascontiguousarray(calving_hflx, dtype=np.float64)
            assert a.shape == b.shape == (d.jec - d.jsc + 1, d.iec - d.isc + 1), a.shape
            cin.calving, cin.calving_hflx = _dp(a), _dp(b)
        cin.on_device = int(on_device)
        scal = np.zeros(T.ENUMS["KID_NCALV_SCALARS"])
        self._check(self.lib.kid_calving(self.h, C.byref(cin), _dp(scal)), "kid_calving")
        return scal

    def set_forcing_device(self, dev_ptrs):
        """dev_ptrs: KID_NFORCING device addresses (0 keeps a plane); asynchronous on the handle's stream."""
        arr = (C.c_void_p * T.ENUMS["KID_NFORCING"])(*[C.c_void_p(int(x)) if x else None for x in dev_ptrs])
        self._check(self.lib.kid_set_forcing_device(self.h, arr), "kid_set_forcing_device")

    def step_prepare(self, dev_ptrs=None):
        """forcing prepass + accumulator zeroing of the coming step in one launch (dev_ptrs as set_forcing_device)"""
        if dev_ptrs is None:
            self._check(self.lib.kid_step_prepare(self.h, None), "kid_step_prepare")
            return
        key = tuple(int(x) if x else 0 for x in dev_ptrs)
        if getattr(self, "_prep_key", None) != key:   # the pointer table is rebuilt only when the addresses change
            self._prep_key = key
            self._prep_arr = (C.c_void_p * T.ENUMS["KID_NFORCING"])(*[C.c_void_p(x) if x else None for x in key])
        self._check(self.lib.kid_step_prepare(self.h, self._prep_arr), "kid_step_prepare")

    def set_params(self, params):
        self.params = params
        self._check(self.lib.kid_set_params(self.h, C.byref(params)), "kid_set_params")

    def set_side_stream(self, stream_ptr, enable=True):
        """enable: False/0 off, True/1 two half launches, 2 the slow-lane schedule (include/kid.h)"""
        self._check(self.lib.kid_set_side_stream(self.h, C.c_void_p(stream_ptr), int(enable)), "kid_set_side_stream")

    def set_stream(self, stream_ptr):
        self._check(self.lib.kid_set_stream(self.h, C.c_void_p(stream_ptr)), "kid_set_stream")

    # ---- berg population ----
    @staticmethod
    def _soa(bergs, n=None):
        s = T.BergSoA()
        s.n = len(bergs["lon"]) if n is None else n
        for k, name in enumerate(T.BERG_F64_NAMES):
            a = bergs[name]
            assert a.dtype == np.float64 and a.flags.c_contiguous
            s.f64[k] = _dp(a)
        for k, name in enumerate(T.BERG_I32_NAMES):
            a = bergs[name]
            assert a.dtype == np.int32 and a.flags.c_contiguous
            s.i32[k] = a.ctypes.data_as(C.POINTER(C.c_int32))
        s.id = bergs["id"].ctypes.data_as(C.POINTER(C.c_int64))
        return s

    def upload_bergs(self, bergs):
        """bergs["_n"] (optional) = number of live rows when the arrays carry spare rows"""
        s = self._soa(bergs, bergs.get("_n"))
        self._check(self.lib.kid_upload_bergs(self.h, C.byref(s)), "kid_upload_bergs")

    # ---- bonds (mts / dem) ----
    @staticmethod
    def _bond_soa(bonds, n):
        s = T.BondSoA()
        s.n, s.max_bonds = n, int(bonds["max_bonds"])
        assert len(bonds["count"]) == n and bonds["count"].dtype == np.int32
        s.count = bonds["count"].ctypes.data_as(C.POINTER(C.c_int32))
        s.other_id = bonds["other_id"].ctypes.data_as(C.POINTER(C.c_int64))
        s.broken = bonds["broken"].ctypes.data_as(C.POINTER(C.c_int32))
        for k, name in enumerate(T.BOND_F64_NAMES):
            a = bonds[name]
            assert a.dtype == np.float64 and len(a) == s.max_bonds * n
            s.f64[k] = _dp(a)
        return s

    def upload_bonds(self, bonds):
        n, _ = self.num_bergs()
        self._check(self.lib.kid_upload_bonds(self.h, C.byref(self._bond_soa(bonds, n))), "kid_upload_bonds")

    def download_bonds(self, max_bonds):
        from .synthetic import empty_bonds
        n, _ = self.num_bergs()
        bd = empty_bonds(n, max_bonds)
        self._check(self.lib.kid_download_bonds(self.h, C.byref(self._bond_soa(bd, n))), "kid_download_bonds")
        return bd

    def set_conglom_ids(self):
        self._check(self.lib.kid_set_conglom_ids(self.h), "kid_set_conglom_ids")

    def evolve_icebergs_mts(self):
        self._check(self.lib.kid_evolve_icebergs_mts(self.h), "kid_evolve_icebergs_mts")

    def num_bergs(self):
        a, b = C.c_int64(), C.c_int64()
        self._check(self.lib.kid_num_bergs(self.h, C.byref(a), C.byref(b)), "kid_num_bergs")
        return a.value, b.value

    def download_bergs(self):
        from .synthetic import empty_bergs
        n, _ = self.num_bergs()
        b = empty_bergs(n)
        s = self._soa(b)
        self._check(self.lib.kid_download_bergs(self.h, C.byref(s)), "kid_download_bergs")
        return b

    def set_store_environment(self, on):
        self._check(self.lib.kid_set_store_environment(self.h, 1 if on else 0), "kid_set_store_environment")

    def bergs_chksum(self):
        """(chksum, chksum2, chksum3, chksum4, chksum5, #) of bergs_chksum (FW:6889-6987), as the reference prints them"""
        out = (C.c_int64 * 6)()
        self._check(self.lib.kid_bergs_chksum(self.h, out), "kid_bergs_chksum")
        return tuple(int(v) for v in out)

    def set_footloose_step(self, step):
        """continue the child-placement sequence of a restarted run (include/kid_rng.h)"""
        self._check(self.lib.kid_set_footloose_step(self.h, int(step)), "kid_set_footloose_step")

    def get_footloose_step(self):
        s = C.c_int64(0)
        self._check(self.lib.kid_get_footloose_step(self.h, C.byref(s)), "kid_get_footloose_step")
        return int(s.value)

    def set_iceberg_counter(self, counter):
        a = np.ascontiguousarray(counter, dtype=np.int32)
        assert a.shape == (self.nj, self.ni)
        self._check(self.lib.kid_set_iceberg_counter(self.h, a.ctypes.data_as(C.POINTER(C.c_int32))), "kid_set_iceberg_counter")

    def get_iceberg_counter(self):
        a = np.zeros((self.nj, self.ni), dtype=np.int32)
        self._check(self.lib.kid_get_iceberg_counter(self.h, a.ctypes.data_as(C.POINTER(C.c_int32))), "kid_get_iceberg_counter")
        return a

    def move_berg_between_cells(self):
        """IB:5437: re-bin (stable sort by cell, dead bergs dropped)."""
        self._check(self.lib.kid_move_berg_between_cells(self.h), "kid_move_berg_between_cells")

    def set_resort_interval(self, steps):
        self._check(self.lib.kid_set_resort_interval(self.h, int(steps)), "kid_set_resort_interval")

    def compact(self):
        self._check(self.lib.kid_compact_bergs(self.h), "kid_compact_bergs")

    # ---- icebergs_run, the hot-path part (IB:5423-5512) ----
    def run(self, nsteps=1):
        """Fused path: per step one per-berg launch (evolve + thermodynamics + spreading) and one gather."""
        self._check(self.lib.kid_run_step(self.h, int(nsteps)), "kid_run_step")

    def step_local(self):
        self._check(self.lib.kid_step_local(self.h), "kid_step_local")

    def step_gather(self):
        self._check(self.lib.kid_step_gather(self.h), "kid_step_gather")

    def run_phases(self, nsteps=1):
        """Phase-by-phase path: exactly the reference's call sequence inside icebergs_run."""
        p = self.params
        for _ in range(nsteps):
            self._check(self.lib.kid_zero_accumulators(self.h), "kid_zero_accumulators")          # IB:5125-5156
            if not p.mts and not p.old_interp_flds_order:
                self._check(self.lib.kid_interp_gridded_fields_to_bergs(self.h), "interp")          # IB:5423
            self._check(self.lib.kid_evolve_icebergs(self.h), "kid_evolve_icebergs")               # IB:5433
            if p.footloose:
                self._check(self.lib.kid_footloose_calving(self.h), "kid_footloose_calving")       # IB:5453
            if not p.old_interp_flds_order:
                self._check(self.lib.kid_interp_gridded_fields_to_bergs(self.h), "interp")          # IB:5473
            self._check(self.lib.kid_thermodynamics(self.h), "kid_thermodynamics")                 # IB:5505
            self._check(self.lib.kid_create_gridded_icebergs_fields(self.h), "kid_create_gridded") # IB:5512

    def fetch(self):
        self._check(self.lib.kid_get_accumulators(self.h, _dp(self.acc), _dp(self.out), _dp(self.scalars)), "kid_get_accumulators")
        return self.acc, self.out, self.scalars

    def sync(self):
        self._check(self.lib.kid_sync(self.h), "kid_sync")

    def accum_device_ptr(self):
        p, n = C.c_void_p(), C.c_int64()
        self._check(self.lib.kid_accum_device_ptr(self.h, C.byref(p), C.byref(n)), "kid_accum_device_ptr")
        return p.value, n.value

    def accum_live_count(self):
        """doubles from the start of the accumulator block that a sharded run sums across GPUs (include/kid.h)"""
        n = C.c_int64()
        self._check(self.lib.kid_accum_live_count(self.h, C.byref(n)), "kid_accum_live_count")
        return n.value

    def bind_accum_buffer(self, ptr, count):
        self._check(self.lib.kid_bind_accum_buffer(self.h, C.c_void_p(ptr), int(count)), "kid_bind_accum_buffer")

    def bind_spread_mass_old(self, ptr, count):
        """find_melt_using_spread_mass in a sharded run: the two planes the ranks must sum (include/kid.h)"""
        self._check(self.lib.kid_bind_spread_mass_old(self.h, ptr, int(count)), "kid_bind_spread_mass_old")

    def last_redo_count(self):
        c = C.c_int64()
        self._check(self.lib.kid_last_redo_count(self.h, C.byref(c)), "kid_last_redo_count")
        return c.value

    def profile(self, on=True):
        self._check(self.lib.kid_profile_enable(self.h, 1 if on else 0), "kid_profile_enable")

    def profile_get(self):
        a, n, b = C.c_double(), C.c_int64(), C.c_double()
        self._check(self.lib.kid_profile_get(self.h, C.byref(a), C.byref(n), C.byref(b)), "kid_profile_get")
        return a.value, n.value, b.value

"""ctypes loader of libkid_hip.so, the C-ABI product library (include/kid.h).

There is no CPU fallback: if the shared object is missing or was not built, ``load()`` raises.
"""
import ctypes as C
import os
import subprocess

from . import types as T

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SO_PATH = os.environ.get("KID_HIP_SO", os.path.join(_CSRC, "libkid_hip.so"))  # override: A/B builds of experiments

# every symbol include/kid.h declares (checked by tests/test_abi.py without a GPU)
SYMBOLS = [
    "kid_create", "kid_destroy", "kid_set_params", "kid_set_stream", "kid_sync", "kid_last_error", "kid_version",
    "kid_sizeof", "kid_set_static_grid", "kid_set_forcing", "kid_set_forcing_device", "kid_upload_bergs", "kid_download_bergs",
    "kid_num_bergs", "kid_compact_bergs", "kid_move_berg_between_cells", "kid_set_resort_interval", "kid_zero_accumulators", "kid_interp_gridded_fields_to_bergs",
    "kid_evolve_icebergs", "kid_footloose_calving", "kid_set_footloose_step", "kid_get_footloose_step", "kid_footloose_uniform", "kid_thermodynamics", "kid_create_gridded_icebergs_fields",
    "kid_set_store_environment", "kid_set_iceberg_counter", "kid_get_iceberg_counter", "kid_step_local", "kid_step_gather", "kid_run_step", "kid_get_accumulators", "kid_accum_device_ptr", "kid_accum_live_count",
    "kid_bind_accum_buffer", "kid_bind_spread_mass_old", "kid_profile_enable", "kid_profile_get",
    "kid_last_redo_count", "kid_set_side_stream", "kid_step_prepare", "kid_upload_bonds", "kid_download_bonds", "kid_evolve_icebergs_mts", "kid_set_conglom_ids", "kid_evolve_icebergs_interactive",
    "kid_ingest_forcing", "kid_get_forcing",
    "kid_set_calving_params", "kid_set_calving_state", "kid_get_calving_state", "kid_calving", "kid_get_calving",
    "kid_restart_write_bergs", "kid_restart_count_bergs", "kid_restart_read_bergs", "kid_restart_write_bonds", "kid_restart_read_bonds", "kid_write_restart", "kid_read_restart", "kid_bergs_chksum",
    "kid_set_traj_params", "kid_record_posn", "kid_num_traj_records", "kid_write_trajectories",
    "kid_num_bond_traj_records", "kid_write_bond_trajectories",
    "kid_buffer_width", "kid_pack_emigrants", "kid_unpack_immigrants", "kid_pack_emigrants_pair", "kid_unpack_immigrants_pair",
]


class KidError(RuntimeError):
    pass


def build(verbose=False):
    """Compile libkid_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", _CSRC, "-j2", "all"] + ([] if verbose else ["-s"]), check=True)   # the library and its exact-math twin (tests only)
    return SO_PATH


def device_reset():
    """hipDeviceReset() through the HIP runtime the process already holds.  For stand-alone scripts that end under a profiler:
    a process that has run a cooperative launch (the fused MTS sub-step kernel) and exits under rocprofv3 faults inside the
    runtime's own exit handlers, after the tool has finalised -- with every handle closed as well; resetting the device while
    everything is still up avoids relying on that exit order.  Never called by the library itself."""
    C.CDLL("libamdhip64.so").hipDeviceReset()


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise KidError("libkid_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                       "there is no CPU fallback")
    # torch ships its own libamdhip64: whichever HIP runtime a process loads second finds no GPUs.  Python callers use
    # torch for device memory and RCCL, so let its runtime load first; libkid_hip.so then binds to the same SONAME.
    import torch  # noqa: F401
    lib = C.CDLL(SO_PATH)
    H = C.c_void_p
    dp = C.POINTER(C.c_double)
    lib.kid_version.restype = C.c_char_p
    lib.kid_last_error.restype = C.c_char_p
    lib.kid_last_error.argtypes = [H]
    lib.kid_sizeof.restype = C.c_int64
    lib.kid_sizeof.argtypes = [C.c_int]
    lib.kid_create.argtypes = [C.POINTER(T.GridDesc), C.POINTER(T.Params), C.c_int64, C.c_int, C.POINTER(H)]
    lib.kid_destroy.argtypes = [H]
    lib.kid_set_params.argtypes = [H, C.POINTER(T.Params)]
    lib.kid_set_stream.argtypes = [H, C.c_void_p]
    lib.kid_sync.argtypes = [H]
    lib.kid_set_static_grid.argtypes = [H, C.POINTER(dp)]
    lib.kid_set_forcing.argtypes = [H, C.POINTER(dp)]
    lib.kid_set_forcing_device.argtypes = [H, C.POINTER(C.c_void_p)]
    lib.kid_ingest_forcing.argtypes = [H, C.POINTER(T.ForcingIn)]
    lib.kid_get_forcing.argtypes = [H, C.POINTER(dp)]
    lib.kid_set_calving_params.argtypes = [H, C.POINTER(T.CalvingParams)]
    lib.kid_set_calving_state.argtypes = [H, dp, dp, dp, dp]
    lib.kid_get_calving_state.argtypes = [H, dp, dp, dp, dp, dp]
    lib.kid_calving.argtypes = [H, C.POINTER(T.CalvingIn), dp]
    lib.kid_get_calving.argtypes = [H, dp, dp]
    lib.kid_restart_write_bergs.argtypes = [C.c_char_p, C.POINTER(T.Params), C.POINTER(T.BergSoA)]
    lib.kid_restart_count_bergs.argtypes = [C.c_char_p, C.POINTER(C.c_int64)]
    lib.kid_restart_read_bergs.argtypes = [C.c_char_p, C.POINTER(T.BergSoA), C.c_int64]
    lib.kid_restart_write_bonds.argtypes = [C.c_char_p, C.POINTER(T.Params), C.POINTER(T.BergSoA), C.POINTER(T.BondSoA)]
    lib.kid_restart_read_bonds.argtypes = [C.c_char_p, C.POINTER(T.BergSoA), C.POINTER(T.BondSoA)]
    lib.kid_write_restart.argtypes = [H, C.c_char_p]
    lib.kid_set_traj_params.argtypes = [H, C.POINTER(T.TrajParams)]
    lib.kid_record_posn.argtypes = [H]
    lib.kid_num_traj_records.argtypes = [H, C.POINTER(C.c_int64)]
    lib.kid_write_trajectories.argtypes = [H, C.c_char_p]
    lib.kid_num_bond_traj_records.argtypes = [H, C.POINTER(C.c_int64)]
    lib.kid_write_bond_trajectories.argtypes = [H, C.c_char_p]
    lib.kid_buffer_width.argtypes = [H, C.POINTER(C.c_int32)]
    lib.kid_pack_emigrants.argtypes = [H, C.c_int32, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int64)]
    lib.kid_unpack_immigrants.argtypes = [H, C.POINTER(C.c_double), C.c_int64]
    lib.kid_pack_emigrants_pair.argtypes = [H, C.c_int32, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int64)]
    lib.kid_unpack_immigrants_pair.argtypes = [H, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.c_int64]
    lib.kid_read_restart.argtypes = [H, C.c_char_p]
    lib.kid_upload_bergs.argtypes = [H, C.POINTER(T.BergSoA)]
    lib.kid_step_prepare.argtypes = [H, C.POINTER(C.c_void_p)]
    lib.kid_set_side_stream.argtypes = [H, C.c_void_p, C.c_int]
    lib.kid_last_redo_count.argtypes = [H, C.POINTER(C.c_int64)]
    lib.kid_upload_bonds.argtypes = [H, C.POINTER(T.BondSoA)]
    lib.kid_download_bonds.argtypes = [H, C.POINTER(T.BondSoA)]
    lib.kid_evolve_icebergs_mts.argtypes = [H]
    lib.kid_set_conglom_ids.argtypes = [H]
    lib.kid_evolve_icebergs_interactive.argtypes = [H]
    lib.kid_download_bergs.argtypes = [H, C.POINTER(T.BergSoA)]
    lib.kid_num_bergs.argtypes = [H, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.kid_set_resort_interval.argtypes = [H, C.c_int]
    for name in ("kid_compact_bergs", "kid_move_berg_between_cells", "kid_zero_accumulators", "kid_interp_gridded_fields_to_bergs",
                 "kid_evolve_icebergs", "kid_footloose_calving", "kid_thermodynamics",
                 "kid_create_gridded_icebergs_fields", "kid_step_local", "kid_step_gather"):
        getattr(lib, name).argtypes = [H]
    lib.kid_run_step.argtypes = [H, C.c_int]
    lib.kid_set_store_environment.argtypes = [H, C.c_int]
    lib.kid_bergs_chksum.argtypes = [H, C.POINTER(C.c_int64)]
    lib.kid_footloose_uniform.argtypes = [C.c_int32, C.c_int64, C.c_int64, C.c_int32]
    lib.kid_set_footloose_step.argtypes = [H, C.c_int64]
    lib.kid_get_footloose_step.argtypes = [H, C.POINTER(C.c_int64)]
    lib.kid_set_iceberg_counter.argtypes = [H, C.POINTER(C.c_int32)]
    lib.kid_get_iceberg_counter.argtypes = [H, C.POINTER(C.c_int32)]
    lib.kid_get_accumulators.argtypes = [H, dp, dp, dp]
    lib.kid_accum_device_ptr.argtypes = [H, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    lib.kid_accum_live_count.argtypes = [H, C.POINTER(C.c_int64)]
    lib.kid_bind_accum_buffer.argtypes = [H, C.c_void_p, C.c_int64]
    lib.kid_bind_spread_mass_old.argtypes = [H, C.c_void_p, C.c_int64]
    lib.kid_profile_enable.argtypes = [H, C.c_int]
    lib.kid_profile_get.argtypes = [H, dp, C.POINTER(C.c_int64), dp]
    for name in SYMBOLS:
        if name not in ("kid_version", "kid_last_error", "kid_sizeof", "kid_footloose_uniform"):
            getattr(lib, name).restype = C.c_int
    lib.kid_footloose_uniform.restype = C.c_double
    assert lib.kid_sizeof(0) == C.sizeof(T.Params), "kid_params layout mismatch between header and library"
    assert lib.kid_sizeof(1) == C.sizeof(T.GridDesc)
    assert lib.kid_sizeof(2) == C.sizeof(T.BergSoA)
    assert lib.kid_sizeof(3) == C.sizeof(T.BondSoA)
    assert lib.kid_sizeof(4) == C.sizeof(T.ForcingIn)
    assert lib.kid_sizeof(5) == C.sizeof(T.CalvingParams) and lib.kid_sizeof(6) == C.sizeof(T.CalvingIn)
    _lib = lib
    return lib

"""icebergs_amd: MI355X-native evolve loop for NOAA-GFDL/icebergs (KID).

Only the per-berg hot path lives here: HIP kernels + C-ABI in ``csrc/``, the
ISO_C_BINDING shim in ``fortran/`` and a thin Python host mirror of ``icebergs_init``/``icebergs_run``.
Importing the package does not load the HIP library; ``icebergs_amd.lib.load()`` does and fails loudly
if it is missing.
"""
__all__ = ["types", "lib", "framework", "synthetic"]

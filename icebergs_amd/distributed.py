"""Particle-index sharding of the evolve loop over the GPUs of one node.

Bergs do not interact in BASELINE configs 2/3/5, so each rank owns a contiguous block of the cell-sorted SoA and
steps it on its own GPU against a replicated grid.  The one exchange per step is the sum of the per-cell
accumulators the path scatters into (melt / heat fluxes, the 9-slot mass/area/momentum-on-ocean planes and the step's
scalar increments): an in-place RCCL all-reduce over xGMI of one contiguous fp64 block, followed on every rank by
the local 9-point gather (sum_up_spread_fields, icebergs.F90:6126-6138).  This replaces the reference's grid halo
updates (icebergs.F90:6107) and berg migration (icebergs_framework.F90:2997-3248), which a replicated grid does not
need.  There is no other data-path collective.
"""
import numpy as np

from . import types as T


def shard_bounds(n, rank, world):
    """Contiguous block [lo, hi) of rank `rank`: blocks differ by at most one berg."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def take_shard(bergs, rank, world):
    lo, hi = shard_bounds(len(bergs["lon"]), rank, world)
    return {k: np.ascontiguousarray(v[lo:hi]) for k, v in bergs.items()}


def accumulator_views(acc_block, ncell, diag_mask=0):
    """(planes that must be reduced, scalar increments) as views of the contiguous accumulator block."""
    diag_planes = sum(T.ENUMS[k] for k in (
        "KID_DIAG_MELT_BY_CLASS", "KID_DIAG_FL_PARENT_MELT", "KID_DIAG_FL_CHILD_MELT", "KID_DIAG_MELT_BUOY",
        "KID_DIAG_MELT_EROS", "KID_DIAG_MELT_CONV", "KID_DIAG_MELT_BUOY_FL", "KID_DIAG_MELT_EROS_FL",
        "KID_DIAG_MELT_CONV_FL", "KID_DIAG_VIRTUAL_AREA", "KID_DIAG_MASS", "KID_DIAG_U_ICEBERG", "KID_DIAG_V_ICEBERG"))
    nplanes = T.NACC if (diag_mask & diag_planes) else T.ENUMS["KID_NACC_CORE"]
    return acc_block[: nplanes * ncell], acc_block[T.NACC * ncell: T.NACC * ncell + T.NSCALAR]


class ShardedStepper:
    """One coupling step of the sharded path: local per-berg work, all-reduce, local gather.

    `backend` is anything with step_local() / step_gather() that accumulates into `acc_block` (a torch tensor:
    device memory bound to the HIP handle in production, host memory in the gloo tests)."""

    def __init__(self, backend, acc_block, ncell, diag_mask=0, dist=None, resort_interval=16):
        self.backend = backend
        self.resort_interval = resort_interval
        self._since_sort = 0
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size() > 1) else None
        self.planes, self.scalars = accumulator_views(acc_block, ncell, diag_mask)

    def step(self):
        self.backend.step_local()
        if self.dist is not None:
            self.dist.all_reduce(self.planes)
            self.dist.all_reduce(self.scalars)
        self.backend.step_gather()
        self._since_sort += 1
        if self.resort_interval and self._since_sort >= self.resort_interval and hasattr(self.backend, "move_berg_between_cells"):
            self.backend.move_berg_between_cells()  # icebergs.F90:5437, amortised over `resort_interval` steps
            self._since_sort = 0

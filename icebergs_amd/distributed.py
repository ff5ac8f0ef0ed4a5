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


def needs_footprint_planes(params):
    """True if the gather reads area/Uvel/Vvel_on_ocean (27 of the 46 core planes): spread_area / spread_uvel /
    spread_vvel / ustar_iceberg exist only for `id_*>0` or pass_fields_to_ocean_model (IB:3419-3474); otherwise only
    mass_on_ocean is summed (IB:3406) and the library neither scatters, zeroes nor exchanges those 27 planes."""
    if params is None:
        return True
    E = T.ENUMS
    diag = E["KID_DIAG_SPREAD_UVEL"] | E["KID_DIAG_SPREAD_VVEL"] | E["KID_DIAG_SPREAD_AREA"] | E["KID_DIAG_USTAR_ICEBERG"]
    return bool(params.pass_fields_to_ocean_model or (params.diag_mask & diag))   # = footprint_needed() in kid_hip.hip


def accumulator_views(acc_block, ncell, diag_mask=0, params=None):
    """(what must be reduced: the step's scalar increments + the planes the step fills, ONE contiguous view; the scalars alone)
    of the accumulator block, which is laid out scalars first (include/kid.h kid_accum_device_ptr)."""
    diag_planes = sum(T.ENUMS[k] for k in (
        "KID_DIAG_MELT_BY_CLASS", "KID_DIAG_FL_PARENT_MELT", "KID_DIAG_FL_CHILD_MELT", "KID_DIAG_MELT_BUOY",
        "KID_DIAG_MELT_EROS", "KID_DIAG_MELT_CONV", "KID_DIAG_MELT_BUOY_FL", "KID_DIAG_MELT_EROS_FL",
        "KID_DIAG_MELT_CONV_FL", "KID_DIAG_VIRTUAL_AREA", "KID_DIAG_MASS", "KID_DIAG_U_ICEBERG", "KID_DIAG_V_ICEBERG"))
    if diag_mask & diag_planes:
        nplanes = T.NACC
    elif needs_footprint_planes(params):
        nplanes = T.ENUMS["KID_NACC_CORE"]
    else:
        nplanes = T.ENUMS["KID_A_MASS_ON_OCEAN"] + 9   # melt/heat/bits planes + the 9 mass_on_ocean slots
    return acc_block[: T.NSCALAR + nplanes * ncell], acc_block[: T.NSCALAR]


def needs_spread_mass_old(params):
    """find_melt_using_spread_mass (IB:5490-5503): the melt flux is what the GRIDDED mass lost over the step, so the gridded
    mass before the thermodynamics (grd%spread_mass_old; with Iceberg_melt_without_decay also spread_mass_tmp, IB:3411-3413)
    is part of what the ranks must sum: two more planes, 9-point gathers of per-cell sums and therefore linear in every
    rank's contribution, reduced between the local work and the gather together with the accumulator planes."""
    return params is not None and bool(params.find_melt_using_spread_mass)


class ShardedStepper:
    """One coupling step of the sharded path: local per-berg work, all-reduce, local gather.

    `backend` is anything with step_local() / step_gather() that accumulates into `acc_block` (a torch tensor:
    device memory bound to the HIP handle in production, host memory in the gloo tests)."""

    def __init__(self, backend, acc_block, ncell, diag_mask=0, dist=None, resort_interval=16, params=None, force_collective=False,
                 spread_mass_old=None):
        """spread_mass_old: with find_melt_using_spread_mass, the 2-plane tensor the backend was told to keep grd%spread_mass_old /
        spread_mass_tmp in (kid_bind_spread_mass_old on the HIP handle)"""
        self.backend = backend
        self.spread_mass_old = spread_mass_old if needs_spread_mass_old(params) else None
        if needs_spread_mass_old(params) and spread_mass_old is None:
            raise ValueError("find_melt_using_spread_mass: pass the tensor bound with kid_bind_spread_mass_old (2 planes)")
        self.resort_interval = resort_interval
        self._since_sort = 0
        self.dist = dist if (dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force_collective)) else None
        self.planes, self.scalars = accumulator_views(acc_block, ncell, diag_mask, params)

    def set_forcing_device(self, ptrs):
        """forcing of the step about to be taken (device addresses); applied by step() in its fused prepass"""
        self._forcing = list(ptrs)

    def step(self):
        if hasattr(self.backend, "step_prepare"):
            self.backend.step_prepare(getattr(self, "_forcing", None))   # forcing prepass + accumulator zeroing, one launch
        self.backend.step_local()
        if self.dist is not None:
            self.dist.all_reduce(self.planes)      # scalars + live planes: one collective
            if self.spread_mass_old is not None:
                self.dist.all_reduce(self.spread_mass_old)
        self.backend.step_gather()
        self._since_sort += 1
        if self.resort_interval and self._since_sort >= self.resort_interval and hasattr(self.backend, "move_berg_between_cells"):
            self.backend.move_berg_between_cells()  # icebergs.F90:5437, amortised over `resort_interval` steps
            self._since_sort = 0


    def flush(self):
        pass


class PipelinedStepper:
    """ShardedStepper for the HIP handle with the exchange taken off the critical path.

    The per-berg kernels of step k+1 do not read what the all-reduce and the 9-point gather of step k produce (those
    feed the host model, not the bergs), so the handle alternates between two accumulator blocks: step k scatters into
    block k%2 on the compute stream; a second stream waits for it, all-reduces the block over RCCL and runs the gather
    while the compute stream is already zeroing and filling the other block.  `flush()` joins both streams; outputs
    (kid_get_accumulators) are those of the last flushed step."""

    def __init__(self, ib, params, dist=None, resort_interval=16, force_collective=False, split_general=False, slow_lane=False):
        import torch
        self.torch, self.ib, self.params = torch, ib, params
        if needs_spread_mass_old(params):
            raise NotImplementedError("find_melt_using_spread_mass: use ShardedStepper (its gather needs this step's summed planes, there is nothing to overlap)")
        self.dist = dist if (dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force_collective)) else None
        self.dev = torch.device("cuda", ib.device)
        _, self.count = ib.accum_device_ptr()
        self.acc = [torch.zeros(self.count, dtype=torch.float64, device=self.dev) for _ in range(2)]
        self.views = [accumulator_views(a, ib.ncell, params.diag_mask, params) for a in self.acc]
        self.compute = torch.cuda.current_stream(self.dev)
        self.comm = torch.cuda.Stream(self.dev)
        self.local_done = [torch.cuda.Event() for _ in range(2)]
        self.gather_done = [torch.cuda.Event() for _ in range(2)]
        # ustar reads the ocean velocity records; with the slow-lane schedule the library double-buffers them itself
        self.gather_reads_forcing = needs_footprint_planes(params) and not slow_lane
        # split_general: the general build (cell hops, bounces) of each half of the population also goes to the second
        # stream (kid_set_side_stream).  Measured on one MI355X it does not pay at 1e6 bergs: two half launches of the hot
        # build cost ~30 us more than one and the four cross-stream dependencies ~45 us, against ~65 us hidden.
        ib.set_stream(self.compute.cuda_stream)
        # slow_lane: the general build (a single-wave latency) leaves the critical path altogether: bergs the hot build
        # hands over are stepped on the second stream for two steps (kid_set_side_stream mode 2, include/kid.h)
        ib.set_side_stream(self.comm.cuda_stream, 2 if slow_lane else (1 if split_general else 0))
        self.resort_interval, self._since_sort, self.k = resort_interval, 0, 0
        # Slow lane: the library orders the streams itself (lanes_eligible / launch_berg_lanes in kid_hip.hip): each prepass
        # waits for everything enqueued on the second stream up to the previous step's carry-over launch, which follows the
        # gather of the step before; and the second stream waits for the hot build before anything of the step runs on
        # it.  The events below would only add barrier packets (~5 us each) to the critical stream.
        self.lib_orders = bool(slow_lane) and self._slow_lane_eligible(ib, params)
        if self.lib_orders:
            ib.set_resort_interval(resort_interval)
        # Slow lane with an exchange (N > 1): the second stream already carries two general-build launches per step
        # (~130 us); the all-reduce and the gather go to a third stream so that neither queue comes near the ~320 us
        # period of the hot builds.  One more barrier on the main stream: the block must not be zeroed under its gather.
        self.three = self.lib_orders and self.dist is not None
        if self.three:
            self.xchg = torch.cuda.Stream(self.dev)
            self.new_done = [torch.cuda.Event() for _ in range(2)]

    def set_forcing_device(self, ptrs):
        """forcing of the step about to be taken (device addresses); applied by step() in its fused prepass"""
        self._forcing = list(ptrs)

    @staticmethod
    def _slow_lane_eligible(ib, p):
        """mirror of lanes_eligible() in csrc/kid_hip.hip: the fused step without footloose, bonds or interactions"""
        return bool(not p.static_icebergs and not p.mts and not p.interactive_icebergs_on
                    and not p.footloose and not (p.grounding_fraction > 0.0) and ib.num_bergs()[0] >= 4096)

    def step(self):
        torch, ib, cur = self.torch, self.ib, self.k & 1
        if self.k >= 2 and (self.three or not self.lib_orders):
            self.compute.wait_event(self.gather_done[cur])   # block `cur` was last read by the gather of step k-2
        if self.gather_reads_forcing and self.k >= 1:        # the previous gather may still be reading the old records
            self.compute.wait_event(self.gather_done[(self.k - 1) & 1])
        ib.bind_accum_buffer(self.acc[cur].data_ptr(), self.count)
        ib.set_stream(self.compute.cuda_stream)
        ib.step_prepare(getattr(self, "_forcing", None))     # forcing prepass + zeroing of block `cur`, one launch
        ib.step_local()
        if self.three:
            self.new_done[cur].record(self.comm)            # behind the general-build launches of this step
            with torch.cuda.stream(self.xchg):
                self.xchg.wait_event(self.new_done[cur])
                if self.dist is not None:
                    self.dist.all_reduce(self.views[cur][0])   # scalars + live planes: one collective
                ib.set_stream(self.xchg.cuda_stream)
                ib.step_gather()
                self.gather_done[cur].record(self.xchg)
            ib.set_stream(self.compute.cuda_stream)
            self._after_step()
            return
        if not self.lib_orders:
            self.local_done[cur].record(self.compute)
        with torch.cuda.stream(self.comm):
            if not self.lib_orders:
                self.comm.wait_event(self.local_done[cur])
            if self.dist is not None:
                self.dist.all_reduce(self.views[cur][0])   # scalars + live planes: one collective
            ib.set_stream(self.comm.cuda_stream)
            ib.step_gather()
            if not self.lib_orders:
                self.gather_done[cur].record(self.comm)
        ib.set_stream(self.compute.cuda_stream)
        self._after_step()

    def _after_step(self):
        self._since_sort += 1
        if self.lib_orders:
            pass   # slow lane: the library re-bins inside the step, between the hot build and the general build
        elif self.resort_interval and self._since_sort >= self.resort_interval:
            self.ib.move_berg_between_cells()
            self._since_sort = 0
        self.k += 1

    def flush(self):
        if getattr(self, "three", False):
            self.xchg.synchronize()
        self.comm.synchronize()
        self.compute.synchronize()

"""The N>1 path on CPU: two gloo ranks shard a population by particle index, all-reduce the per-cell accumulators
and gather; every rank must end with the single-process result.  The per-berg compute in these
CPU tests is the oracle standing in for the GPU backend; what is under test is the sharding / reduction logic of
icebergs_amd.distributed, which bench.py drives with the HIP backend."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleBackend:
    def __init__(self, oracle, bergs):
        self.o, self.b = oracle, bergs

    def step_local(self):
        self.o.step_local(self.b)

    def step_gather(self):
        self.o.step_gather()


def _worker(rank, world, port, nbergs, nsteps, out_dir, find_melt=0):
    import oracle_lib
    from icebergs_amd import synthetic as S, distributed as D, types as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grid, p, b = S.config_c2(n=nbergs, seed=17, continents=True)
    p.find_melt_using_spread_mass = 1 if find_melt else 0
    p.Iceberg_melt_without_decay = 1 if find_melt == 2 else 0
    mine = D.take_shard(b, rank, world)
    o = oracle_lib.Oracle(grid, p)
    ncell = o.ni * o.nj
    # one contiguous block [scalars | planes], as the HIP handle lays it out (the scalars and the live planes are ONE all-reduce)
    block = np.zeros(T.NSCALAR + T.NACC * ncell)
    o.acc = block[T.NSCALAR:].reshape(T.NACC, o.nj, o.ni)
    o.scalars = block[: T.NSCALAR]
    spread_old = None
    if find_melt:   # grd%spread_mass_old (+ spread_mass_tmp): two more planes the ranks sum
        spread_old = np.zeros(2 * ncell)
        o.set_spread_mass_buffer(spread_old)
    stepper = D.ShardedStepper(OracleBackend(o, mine), torch.from_numpy(block), ncell, p.diag_mask, dist, params=p,
                               spread_mass_old=(torch.from_numpy(spread_old) if find_melt else None))
    totals = np.zeros(T.NSCALAR)
    for _ in range(nsteps):
        o.scalars[:] = 0.0
        stepper.step()
        totals += o.scalars
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), acc=o.acc, out=o.out, totals=totals,
             **{"b_" + k: v for k, v in mine.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    from icebergs_amd import distributed as D
    for n in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [2])
def test_two_rank_gloo_matches_single_process(tmp_path, world):
    import oracle_lib
    import parity as P
    from icebergs_amd import synthetic as S, types as T
    oracle_lib.build()
    nbergs, nsteps = 6000, 4
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, nbergs, nsteps, str(tmp_path)), nprocs=world, join=True)
    grid, p, b = S.config_c2(n=nbergs, seed=17, continents=True)
    rb, racc, rout, rscal = P.run_oracle(grid, p, b, nsteps)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for r in range(world):  # every rank holds the same reduced fields
        # with this namelist the gather reads only the melt/heat/bits planes and mass_on_ocean: those are what the
        # all-reduce carries (the area/Uvel/Vvel footprint planes stay rank-local partial sums nobody reads)
        for k in range(T.ENUMS["KID_A_MASS_ON_OCEAN"] + 9):
            assert P.rel_err(parts[r]["acc"][k], racc[k]) <= P.TOL_GRID, (r, k)
        for k in range(T.NOUT):
            assert P.rel_err(parts[r]["out"][k], rout[k]) <= P.TOL_GRID, (r, k)
        assert parts[r]["totals"][T.SCALAR_NAMES["nbergs_melted"]] == rscal[T.SCALAR_NAMES["nbergs_melted"]]
    for name in P.TRAJ_FIELDS + P.SIZE_FIELDS:  # the shards, put back together, are the single-process population
        got = np.concatenate([parts[r]["b_" + name] for r in range(world)])
        assert np.array_equal(got, rb[name]), name


@pytest.mark.parametrize("variant", [1, 2])
def test_two_rank_find_melt_using_spread_mass(tmp_path, variant):
    """find_melt_using_spread_mass in the sharded path (IB:5490-5503, 3436-3445): grd%spread_mass_old (variant 2: with
    Iceberg_melt_without_decay also spread_mass_tmp, IB:3411-3413) is summed over the ranks with the accumulator planes; the
    melt flux every rank ends with is the single-process one"""
    import oracle_lib
    import parity as P
    from icebergs_amd import synthetic as S, types as T
    oracle_lib.build()
    nbergs, nsteps, world = 5000, 3, 2
    port = 31500 + (os.getpid() % 2000) + variant
    mp.spawn(_worker, args=(world, port, nbergs, nsteps, str(tmp_path), variant), nprocs=world, join=True)
    grid, p, b = S.config_c2(n=nbergs, seed=17, continents=True)
    p.find_melt_using_spread_mass = 1
    p.Iceberg_melt_without_decay = 1 if variant == 2 else 0
    rb, racc, rout, rscal = P.run_oracle(grid, p, b, nsteps)
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    fm = T.ACC_NAMES["floating_melt"]
    assert np.abs(racc[fm]).max() > 0
    for r in range(world):
        assert P.rel_err(parts[r]["acc"][fm], racc[fm]) <= P.TOL_GRID, r
        for k in range(T.NOUT):
            assert P.rel_err(parts[r]["out"][k], rout[k]) <= P.TOL_GRID, (r, k)
    for name in P.TRAJ_FIELDS + P.SIZE_FIELDS:
        got = np.concatenate([parts[r]["b_" + name] for r in range(world)])
        assert np.array_equal(got, rb[name]), name


def test_sharded_find_melt_needs_its_planes():
    """find_melt_using_spread_mass in the sharded path reduces two more planes: the stepper refuses to run without them"""
    from icebergs_amd import synthetic as S
    from icebergs_amd.distributed import ShardedStepper, accumulator_views
    from icebergs_amd import types as T
    p = S.default_params()
    p.find_melt_using_spread_mass = 1
    block = torch.zeros(T.NACC * 16 + T.NSCALAR, dtype=torch.float64)
    with pytest.raises(ValueError):
        ShardedStepper(None, block, 16, 0, None, params=p)
    st = ShardedStepper(None, block, 16, 0, None, params=p, spread_mass_old=torch.zeros(32, dtype=torch.float64))
    assert st.spread_mass_old is not None
    planes, scalars = accumulator_views(block, 16, 0, p)
    assert scalars.numel() == T.NSCALAR and (planes.numel() - T.NSCALAR) % 16 == 0
    assert planes.data_ptr() == scalars.data_ptr() == block.data_ptr()   # one contiguous range from the head of the block

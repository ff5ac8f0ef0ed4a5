"""Domain-decomposed runs (icebergs_amd/decomposed.py): one tile and one rank each, bergs migrating between ranks with
torch.distributed point-to-point messages in the reference's wire format (send_bergs_to_other_pes FW:2997-3247).  On the CPU
the tiles are stepped by the oracle (what is under test is the exchange between processes); on the GPU by the HIP handles,
two ranks on one card."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

NI, NJ, DL = 16, 12, 0.02
FIELDS = ("lon", "lat", "uvel", "vvel", "mass", "thickness", "xi", "yj")


def _grid(tx, ntx):
    from icebergs_amd import synthetic as S
    g = S.c2_forcing(S.latlon_grid(ni=NI * ntx if tx is None else NI, nj=NJ, lon0=10.0 + (0 if tx is None else tx * NI * DL), dlon=DL, lat0=-60.0, dlat=DL))
    f, st = g["forcing"], g["static"]
    rad = np.pi / 180.0
    f["uo"][:] = 0.5 * np.cos(40.0 * st["lat"] * rad)
    f["vo"][:] = 0.2 * np.sin(30.0 * st["lon"] * rad)
    return g


RES = 2000.0    # the periodic Cartesian channel: cells of 2 km, Lx = the whole box


def _channel(tx, ntx):
    """a zonally periodic channel (the stand-alone driver's Cartesian grid with Lx = its width), or one tile of it; the flow
    is periodic in x, so a tile's halo across the seam holds what its neighbour's cells hold"""
    from icebergs_amd import synthetic as S
    Lx = ntx * NI * RES
    g = S.cartesian_grid(ni=NI * ntx if tx is None else NI, nj=NJ, gridres=RES, Lx=Lx)
    st, f = g["static"], g["forcing"]
    if tx:
        st["lon"] += tx * NI * RES
        st["lonc"] += tx * NI * RES
    w = 2.0 * np.pi / Lx
    f["uo"][:] = 0.6 + 0.2 * np.sin(w * st["lon"])
    f["vo"][:] = 0.05 * np.cos(2.0 * w * st["lon"]) * np.sin(np.pi * st["lat"] / (NJ * RES))
    f["ua"][:] = 4.0
    f["sst"][:] = 1.0 + np.sin(w * st["lonc"])
    f["sss"][:] = -1.0
    return g


def _population(ntx, n, cyclic=False):
    from icebergs_amd import synthetic as S
    whole = _channel(None, ntx) if cyclic else _grid(None, ntx)
    p = S.default_params()
    p.dt = 1800.0
    if cyclic:
        p.lat_ref, p.use_f_plane, p.periodic_reentry = -70.0, 1, 1      # the whole-grid run treats its seam as the boundary between two PEs
    return whole, p, S.place_bergs(whole, n, 3, (2, ntx * NI - 1), (3, NJ - 2))


class OracleTile:
    """the oracle behind the tile interface of icebergs_amd.decomposed (test infrastructure)"""

    def __init__(self, grid, p, bergs):
        import oracle_lib
        self.o, self.b = oracle_lib.Oracle(grid, p), bergs

    def evolve(self):
        from oracle_lib import _dp
        s = self.o.soa(self.b)
        self.o.lib.ko_evolve_icebergs(C.byref(self.o.kg), C.byref(self.o.params), C.byref(s), _dp(self.o.scalars))

    def thermo(self):
        from oracle_lib import _dp
        self.o.acc[:] = 0.0
        s = self.o.soa(self.b)
        self.o.lib.ko_thermodynamics(C.byref(self.o.kg), C.byref(self.o.params), C.byref(s), _dp(self.o.acc), _dp(self.o.scalars))

    def pack_pair(self, axis):
        return self.o.send_bergs(self.b, 2 * axis), self.o.send_bergs(self.b, 2 * axis + 1)

    def unpack_pair(self, from_lo, from_hi):
        assert self.o.unpack_bergs(self.b, from_lo) == 0 and self.o.unpack_bergs(self.b, from_hi) == 0

    def live(self):
        m = self.b["_n"]
        a = self.b["alive"][:m] != 0
        return {k: self.b[k][:m][a] for k in FIELDS + ("id",)}


class HipRun:
    def __init__(self, grid, p, bergs, cap):
        from icebergs_amd.decomposed import HipTile
        from icebergs_amd.framework import Icebergs
        self.ib = Icebergs(grid, p, capacity=cap)
        m = bergs["_n"]
        self.ib.upload_bergs({k: (v[:m].copy() if isinstance(v, np.ndarray) else v) for k, v in bergs.items() if k != "_n"})
        self.tile = HipTile(self.ib)
        self.pack_pair, self.unpack_pair = self.tile.pack_pair, self.tile.unpack_pair

    def _call(self, name):
        self.ib._check(getattr(self.ib.lib, name)(self.ib.h), name)

    def evolve(self):
        self._call("kid_zero_accumulators")
        self._call("kid_evolve_icebergs")

    def thermo(self):
        self._call("kid_thermodynamics")

    def live(self):
        b = self.ib.download_bergs()
        a = b["alive"] != 0
        return {k: b[k][a] for k in FIELDS + ("id",)}


def _worker(rank, world, port, backend, nbergs, nsteps, out_dir, cyclic=False):
    from icebergs_amd import synthetic as S
    from icebergs_amd.decomposed import TileExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    whole, p, b = _population(world, nbergs, cyclic)
    if cyclic:
        p.periodic_reentry = 0            # on a tile the seam is a real boundary between ranks
    sel = (b["ine"] - 1) // NI == rank
    cap = nbergs
    big = S.empty_bergs(cap)
    m = int(sel.sum())
    for k, v in b.items():
        if isinstance(v, np.ndarray):
            big[k][:m] = v[sel]
    big["ine"][:m] -= rank * NI
    big["_n"] = m
    g = _channel(rank, world) if cyclic else _grid(rank, world)
    tile = OracleTile(g, p, big) if backend == "oracle" else HipRun(g, p, big, cap)
    ex = TileExchange(world, 1, dist, cyclic_x=cyclic)
    assert ex.neighbour(1, 0) == ((rank + 1) % world if cyclic else (rank + 1 if rank + 1 < world else None)) and ex.neighbour(0, 1) is None
    for _ in range(nsteps):
        tile.evolve()
        ex.exchange(tile)
        tile.thermo()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), sent=ex.sent, received=ex.received, **tile.live())
    dist.barrier()
    dist.destroy_process_group()


def _check(tmp_path, world, nbergs, nsteps, tol, cyclic=False):
    import oracle_lib
    whole, p, b = _population(world, nbergs, cyclic)
    ref = OracleTile(whole, p, b)
    for _ in range(nsteps):
        ref.evolve()
        ref.thermo()
    b["_n"] = nbergs
    want = ref.live()
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert sum(int(q["sent"]) for q in parts) == sum(int(q["received"]) for q in parts) > 20
    ids = np.concatenate([q["id"] for q in parts])
    assert len(np.unique(ids)) == len(ids) and set(ids) == set(want["id"]) and (cyclic or len(ids) < nbergs)
    o1, o2 = np.argsort(want["id"]), np.argsort(ids)
    for f in ("lon", "lat", "uvel", "vvel", "mass", "thickness"):
        got = np.concatenate([q[f] for q in parts])
        assert np.allclose(want[f][o1], got[o2], rtol=tol, atol=tol * 1e-1), (f, float(np.abs(want[f][o1] - got[o2]).max()))


def test_two_ranks_gloo_oracle_tiles(tmp_path):
    import oracle_lib
    oracle_lib.build()
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, "oracle", 400, 30, str(tmp_path)), nprocs=2, join=True)
    _check(tmp_path, 2, 400, 30, 1e-12)


@pytest.mark.parametrize("world", [2, 3])
def test_cyclic_channel_gloo_oracle_tiles(tmp_path, world):
    """a zonally periodic channel cut into 2 (east and west neighbour are the same rank) or 3 tiles: bergs go round and round;
    against the undivided channel with periodic_reentry"""
    import oracle_lib
    oracle_lib.build()
    port = 30100 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, "oracle", 300, 60, str(tmp_path), True), nprocs=world, join=True)
    _check(tmp_path, world, 300, 60, 1e-11, cyclic=True)


def test_same_peer_swap_and_layout():
    """the rank arithmetic, and the message layout when east and west are the same rank (two tiles, cyclic)"""
    from icebergs_amd.decomposed import TileExchange

    class FakeDist:
        def __init__(self, rank, world): self.r, self.w = rank, world
        def get_rank(self): return self.r
        def get_world_size(self): return self.w
    ex = TileExchange(3, 2, FakeDist(4, 6))
    assert (ex.tx, ex.ty) == (1, 1) and ex.neighbour(1, 0) == 5 and ex.neighbour(-1, 0) == 3 and ex.neighbour(0, -1) == 1 and ex.neighbour(0, 1) is None
    assert TileExchange(3, 2, FakeDist(5, 6)).neighbour(1, 0) is None and TileExchange(3, 2, FakeDist(5, 6), cyclic_x=True).neighbour(1, 0) == 3
    assert TileExchange(2, 1, FakeDist(0, 2), cyclic_x=True).neighbour(1, 0) == 1 == TileExchange(2, 1, FakeDist(0, 2), cyclic_x=True).neighbour(-1, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("cyclic", [False, True])
def test_two_ranks_hip_tiles_on_one_gpu(tmp_path, cyclic):
    """the same two-rank runs with the HIP handles (both ranks on GPU 0, gloo for the messages), against the undivided oracle"""
    port = 29900 + (os.getpid() % 2000) + int(cyclic)
    nbergs, nsteps = (300, 60) if cyclic else (400, 30)
    mp.spawn(_worker, args=(2, port, "hip", nbergs, nsteps, str(tmp_path), cyclic), nprocs=2, join=True)
    _check(tmp_path, 2, nbergs, nsteps, 1e-9, cyclic=cyclic)

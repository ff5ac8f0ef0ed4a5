"""Berg migration between the handles of a decomposed domain (first slice): send_bergs_to_other_pes
(FW:2997-3247) with the reference's wire format (pack_berg_into_buffer2 FW:3250-3301, unpack_berg_from_buffer2 FW:3455-3680).
A 2 x 2 decomposition of a lat-lon box, exchanged east/west then north/south after evolve_icebergs as icebergs_run does
(IB:5433-5447), against the same bergs on the undivided grid."""
import ctypes as C

import numpy as np
import pytest

from icebergs_amd import synthetic as S
from icebergs_amd import types as T

pytestmark = pytest.mark.gpu

E, W, N, So = (T.ENUMS[k] for k in ("KID_DIR_E", "KID_DIR_W", "KID_DIR_N", "KID_DIR_S"))
NI, NJ, DL = 24, 20, 0.02          # cells per tile and their size in degrees (about 1 x 2 km: a berg changes cell every few steps)


def _grid(tx, ty, ntx, nty):
    g = S.latlon_grid(ni=NI * ntx if tx is None else NI, nj=NJ * nty if ty is None else NJ, lon0=10.0 + (0 if tx is None else tx * NI * DL), dlon=DL,
                      lat0=-60.0 + (0 if ty is None else ty * NJ * DL), dlat=DL)
    g = S.c2_forcing(g)
    f, st = g["forcing"], g["static"]
    rad = np.pi / 180.0                                    # a flow with structure at the scale of this box, analytic in (lon, lat)
    f["uo"][:] = 0.4 * np.cos(40.0 * st["lat"] * rad) + 0.2
    f["vo"][:] = 0.4 * np.sin(30.0 * st["lon"] * rad)
    f["ua"][:] = 6.0 * np.sin(25.0 * st["lat"] * rad)
    f["va"][:] = 5.0 * np.cos(35.0 * st["lon"] * rad)
    return g


def _phases(ibs, exchange, rebin=False):
    """one step of icebergs_run over all handles, phase by phase (IB:5125-5512), the exchange after evolve_icebergs.
    rebin: the reference's own order -- evolve_icebergs, move_berg_between_cells (IB:5437), then send_bergs_to_other_pes
    (IB:5447) -- with a count query in between: the bergs waiting to be packed are dead rows at that point and must
    survive the re-binning, the dead-tail drop of kid_num_bergs and the compaction"""
    def call(ib, name):
        ib._check(getattr(ib.lib, name)(ib.h), name)
    p = next(iter(ibs.values())).params
    for ib in ibs.values():
        call(ib, "kid_zero_accumulators")
        if not p.old_interp_flds_order:
            call(ib, "kid_interp_gridded_fields_to_bergs")
        call(ib, "kid_evolve_icebergs")
        if rebin:
            ib.move_berg_between_cells()
            ib.num_bergs()
            if rebin == "compact":
                ib.compact()
    sent = exchange(ibs)
    for ib in ibs.values():
        if not p.old_interp_flds_order:
            call(ib, "kid_interp_gridded_fields_to_bergs")
        call(ib, "kid_thermodynamics")
        call(ib, "kid_create_gridded_icebergs_fields")
    return sent


def _exchange(ntx, nty, pair=False):
    def run(ibs):
        sent = 0
        empty = np.empty((0, 34))
        for axis, (first, second, shift) in enumerate(((E, W, (1, 0)), (N, So, (0, 1)))):
            if pair:
                out = {key: ib.pack_emigrants_pair(axis) for key, ib in ibs.items()}
            else:
                out = {key: (ib.pack_emigrants(first), ib.pack_emigrants(second)) for key, ib in ibs.items()}
            for (tx, ty), ib in ibs.items():
                lo, hi = (tx - shift[0], ty - shift[1]), (tx + shift[0], ty + shift[1])
                from_lo, from_hi = out[lo][0] if lo in out else empty, out[hi][1] if hi in out else empty
                sent += len(from_lo) + len(from_hi)
                if pair:
                    ib.unpack_immigrants_pair(from_lo, from_hi)
                else:                                          # from the west / south neighbour first (FW:3064, 3160)
                    ib.unpack_immigrants(from_lo)
                    ib.unpack_immigrants(from_hi)
        return sent
    return run


@pytest.mark.parametrize("old_order,pair,rebin", [(1, False, False), (0, False, False), (1, True, False), (1, True, True), (1, False, "compact")])
def test_two_by_two_tiles_match_the_undivided_grid(old_order, pair, rebin):
    from icebergs_amd.framework import Icebergs
    ntx = nty = 2
    whole = _grid(None, None, ntx, nty)
    p = S.default_params()
    p.dt, p.old_interp_flds_order = 1800.0, old_order
    n = 6000
    b = S.place_bergs(whole, n, 11, (2, NI * ntx - 1), (2, NJ * nty - 1))
    ref = Icebergs(whole, p, capacity=n)
    ref.upload_bergs(b)
    # with re-binning / compaction between evolve and the exchange, a tile has room for little more than its own share:
    # the rows of the bergs that left must be reclaimed (> 500 arrivals per tile over the run, ~1500 residents)
    tile_cap = n if not rebin else 2300
    tiles = {}
    for tx in range(ntx):
        for ty in range(nty):
            g = _grid(tx, ty, ntx, nty)
            sel = ((b["ine"] - 1) // NI == tx) & ((b["jne"] - 1) // NJ == ty)
            bt = {k: (v[sel].copy() if isinstance(v, np.ndarray) else v) for k, v in b.items()}
            bt["ine"] = bt["ine"] - tx * NI
            bt["jne"] = bt["jne"] - ty * NJ
            ib = Icebergs(g, p, capacity=tile_cap)
            assert ib.buffer_width() == 34          # the decomposed host's first call (ice_bergs_framework_init, FW:1263): decomposed mode from here on
            ib.upload_bergs(bt)
            tiles[(tx, ty)] = ib
    assert ref.buffer_width() == 34
    moved = 0
    for _ in range(40):
        _phases({"whole": ref}, lambda ibs: 0)
        moved += _phases(tiles, _exchange(ntx, nty, pair), rebin)
    assert moved > 500, moved                                   # bergs did cross tile boundaries, corners included
    rb = ref.download_bergs()
    ra = rb["alive"] != 0
    parts = [ib.download_bergs() for ib in tiles.values()]
    tb = {k: np.concatenate([q[k][q["alive"] != 0] for q in parts]) for k in ("id", "lon", "lat", "uvel", "vvel", "mass", "thickness", "axn", "ayn", "bxn", "byn",
                                                                               "mass_of_bits", "heat_density", "start_mass")}
    assert len(np.unique(tb["id"])) == len(tb["id"])           # nobody was sent twice or kept on both sides
    assert int(ra.sum()) == len(tb["id"]) and int(ra.sum()) < n   # and the ones that left the box are gone on both sides
    o1, o2 = np.argsort(rb["id"][ra]), np.argsort(tb["id"])
    assert np.array_equal(rb["id"][ra][o1], tb["id"][o2])
    for f in ("lon", "lat", "uvel", "vvel", "mass", "thickness", "axn", "ayn", "bxn", "byn", "mass_of_bits", "heat_density", "start_mass"):
        x, y = rb[f][ra][o1], tb[f][o2]
        # rounding level, not equality: a tile's corner longitudes lon0 + dlon * i are rounded differently from the whole grid's
        assert np.allclose(x, y, rtol=1e-12, atol=1e-13), (f, float(np.abs(x - y).max()))
    # every live berg of a tile sits inside that tile's computational domain
    for q in parts:
        a = q["alive"] != 0
        assert np.all((q["ine"][a] >= 1) & (q["ine"][a] <= NI) & (q["jne"][a] >= 1) & (q["jne"][a] <= NJ))
    ref.close()
    for ib in tiles.values():
        ib.close()


def test_wire_format_and_refusals():
    """one berg through the buffer: the 34 reals of pack_berg_into_buffer2 in order, integers as reals, the id split in two"""
    from icebergs_amd.framework import Icebergs
    g = _grid(0, 0, 2, 2)
    p = S.default_params()
    b = S.place_bergs(g, 3, 5, (3, 6), (3, 6))
    b["ine"][1] = NI + 1                                        # already past the eastern edge: selected without a step
    b["lon"][1] = g["static"]["lon"][5, NI + S.HALO - 1] + 0.4 * DL
    b["id"][1] = (7 << 32) | 123456
    b["start_year"][1] = 1999
    ib = Icebergs(g, p, capacity=8)
    ib.upload_bergs(b)
    assert len(ib.pack_emigrants(W)) == 0
    buf = ib.pack_emigrants(E)
    assert buf.shape == (1, 34)
    r = buf[0]
    names = ["lon", "lat", "uvel", "vvel", "uvel_prev", "vvel_prev", "xi", "yj", "start_lon", "start_lat", "start_year", "start_day", "start_mass", "mass",
             "thickness", "width", "length", "fl_k", "mass_scaling", "mass_of_bits", "mass_of_fl_bits", "mass_of_fl_bergy_bits", "heat_density", "ine", "jne",
             "axn", "ayn", "bxn", "byn", "halo_berg", "static_berg"]
    for q, name in enumerate(names):
        assert r[q] == float(b[name][1]), name
    assert (r[31], r[32]) == (7.0, 123456.0) and r[33] == b["od"][1]
    assert ib.num_bergs()[1] == 2 and len(ib.pack_emigrants(E)) == 0        # consumed
    # the neighbour to the east takes it: its own cell index, xi / yj on its grid, *_old reset
    east = Icebergs(_grid(1, 0, 2, 2), p, capacity=8)
    east.upload_bergs({k: (v[:0].copy() if isinstance(v, np.ndarray) else v) for k, v in b.items()})
    east.unpack_immigrants(buf)
    e = east.download_bergs()
    assert e["alive"][0] == 1 and e["ine"][0] == 1 and e["jne"][0] == b["jne"][1] and e["id"][0] == (7 << 32) | 123456
    assert e["xi"][0] == pytest.approx(0.4, abs=1e-9) and e["lon_old"][0] == e["lon"][0] and e["uvel_old"][0] == e["uvel"][0] and e["start_year"][0] == 1999
    # a berg no cell of the data domain takes: dropped, and the call says so
    far = buf.copy()
    far[0, 0] += 30.0
    with pytest.raises(RuntimeError, match="can not find a cell"):
        east.unpack_immigrants(far)
    assert east.num_bergs()[1] == 1
    # bonded layouts are refused
    pb = S.params_copy(p)
    pb.iceberg_bonds_on, pb.interactive_icebergs_on, pb.Runge_not_Verlet = 1, 1, 0
    ib.set_params(pb)
    with pytest.raises(RuntimeError, match="without bonds"):
        ib.pack_emigrants(E)
    ib.close()
    east.close()


def test_tiles_against_the_oracle():
    """the same 2 x 2 exchange stepped by the CPU oracle (oracle/kid_oracle.c: ko_send_bergs, ko_unpack_bergs around
    ko_evolve_icebergs / ko_thermodynamics): per tile the same bergs with the same state, and the same records on the wire"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
    from oracle_lib import Oracle, _dp
    from icebergs_amd.framework import Icebergs
    ntx = nty = 2
    whole = _grid(None, None, ntx, nty)
    p = S.default_params()
    p.dt, p.old_interp_flds_order = 1800.0, 1
    n, cap = 3000, 3600
    b = S.place_bergs(whole, n, 23, (2, NI * ntx - 1), (2, NJ * nty - 1))
    tiles, orc, ob = {}, {}, {}
    for tx in range(ntx):
        for ty in range(nty):
            g = _grid(tx, ty, ntx, nty)
            sel = ((b["ine"] - 1) // NI == tx) & ((b["jne"] - 1) // NJ == ty)
            bt = {k: (v[sel].copy() if isinstance(v, np.ndarray) else v) for k, v in b.items()}
            bt["ine"] = bt["ine"] - tx * NI
            bt["jne"] = bt["jne"] - ty * NJ
            ib = Icebergs(g, p, capacity=cap)
            ib.upload_bergs(bt)
            tiles[(tx, ty)] = ib
            big = S.empty_bergs(cap)
            m = int(sel.sum())
            for k, v in bt.items():
                if isinstance(v, np.ndarray):
                    big[k][:m] = v
            big["_n"] = m
            ob[(tx, ty)] = big
            orc[(tx, ty)] = Oracle(g, p)

    def oracle_step():
        wire = []
        for key, o in orc.items():
            o.acc[:] = 0.0
            s = o.soa(ob[key])
            o.lib.ko_evolve_icebergs(C.byref(o.kg), C.byref(o.params), C.byref(s), _dp(o.scalars))
        for first, second, shift in ((0, 1, (1, 0)), (2, 3, (0, 1))):
            out = {key: (o.send_bergs(ob[key], first), o.send_bergs(ob[key], second)) for key, o in orc.items()}
            for (tx, ty), o in orc.items():
                lo, hi = (tx - shift[0], ty - shift[1]), (tx + shift[0], ty + shift[1])
                if lo in out:
                    assert o.unpack_bergs(ob[(tx, ty)], out[lo][0]) == 0
                if hi in out:
                    assert o.unpack_bergs(ob[(tx, ty)], out[hi][1]) == 0
            wire.append(out)
        for key, o in orc.items():
            s = o.soa(ob[key])
            o.lib.ko_thermodynamics(C.byref(o.kg), C.byref(o.params), C.byref(s), _dp(o.acc), _dp(o.scalars))
        return wire

    def hip_step():
        wire = []

        def exchange(ibs):
            for first, second, shift in ((E, W, (1, 0)), (N, So, (0, 1))):
                out = {key: (ib.pack_emigrants(first), ib.pack_emigrants(second)) for key, ib in ibs.items()}
                for (tx, ty), ib in ibs.items():
                    lo, hi = (tx - shift[0], ty - shift[1]), (tx + shift[0], ty + shift[1])
                    if lo in out:
                        ib.unpack_immigrants(out[lo][0])
                    if hi in out:
                        ib.unpack_immigrants(out[hi][1])
                wire.append(out)
            return 0
        _phases(tiles, exchange)
        return wire

    crossings = 0
    for step in range(16):
        wo, wh = oracle_step(), hip_step()
        for po, ph in zip(wo, wh):                               # the records on the wire, pass by pass, tile by tile
            for key in po:
                for a, c in zip(po[key], ph[key]):
                    assert a.shape == c.shape, (step, key, a.shape, c.shape)
                    if len(a):
                        ida = a[:, 31] * 2.0 ** 32 + a[:, 32]
                        idc = c[:, 31] * 2.0 ** 32 + c[:, 32]
                        a, c = a[np.argsort(ida)], c[np.argsort(idc)]
                        assert np.array_equal(a[:, [10, 23, 24, 31, 32]], c[:, [10, 23, 24, 31, 32]])      # the integers
                        assert np.allclose(a, c, rtol=1e-9, atol=1e-12), float(np.abs(a - c).max())
                        crossings += len(a)
    assert crossings > 100, crossings
    for key, ib in tiles.items():
        hb, o = ib.download_bergs(), ob[key]
        m = o["_n"]
        ha, oa = hb["alive"] != 0, o["alive"][:m] != 0
        o1, o2 = np.argsort(hb["id"][ha]), np.argsort(o["id"][:m][oa])
        assert np.array_equal(hb["id"][ha][o1], o["id"][:m][oa][o2]), key
        for f in ("ine", "jne", "start_year"):
            assert np.array_equal(hb[f][ha][o1], o[f][:m][oa][o2]), (key, f)
        for f in ("lon", "lat", "uvel", "vvel", "xi", "yj", "mass", "thickness", "width", "length", "axn", "ayn", "bxn", "byn", "mass_of_bits", "heat_density",
                  "lon_old", "uvel_old"):
            x, y = hb[f][ha][o1], o[f][:m][oa][o2]
            assert np.allclose(x, y, rtol=1e-9, atol=1e-11), (key, f, float(np.abs(x - y).max()))
        ib.close()


def test_pack_capacity_contract():
    """a buffer that is too small: KID_ECAPACITY, *n = how many wait, nothing consumed; the same call with room then gets all
    of them, once; the pair call keeps the two directions apart"""
    from icebergs_amd.framework import Icebergs, _dp
    g = _grid(0, 0, 2, 2)
    p = S.default_params()
    b = S.place_bergs(g, 12, 9, (3, 8), (3, 8))
    b["ine"][:5] = NI + 1                                       # five wait for the east, three for the west
    b["ine"][5:8] = 0
    ib = Icebergs(g, p, capacity=16)
    ib.upload_bergs(b)
    n = C.c_int64()
    small = np.zeros((2, 34))
    assert ib.lib.kid_pack_emigrants(ib.h, E, _dp(small), 2, C.byref(n)) == -4 and n.value == 5 and not small.any()
    assert ib.num_bergs()[1] == 12                               # nobody was deleted
    na, nb = C.c_int64(), C.c_int64()
    ea, we = np.zeros((8, 34)), np.zeros((2, 34))
    assert ib.lib.kid_pack_emigrants_pair(ib.h, 0, _dp(ea), 8, C.byref(na), _dp(we), 2, C.byref(nb)) == -4 and (na.value, nb.value) == (5, 3)
    assert ib.num_bergs()[1] == 12
    we = np.zeros((4, 34))
    assert ib.lib.kid_pack_emigrants_pair(ib.h, 0, _dp(ea), 8, C.byref(na), _dp(we), 4, C.byref(nb)) == 0 and (na.value, nb.value) == (5, 3)
    assert sorted(ea[:5, 32].astype(int)) == sorted(int(i) & 0xffffffff for i in b["id"][:5])
    assert sorted(we[:3, 32].astype(int)) == sorted(int(i) & 0xffffffff for i in b["id"][5:8])
    assert np.all(ea[:5, 23] == NI + 1) and np.all(we[:3, 23] == 0)
    assert ib.num_bergs()[1] == 4 and len(ib.pack_emigrants(E)) == 0 and len(ib.pack_emigrants(W)) == 0
    ib.close()

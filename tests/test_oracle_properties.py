"""Properties the oracle must satisfy where the reference holds no stored vector: internal
consistency of the cell search, the interpolation, the footprint weights and the melt bookkeeping."""
import ctypes as C

import numpy as np

from icebergs_amd import synthetic as S
from icebergs_amd import types as T


def _oracle(grid, p):
    import oracle_lib
    return oracle_lib.Oracle(grid, p)


def test_pos_within_cell_inverts_bilin_on_a_distorted_quad_grid(oracle):
    """calc_xiyj (FW:6439-6534) must invert the bilinear map of the corner coordinates (old_bug_bilin=F)."""
    grid, p, _ = S.config_c2(n=1)
    p.old_bug_bilin = 0
    rng = np.random.default_rng(1)
    st = grid["static"]
    st["lon"] += 0.15 * rng.uniform(-1, 1, st["lon"].shape)   # distort: general quadrilaterals
    st["lat"] += 0.10 * rng.uniform(-1, 1, st["lat"].shape)
    o = _oracle(grid, p)
    lonp = o._keep[T.GRID_STATIC_NAMES.index("lon")].ctypes.data_as(C.POINTER(C.c_double))
    latp = o._keep[T.GRID_STATIC_NAMES.index("lat")].ctypes.data_as(C.POINTER(C.c_double))
    for _ in range(300):
        i, j = int(rng.integers(5, 355)), int(rng.integers(5, 195))
        xi, yj = rng.uniform(0.01, 0.99, 2)
        x = oracle.ko_bilin(C.byref(o.kg), C.byref(p), lonp, i, j, xi, yj)
        y = oracle.ko_bilin(C.byref(o.kg), C.byref(p), latp, i, j, xi, yj)
        oxi, oyj, err = C.c_double(), C.c_double(), C.c_int(0)
        inside = oracle.ko_pos_within_cell(C.byref(o.kg), C.byref(p), x, y, i, j, C.byref(oxi), C.byref(oyj), C.byref(err))
        assert inside == 1 and err.value == 0
        assert abs(oxi.value - xi) < 1e-9 and abs(oyj.value - yj) < 1e-9


def test_cell_edges_belong_to_exactly_one_cell(oracle):
    """South and east edges belong to the cell, north and west do not (FW:6199-6206)."""
    grid, p, _ = S.config_c1()
    o = _oracle(grid, p)
    x, y = 5000.0, 7000.0  # the corner shared by cells (5,7),(6,7),(5,8),(6,8); lon(i,j) is the NE corner of cell (i,j)
    owners = [(i, j) for i in (5, 6) for j in (7, 8) if oracle.ko_is_point_in_cell(C.byref(o.kg), x, y, i, j)]
    assert len(owners) == 1
    owners = [(i, j) for i in (5, 6) for j in (8,) if oracle.ko_is_point_in_cell(C.byref(o.kg), x, 7400.0, i, j)]
    assert len(owners) == 1  # a point on a vertical edge


def test_footprint_weights_partition_unity(oracle):
    """spread_mass_across_ocean_cells: the nine weights sum to 1 on open ocean, rectangular and hexagonal."""
    grid, p, _ = S.config_c2(n=1)
    o = _oracle(grid, p)
    rng = np.random.default_rng(2)
    for hexa in (0, 1):
        for old in (0, 1):
            p.hexagonal_icebergs, p.use_old_spreading = hexa, old
            for _ in range(200):
                w = (C.c_double * 9)()
                ifu = C.c_double()
                area = float(rng.uniform(1e4, 3e8))
                oracle.ko_spread_weights(C.byref(o.kg), C.byref(p), 100, 100, float(rng.uniform(0, 1)), float(rng.uniform(0, 1)),
                                         area, 0.0, w, C.byref(ifu))
                assert abs(sum(w) - 1.0) < 1e-12 and min(w) >= -1e-12
                assert abs(ifu.value - 1.0) < 1e-12


def test_melt_bookkeeping(oracle):
    """One step: berg mass never grows, and the mass a cell receives (berg_melt * area * dt / mass_scaling summed) equals
    the mass the bergs of that cell lost (IB:3132-3133)."""
    grid, p, b = S.config_c2(n=3000, seed=4)
    o = _oracle(grid, p)
    b0 = S.copy_bergs(b)
    o.run_step(b, 1)
    assert np.all(b["mass"] <= b0["mass"] * (1 + 1e-15))
    d = grid["desc"]
    area = grid["static"]["area"]
    lost = np.zeros_like(area)
    alive0 = b0["alive"] != 0
    dm = (b0["mass"] - np.where(b["alive"] != 0, b["mass"], 0.0)) * b0["mass_scaling"]
    np.add.at(lost, (b["jne"][alive0] - d.jsd, b["ine"][alive0] - d.isd), dm[alive0])
    got = o.acc[T.ACC_NAMES["berg_melt"]] * area * p.dt
    assert np.allclose(got, lost, rtol=1e-9, atol=1e-3)


def test_rolling_keeps_volume_and_orders_width_length(oracle):
    p = S.default_params()
    rng = np.random.default_rng(3)
    for scheme in range(3):
        p.use_updated_rolling_scheme = 1 if scheme == 0 else 0
        p.tip_parameter = 1000.0 if scheme == 1 else 0.0
        for _ in range(300):
            t0, w0, l0 = rng.uniform(5, 400), rng.uniform(5, 400), rng.uniform(5, 400)
            t, w, l = C.c_double(t0), C.c_double(w0), C.c_double(l0)
            oracle.ko_rolling(C.byref(p), C.byref(t), C.byref(w), C.byref(l))
            assert abs(t.value * w.value * l.value - t0 * w0 * l0) <= 1e-9 * t0 * w0 * l0
            assert sorted([t.value, w.value, l.value]) == sorted([t0, w0, l0])

"""Pins the CPU oracle against the reference's OWN known-answer tests.

  hexagon_test            /root/reference/src/icebergs.F90:247-353
  point_in_triangle_test  /root/reference/src/icebergs.F90:226-244
  basal_melt_test         /root/reference/src/icebergs.F90:205-223 (inputs only; the printed values are a survey probe, not reference-held)
  unit_tests (bilin)      /root/reference/src/icebergs_framework.F90:7299-7327
"""
import ctypes as C
import math

import numpy as np
import pytest

from icebergs_amd import synthetic as S
from icebergs_amd import types as T


def hexq(lib, x0, y0, H, theta=0.0):
    out = [C.c_double() for _ in range(5)]
    lib.ko_hexagon_into_quadrants(x0, y0, H, theta, *[C.byref(o) for o in out])
    return [o.value for o in out]


def test_hexagon_test(oracle):
    tol = 1.0e-10  # IB:261
    H = 1.0
    Sd = 2.0 * H / math.sqrt(3.0)
    A, q1, q2, q3, q4 = hexq(oracle, 0.0, 0.0, H)          # test 1, IB:267-278
    assert abs(A - (3.0 * math.sqrt(3.0) / 2.0) * Sd * Sd) <= tol
    for q in (q1, q2, q3, q4):
        assert abs(A / 4 - q) <= tol
    A, q1, q2, q3, q4 = hexq(oracle, Sd, 0.0, H)           # 2a, IB:282-288
    assert abs(A / 2 - q1) <= tol and abs(q2) <= tol and abs(q3) <= tol and abs(A / 2 - q4) <= tol
    A, q1, q2, q3, q4 = hexq(oracle, -Sd, 0.0, H)          # 2b
    assert abs(A / 2 - q2) <= tol and abs(q1) <= tol and abs(q4) <= tol and abs(A / 2 - q3) <= tol
    A, q1, q2, q3, q4 = hexq(oracle, 0.0, H, H)            # 2c
    assert abs(A / 2 - q1) <= tol and abs(q3) <= tol and abs(q4) <= tol and abs(A / 2 - q2) <= tol
    A, q1, q2, q3, q4 = hexq(oracle, 0.0, -H, H)           # 2d
    assert abs(A / 2 - q3) <= tol and abs(q1) <= tol and abs(q2) <= tol and abs(A / 2 - q4) <= tol
    A, q1, q2, q3, q4 = hexq(oracle, Sd / 2, 0.0, H)       # 3a, IB:316-322
    assert abs(2.5 * A / 6 - q1) <= tol and abs(0.5 * A / 6 - q2) <= tol
    assert abs(0.5 * A / 6 - q3) <= tol and abs(2.5 * A / 6 - q4) <= tol
    A, q1, q2, q3, q4 = hexq(oracle, -Sd / 2, 0.0, H)      # 3b
    assert abs(2.5 * A / 6 - q2) <= tol and abs(0.5 * A / 6 - q1) <= tol
    assert abs(0.5 * A / 6 - q4) <= tol and abs(2.5 * A / 6 - q3) <= tol


def test_hexagon_partition_property(oracle):
    """Quadrant areas always sum to the hexagon area (IB:4653-4668 forces the residual to zero)."""
    rng = np.random.default_rng(0)
    for _ in range(2000):
        H = rng.uniform(0.01, 0.43)
        x0, y0 = rng.uniform(-0.5, 0.5, 2)
        A, q1, q2, q3, q4 = hexq(oracle, x0, y0, H, rng.uniform(0, 60))
        assert abs(A - (q1 + q2 + q3 + q4)) < 1e-12
        assert min(q1, q2, q3, q4) > -1e-10


def test_point_in_triangle_test(oracle):
    # IB:234-241: a near-degenerate triangle around the origin
    assert oracle.ko_point_in_triangle(-2.695732526092343E-012, 0.204344508198090,
                                       -2.695750202346321E-012, -8.433062639672301E-002,
                                       0.249999999997304, 6.000694090068343E-002, 0.0, 0.0) == 1


def test_basal_melt_test(oracle):
    """IB:214: dvo=0.2, lat=0, salt=35, temp=2, thickness=100 with namelist defaults (the inputs are reference-held).
    The two values are NOT a reference-held pin: the reference only prints them; the survey recorded them from a probe build
    against stand-in FMS modules.  Kept as a regression value of the oracle."""
    p = S.default_params()
    d = T.GridDesc()
    d.grid_is_latlon = 1
    two = oracle.ko_find_basal_melt(C.byref(d), C.byref(p), 0.2, 0.0, 35.0, 2.0, 0, 100.0)
    three = oracle.ko_find_basal_melt(C.byref(d), C.byref(p), 0.2, 0.0, 35.0, 2.0, 1, 100.0)
    assert two == pytest.approx(4.33063180897577E-06, rel=1e-13)
    assert three == pytest.approx(7.090487055660092E-06, rel=1e-13)


def test_bilin_corner_identities(oracle):
    """FW:7313-7316: with old_bug_bilin=F, bilin at (0,0),(1,0),(0,1),(1,1) returns the corner values."""
    import oracle_lib
    grid = S.cartesian_grid(4, 4)
    p = S.default_params()
    p.old_bug_bilin = 0
    o = oracle_lib.Oracle(grid, p)
    d = grid["desc"]
    fld = np.arange(o.ni * o.nj, dtype=np.float64).reshape(o.nj, o.ni) ** 1.5
    i, j = 2, 3
    fp = fld.ctypes.data_as(C.POINTER(C.c_double))

    def f(ii, jj):
        return fld[jj - d.jsd, ii - d.isd]
    assert oracle.ko_bilin(C.byref(o.kg), C.byref(p), fp, i, j, 0.0, 0.0) == f(i - 1, j - 1)
    assert oracle.ko_bilin(C.byref(o.kg), C.byref(p), fp, i, j, 1.0, 0.0) == f(i, j - 1)
    assert oracle.ko_bilin(C.byref(o.kg), C.byref(p), fp, i, j, 0.0, 1.0) == f(i - 1, j)
    assert oracle.ko_bilin(C.byref(o.kg), C.byref(p), fp, i, j, 1.0, 1.0) == f(i, j)
    p.old_bug_bilin = 1  # the default inverted weights mirror the cell (FW:7081-7083)
    assert oracle.ko_bilin(C.byref(o.kg), C.byref(p), fp, i, j, 0.0, 0.0) == f(i, j)


def test_modulo_semantics(oracle):
    """Fortran MODULO carries the sign of P (FW:6568)."""
    assert oracle.ko_modulo(-1.0, 360.0) == 359.0
    assert oracle.ko_modulo(361.0, 360.0) == 1.0
    assert oracle.ko_modulo(0.0, 360.0) == 0.0
    assert oracle.ko_apply_modulo_around_point(359.5, 0.25, 360.0) == pytest.approx(-0.5)
    assert oracle.ko_apply_modulo_around_point(359.5, 0.25, -1.0) == 359.5


def test_philox_known_answers(oracle):
    """include/kid_rng.h (the generator that places footloose children, in the oracle and in the HIP library alike) is
    Philox-4x32-10 as published: the known-answer vectors of the Random123 distribution (kat_vectors: zeros, ones, pi)."""
    import ctypes as C
    U = C.c_uint32

    def run(ctr, key):
        c, k, o = (U * 4)(*ctr), (U * 2)(*key), (U * 4)()
        oracle.ko_philox4x32_10(c, k, o)
        return [int(x) for x in o]
    assert run([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert run([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert run([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    # the uniform number built from it: 53 bits in [0, 1), and the same function on the product side of the C ABI
    from icebergs_amd import lib as L
    L.build()
    lib = L.load()
    seen = []
    for (seed, bid, step, draw) in [(1, 42, 0, 0), (1, 42, 1, 0), (20240807, (7 << 32) + 123, 39, 1), (-5, 0, 0, 0)]:
        a, b = oracle.ko_fl_uniform(seed, bid, step, draw), lib.kid_footloose_uniform(seed, bid, step, draw)
        assert a == b and 0.0 <= a < 1.0
        seen.append(a)
    assert len(set(seen)) == len(seen)
    u = np.array([oracle.ko_fl_uniform(7, k, 3, 0) for k in range(1, 4001)])
    assert abs(u.mean() - 0.5) < 0.02 and abs((u < 0.25).mean() - 0.25) < 0.03 and u.min() >= 0.0 and u.max() < 1.0

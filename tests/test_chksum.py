"""bergs_chksum (icebergs_framework.F90:6889-6987) on the structure of arrays: the routine itself.

The reference's regression tests record one line per run, "chksum=.. chksum2=.. chksum3=.. chksum4=.. chksum5=.. #=..", printed by
bergs_chksum at icebergs_save_restart (tests/collision_tests/README:13-22, the head of every tests/*/input*.nml).  Which of those
integers are reachable, and the tests that reproduce them (chksum3 = chksum4 and '#': functions of log(mass) and of the final
per-cell occupancy only), are in tests/test_reference_chksums.py.  chksum / chksum2 (bit patterns of the positions, velocities
and sizes of the bergs of the LAST occupied cell, out of another compiler's binary) and chksum5 (berg_chksum of the bergs in each
PE's last cell: 0 in most recorded lines, -81885495 / -845603363 in tests/dem_cbeam_test/input.nml:2,5 where the bent beam
crosses the last cell of a PE of the 4-PE run) depend on bit patterns of positions and stay out of reach.
Held here:
  * the routine, restated three times independently -- numpy below, the CPU oracle (ko_bergs_chksum) and the HIP library's host
    side (kid_bergs_chksum) -- must give identical integers on identical states;
  * its structure: the per-cell sum chksum5 starts over in every cell, so on one PE it is 0 whenever the last cell is empty.
"""
import numpy as np
import pytest

from icebergs_amd import synthetic as S


def _u32(x):
    return np.uint32(int(x) & 0xFFFFFFFF)


def _i32(u):
    u = int(u) & 0xFFFFFFFF
    return u - (1 << 32) if u >= (1 << 31) else u


def numpy_bergs_chksum(grid, b, n=None):
    """bergs_chksum written from the Fortran source (FW:6889-7068, 4364-4376) with plain Python integers"""
    d = grid["desc"]
    n = int(b.get("_n", len(b["lon"]))) if n is None else n
    rows = [k for k in range(n) if b["alive"][k] and d.isc <= b["ine"][k] <= d.iec and d.jsc <= b["jne"][k] <= d.jec]
    rows.sort(key=lambda k: (b["jne"][k], b["ine"][k], b["start_year"][k], b["start_day"][k], b["start_mass"][k], b["start_lon"][k], b["start_lat"][k], k))
    nb = max(len(rows), 1)
    fld, fld2 = np.zeros((nb, 19)), np.zeros((nb, 19))
    ni = d.ied - d.isd + 1
    tmp = np.zeros((d.jed - d.jsd + 1, ni))
    icnt = np.zeros_like(tmp, dtype=np.int64)
    src = ["lon", "lat", "uvel", "vvel", "mass", "thickness", "width", "length", "axn", "ayn", "bxn", "byn", "uvel_old", "vvel_old", "lon_old", "lat_old"]

    def berg_chksum(k):
        w = int(np.array([b["lon"][k]]).view(np.uint64)[0]) & 0xFFFFFFFF   # transfer(rtmp, i8): scalar mold -> the low word of rtmp(1)
        itmp = [w] * 36 + [int(b["halo_berg"][k]) & 0xFFFFFFFF, int(b["static_berg"][k]) & 0xFFFFFFFF, int(b["start_year"][k]) & 0xFFFFFFFF,
                           int(b["ine"][k]) & 0xFFFFFFFF, int(b["jne"][k]) & 0xFFFFFFFF, (int(b["id"][k]) >> 32) & 0xFFFFFFFF, int(b["id"][k]) & 0xFFFFFFFF]
        c1 = sum(itmp) & 0xFFFFFFFF
        c2 = sum(v * (q + 1) for q, v in enumerate(itmp)) & 0xFFFFFFFF
        c3 = sum(v * (q + 1) * (q + 1) for q, v in enumerate(itmp)) & 0xFFFFFFFF
        return _i32(c1 + c2 + c3)
    ichk5, cur, i = 0, None, 0
    by_cell = {}
    for k in rows:
        by_cell.setdefault((int(b["jne"][k]), int(b["ine"][k])), []).append(k)
    for gj in range(d.jsc, d.jec + 1):
        for gi in range(d.isc, d.iec + 1):
            i, ichk5 = 0, 0
            for k in by_cell.get((gj, gi), []):
                iberg = berg_chksum(k)
                th = b["start_day"][k] + 366.0 * float(b["start_year"][k])
                ph = b["start_lon"][k] + 360.0 * (b["start_lat"][k] + 90.0)
                fld[i, :16] = [b[f][k] for f in src]
                fld[i, 16:] = [th, ph, float(iberg)]
                icnt[gj - d.jsd, gi - d.isd] += 1
                fld2[i, :] = fld[i, :] * float(icnt[gj - d.jsd, gi - d.isd])
                tmp[gj - d.jsd, gi - d.isd] = (tmp[gj - d.jsd, gi - d.isd] + th * ph) + np.log(b["mass"][k])   # left to right, FW:6943
                ichk5 = (ichk5 + iberg) & 0xFFFFFFFF
                i += 1

    def mpp_chksum(a):
        return _i32(int(np.ascontiguousarray(a).view(np.uint64).sum(dtype=np.uint64)) & 0xFFFFFFFF)
    comp = tmp[d.jsc - d.jsd:d.jec - d.jsd + 1, d.isc - d.isd:d.iec - d.isd + 1]
    return (mpp_chksum(fld), mpp_chksum(fld2), mpp_chksum(tmp), mpp_chksum(comp), _i32(ichk5), len(rows))


def _population(seed=7, n=300):
    grid, p, b = S.config_c2(n=n, seed=seed)
    rng = np.random.default_rng(seed)
    b["ine"][:60] = 40                      # several bergs share cells: the row index of fld restarts per cell
    b["jne"][:60] = rng.integers(50, 54, 60)
    b["start_year"][:] = rng.integers(1, 4, n)
    b["start_day"][:] = rng.uniform(0, 300, n)
    b["id"][:] = (rng.integers(1, 9, n).astype(np.int64) << 32) + rng.integers(1, 70000, n)
    b["uvel"][:], b["vvel"][:] = rng.normal(0, 0.2, n), rng.normal(0, 0.2, n)
    b["alive"][5::17] = 0
    return grid, p, b


def test_oracle_bergs_chksum_against_the_source(oracle):
    import oracle_lib
    grid, p, b = _population()
    o = oracle_lib.Oracle(grid, p)
    got, want = o.bergs_chksum(b), numpy_bergs_chksum(grid, b)
    assert got == want, (got, want)
    assert got[5] == int((b["alive"] != 0).sum())
    # the structure of the routine: chksum5 is the per-cell sum of the LAST cell, 0 when that cell is empty (every recorded line)
    assert got[4] == 0
    d = grid["desc"]
    b["ine"][0], b["jne"][0], b["alive"][0] = d.iec, d.jec, 1
    assert o.bergs_chksum(b)[4] != 0 and o.bergs_chksum(b) == numpy_bergs_chksum(grid, b)
    # one berg per cell at most: fld holds one row and chksum2 == chksum (tests/collision_tests/README: KID and MTS_KID lines)
    grid, p, b1 = S.config_c1()
    o1 = oracle_lib.Oracle(grid, p)
    c = o1.bergs_chksum(b1)
    cells = set(zip(b1["ine"].tolist(), b1["jne"].tolist()))
    if len(cells) == len(b1["ine"]):
        assert c[0] == c[1]
    assert c[2] == c[3]     # all bergs on the computational domain: chksum3 == chksum4 (every recorded line)


def test_dem_ground_frac_generator_gives_the_recorded_count(oracle):
    """'#=69' (tests/dem_ground_frac_test/input.nml:7, 10): the generator's parameters restated, and a run of the oracle that
    keeps every element (the conglomerate fractures on the seamount, nobody melts or leaves)."""
    import oracle_lib
    xs, ys = S.dem_ground_frac_elements()
    assert len(xs) == 69
    grid, p, b, bd = S.config_c4(reference_pattern=True, sub_steps=20)
    assert len(b["lon"]) == 69
    o = oracle_lib.Oracle(grid, p)
    o.run_step_mts(b, bd, 6)
    c = o.bergs_chksum(b)
    assert c[5] == 69 and c[4] == 0 and c[2] == c[3]
    assert c == numpy_bergs_chksum(grid, b)


@pytest.mark.gpu
def test_hip_bergs_chksum(oracle):
    import oracle_lib
    from icebergs_amd.framework import Icebergs
    grid, p, b = _population(seed=11, n=5000)
    ib = Icebergs(grid, p, capacity=len(b["lon"]))
    try:
        ib.upload_bergs(b)
        assert ib.bergs_chksum() == numpy_bergs_chksum(grid, b) == oracle_lib.Oracle(grid, p).bergs_chksum(b)
        ib.run(3)                                   # rows re-binned, dead rows dropped: the line is a function of the state only
        got = ib.download_bergs()
        assert ib.bergs_chksum() == numpy_bergs_chksum(grid, got)
    finally:
        ib.close()


@pytest.mark.gpu
def test_hip_dem_ground_frac_count(oracle):
    """the same 69 elements through the HIP library: '#=69' after the run, chksum5 = 0, chksum3 = chksum4"""
    import parity as P
    grid, p, b, bd = S.config_c4(reference_pattern=True, sub_steps=50)
    got, gbd = P.run_hip_mts(grid, p, b, bd, 8)
    from icebergs_amd.framework import Icebergs
    ib = Icebergs(grid, p, capacity=len(b["lon"]))
    try:
        ib.upload_bergs(got[0])
        c = ib.bergs_chksum()
    finally:
        ib.close()
    assert c[5] == 69 and c[4] == 0 and c[2] == c[3]
    assert c == numpy_bergs_chksum(grid, got[0])

"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (SURVEY.md 8d / BASELINE.json): trajectories and berg sizes 1e-10 relative, per-cell fields 1e-9
relative to the field max; cell indices, the set of surviving bergs and event counters exact.
"""
import os

import numpy as np
import pytest

from icebergs_amd import synthetic as S
import parity as P

pytestmark = pytest.mark.gpu


def _both(grid, p, b, nsteps, mode):
    ref = P.run_oracle(grid, p, b, nsteps)
    got = P.run_hip(grid, p, b, nsteps, mode=mode)
    return ref, got


@pytest.mark.parametrize("mode", ["fused", "phases"])
def test_c1_rk4_defaults(oracle, mode):
    """BASELINE config 1: 10 bergs, 20x20 f-plane Cartesian grid, RK4 + old_bug_bilin=T, 144 steps of 600 s."""
    grid, p, b = S.config_c1()
    S.set_diag_all(p)
    ref, got = _both(grid, p, b, 144, mode)
    rep = P.compare(ref, got, "C1/rk4/" + mode, params=p)
    assert ref[0]["alive"].sum() >= 6  # most bergs stay on the 20x20 domain
    print({k: v for k, v in rep.items() if k in ("lon", "lat", "uvel", "mass")})


@pytest.mark.parametrize("mode", ["fused", "phases"])
def test_c1_verlet_newbilin(oracle, mode):
    """Config 1 run the second way SURVEY 8d asks for: Verlet + old_bug_bilin=F."""
    grid, p, b = S.config_c1()
    p.Runge_not_Verlet = 0
    p.old_bug_bilin = 0
    S.set_diag_all(p)
    ref, got = _both(grid, p, b, 144, mode)
    P.compare(ref, got, "C1/verlet/" + mode, params=p)


def test_c1_verlet_new_interp_order(oracle):
    """.not.old_interp_flds_order: environment interpolated once before evolve and once before thermodynamics
    (IB:5423, 5473), as the footloose/MTS profiles run."""
    grid, p, b = S.config_c1()
    p.Runge_not_Verlet = 0
    p.old_bug_bilin = 0
    p.old_interp_flds_order = 0
    p.use_new_predictive_corrective = 1
    for mode in ("fused", "phases"):
        ref, got = _both(grid, p, b, 60, mode)
        P.compare(ref, got, "C1/verlet-neworder/" + mode, params=p)


@pytest.mark.parametrize("verlet", [False, True])
def test_c2_new_interp_order_fused(oracle, verlet):
    """.not.old_interp_flds_order at config-2 size with coasts: interpolate, evolve, interpolate again, melt and spread as
    ONE launch per step (the second interpolation sits between the phases of berg_kernel), against the oracle's four
    separate sweeps"""
    grid, p, b = S.config_c2(n=30000, seed=41, continents=True)
    S.set_diag_all(p)
    p.old_interp_flds_order = 0
    if verlet:
        p.Runge_not_Verlet, p.use_new_predictive_corrective = 0, 1
    ref, got = _both(grid, p, b, 12, "fused")
    P.compare(ref, got, "C2/new-order/verlet=%s" % verlet, params=p)


@pytest.mark.parametrize("continents", [False, True])
def test_c2_latlon(oracle, continents):
    """BASELINE config 2 at an oracle-sized population: lat-lon 360x200 (calc_xiyj path), RK4, melt,
    rectangular mass spreading; `continents` adds land rectangles so that bergs bounce off coasts."""
    grid, p, b = S.config_c2(n=40000, seed=2, continents=continents)
    ref, got = _both(grid, p, b, 48, "fused")   # 48 steps of 1800 s (one model day)
    P.compare(ref, got, "C2/continents=%s" % continents, params=p)


def test_c2_phases_equals_fused(oracle):
    grid, p, b = S.config_c2(n=20000, seed=7)
    S.set_diag_all(p)
    a = P.run_hip(grid, p, b, 5, mode="fused")
    c = P.run_hip(grid, p, b, 5, mode="phases")
    for f in P.TRAJ_FIELDS + P.SIZE_FIELDS:
        assert np.array_equal(a[0][f], c[0][f]), f  # same device arithmetic -> bitwise equal state (5 steps: no re-binning yet)
    for k in range(a[1].shape[0]):
        assert P.rel_err(a[1][k], c[1][k]) <= 1e-12


def test_c2_bergy_bits_and_rolling_schemes(oracle):
    grid, p, b = S.config_c2(n=20000, seed=11)
    p.bergy_bit_erosion_fraction = 0.4
    p.use_updated_rolling_scheme = 1
    p.use_old_spreading = 0
    p.speed_limit = 0.05
    S.set_diag_all(p)
    ref, got = _both(grid, p, b, 8, "fused")
    P.compare(ref, got, "C2/bergy-bits", params=p)
    assert ref[3][1] >= 0  # counters compared exactly inside compare()


def test_melting_to_death_and_compaction(oracle):
    """Small warm-water bergs melt completely: deletion (IB:3271-3296) and SoA compaction."""
    from icebergs_amd.framework import Icebergs
    grid, p, b = S.config_c2(n=5000, seed=5)
    grid["forcing"]["sst"][:] = 12.0
    b["mass"] *= 1e-3
    b["thickness"][:] = 8.0
    b["width"] = np.sqrt(b["mass"] / (1.5 * 850.0 * b["thickness"]))
    b["length"] = 1.5 * b["width"]
    p.dt = 86400.0
    ref = P.run_oracle(grid, p, b, 10)
    got = P.run_hip(grid, p, b, 10, mode="fused")
    P.compare(ref, got, "melt-to-death", params=p)
    assert ref[3][1] > 0, "the test must actually melt some bergs"
    ib = Icebergs(grid, p, capacity=len(b["lon"]))
    ib.set_resort_interval(0)
    ib.upload_bergs(b)
    ib.run(10)
    slots, alive = ib.num_bergs()
    ib.compact()
    slots2, alive2 = ib.num_bergs()
    assert slots2 == alive == alive2 < slots
    cb = ib.download_bergs()
    keep = got[0]["alive"] != 0  # run_hip did 10 steps: below the default re-binning interval, order untouched
    assert np.array_equal(cb["id"], got[0]["id"][keep])  # compaction keeps the order
    assert np.array_equal(cb["lon"], got[0]["lon"][keep])
    ib.close()


def test_empty_and_ragged_populations(oracle):
    """Edge cases: a population that is not a multiple of the wave/block size, a single berg, no bergs."""
    from icebergs_amd.framework import Icebergs
    for n in (1, 63, 65, 257):
        grid, p, b = S.config_c2(n=n, seed=100 + n)
        ref, got = _both(grid, p, b, 3, "fused")
        P.compare(ref, got, "ragged n=%d" % n, params=p)
    grid, p, b = S.config_c2(n=1, seed=3)
    ib = Icebergs(grid, p, capacity=8)
    e = S.empty_bergs(0)
    ib.upload_bergs(e)
    ib.run(2)
    acc, out, scal = ib.fetch()
    assert not acc.any() and not out.any()
    ib.close()


def test_hexagonal_spreading(oracle):
    grid, p, b = S.config_c2(n=8000, seed=21)
    p.hexagonal_icebergs = 1
    p.initial_orientation = 10.0
    ref, got = _both(grid, p, b, 3, "fused")
    P.compare(ref, got, "hexagonal", params=p)


def test_move_berg_between_cells(oracle):
    """Re-binning (IB:5437) is a stable sort by cell that drops dead bergs; results do not depend on how often it runs."""
    from icebergs_amd.framework import Icebergs
    grid, p, b = S.config_c2(n=30000, seed=13, continents=True)
    rng = np.random.default_rng(0)
    shuffle = rng.permutation(len(b["lon"]))          # start from a completely unsorted population
    b = {k: np.ascontiguousarray(v[shuffle]) for k, v in b.items()}
    b["alive"][::97] = 0                                # and some bergs that are already gone
    ref = P.run_oracle(grid, p, b, 6)
    res = []
    for interval in (0, 1, 4):
        ib = Icebergs(grid, p, capacity=len(b["lon"]))
        ib.set_resort_interval(interval)
        ib.upload_bergs(b)
        ib.run(6)
        acc, out, scal = ib.fetch()
        got = (ib.download_bergs(), acc.copy(), out.copy(), scal.copy())
        P.compare(ref, got, "resort interval %d" % interval, params=p)
        res.append(got)
        if interval == 1:
            gb = got[0]
            d = grid["desc"]
            key = (gb["jne"].astype(np.int64) - d.jsd) * (d.ied - d.isd + 1) + (gb["ine"] - d.isd)
            assert gb["alive"].all() and len(gb["lon"]) == int(ref[0]["alive"].sum())
            # sorted by the cell the bergs were in when the last re-binning ran (the last step moved a few on)
            assert np.mean(np.diff(key) >= 0) > 0.97
        ib.close()
    order = [np.argsort(r[0]["id"][r[0]["alive"] != 0]) for r in res]
    for f in P.TRAJ_FIELDS + P.SIZE_FIELDS:
        base = res[0][0][f][res[0][0]["alive"] != 0][order[0]]
        for r, o in zip(res[1:], order[1:]):
            assert np.array_equal(base, r[0][f][r[0]["alive"] != 0][o]), f


@pytest.mark.parametrize("style", ["fl_bits", "new_bergs"])
@pytest.mark.parametrize("mode", ["fused", "phases"])
def test_c3_footloose(oracle, style, mode):
    """BASELINE config 3 (tests/footloose_tests profile): Verlet + footloose calving into FL bits (with new bergs
    spawned from the bits above 3e11 kg) or directly into child bergs; children are appended to the SoA with ids
    from the per-cell counter (generate_id)."""
    grid, p, b = S.config_c3(n=400, seed=3, fl_style=style)
    S.set_diag_all(p)
    ref, got = _both(grid, p, b, 40, mode)
    P.compare(ref, got, "C3/%s/%s" % (style, mode), params=p)
    from icebergs_amd import types as T
    ncalved = ref[3][T.SCALAR_NAMES["nbergs_calved_fl"]]
    assert ncalved >= 5, ncalved
    assert ref[0]["_n"] == len(got[0]["lon"]) or (got[0]["alive"] != 0).sum() == (ref[0]["alive"][:ref[0]["_n"]] != 0).sum()


@pytest.mark.parametrize("style,mode,periodic,by_pe", [("new_bergs", "fused", False, False), ("new_bergs", "phases", True, False),
                                                      ("fl_bits", "fused", True, False), ("new_bergs", "fused", False, True)])
def test_c3_footloose_displaced(oracle, style, mode, periodic, by_pe):
    """config 3 with its own namelist's displace_fl_bergs=T (the reference default, tests/footloose_tests/input.nml:32): a child
    berg is put at a random place on its parent's perimeter (IB:2688-2732), in whatever cell that is (find_cell, the corner and
    grounded-cell fall-backs, pos_within_cell: IB:6432-6478).  The random number is the counter-based generator of
    include/kid_rng.h on both sides, so positions, cells and everything downstream must agree with the oracle."""
    grid, p, b = S.config_c3(n=400, seed=3, fl_style=style, displace=True, periodic=periodic, by_pe=by_pe)
    S.set_diag_all(p)
    ref, got = _both(grid, p, b, 40, mode)
    P.compare(ref, got, "C3-displaced/%s/%s" % (style, mode), params=p)
    from icebergs_amd import types as T
    assert ref[3][T.SCALAR_NAMES["nbergs_calved_fl"]] >= 5
    rb, n = ref[0], ref[0]["_n"]
    child = (rb["id"][:n] >= (1 << 32)) & (rb["alive"][:n] != 0)
    assert child.sum() >= 5
    # the children really are displaced: their start position is not any parent's position of that moment on a 1 m scale
    moved = np.hypot(rb["start_lon"][:n][child] - rb["lon_old"][:n][child], rb["start_lat"][:n][child] - rb["lat_old"][:n][child])
    assert np.isfinite(moved).all()
    if not by_pe:
        # different events draw different numbers: the children do not all sit on the same side of their parents
        par = {int(i): k for k, i in enumerate(rb["id"][:n])}
        assert len(np.unique(np.round(rb["start_lon"][:n][child] % 1000.0, 3))) > 3


def test_c3_child_ids_are_deterministic(oracle):
    """A13, id assignment order: two children of one cell calved in the same step get the counter values in the order the
    reference's loop meets their parents (cell list order: `inorder`), not in the order the lanes' atomics land.  A population
    packed into a few cells, calving heavily: the ids must equal the oracle's one for one (compare() matches bergs by id), and two
    runs of the library must give identical id arrays."""
    from icebergs_amd.framework import Icebergs
    grid, p, b = S.config_c3(n=600, seed=11, fl_style="new_bergs", displace=True)
    n = b["_n"]
    b["ine"][:n] = 10 + (np.arange(n) % 3)          # three cells hold all the parents: many events per cell and step
    b["jne"][:n] = 8
    d = grid["desc"]
    gridres = 1000.0
    b["lon"][:n] = gridres * (b["ine"][:n] - 1) + gridres * b["xi"][:n]
    b["lat"][:n] = gridres * (b["jne"][:n] - 1) + gridres * b["yj"][:n]
    for f, g_ in (("lon_old", "lon"), ("lat_old", "lat"), ("start_lon", "lon"), ("start_lat", "lat")):
        b[f][:n] = b[g_][:n]
    b = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in b.items()}
    order = np.lexsort((b["start_lat"][:n], b["start_lon"][:n], b["start_mass"][:n], b["start_day"][:n], b["start_year"][:n], b["ine"][:n], b["jne"][:n]))
    for k, v in b.items():
        if hasattr(v, "dtype"):
            v[:n] = v[:n][order]
    ref, got = _both(grid, p, b, 30, "fused")
    P.compare(ref, got, "C3/child ids", params=p)
    rb, nr = ref[0], ref[0]["_n"]
    child = (rb["id"][:nr] >= (1 << 32)) & (rb["alive"][:nr] != 0)
    cnt = (rb["id"][:nr][child] >> 32)
    assert child.sum() >= 20 and cnt.max() >= 3      # several children per cell: the order of the counter values mattered
    runs = []
    for rep in range(2):
        g2 = P.run_hip(grid, p, b, 30, mode="fused")[0]
        o = np.lexsort((g2["start_lat"], g2["start_lon"], g2["start_day"], g2["mass_scaling"], g2["lon"]))
        runs.append((g2["id"][o].copy(), g2["lon"][o].copy()))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])


def test_c3_footloose_capacity_error(oracle):
    """children need spare rows: without them kid_footloose_calving reports KID_ECAPACITY instead of writing
    past the arrays"""
    from icebergs_amd.framework import Icebergs
    from icebergs_amd.lib import KidError
    grid, p, b = S.config_c3(n=200, seed=4, fl_style="new_bergs")
    ib = Icebergs(grid, p, capacity=200, device=0)
    try:
        ib.upload_bergs(b)
        with pytest.raises(KidError) as e:
            ib.run(40)
        assert "capacity" in str(e.value)
    finally:
        ib.close()


def test_c3_iceberg_counter_roundtrip(oracle):
    grid, p, b = S.config_c3(n=50, seed=5, fl_style="new_bergs")
    from icebergs_amd.framework import Icebergs
    ib = Icebergs(grid, p, capacity=len(b["lon"]), device=0)
    try:
        c0 = np.arange(ib.ni * ib.nj, dtype=np.int32).reshape(ib.nj, ib.ni)
        ib.set_iceberg_counter(c0)
        assert np.array_equal(ib.get_iceberg_counter(), c0)
        ib.upload_bergs(b)
        ib.run(30)
        c1 = ib.get_iceberg_counter()
        got = ib.download_bergs()
        nchild = int((got["id"] >= (1 << 32)).sum())
        assert nchild > 0 and int((c1 - c0).sum()) >= nchild  # every child drew one counter value
        assert len(np.unique(got["id"])) == len(got["id"])
    finally:
        ib.close()


def test_store_environment_off(oracle):
    """kid_set_store_environment(0): same trajectories, sizes and fields; berg%uo..hi are left as uploaded"""
    from icebergs_amd.framework import Icebergs
    from icebergs_amd.lib import KidError
    grid, p, b = S.config_c2(n=20000, seed=11)
    b["uo"][:] = 123.0
    ref = P.run_hip(grid, p, b, 6)
    ib = Icebergs(grid, p, capacity=len(b["lon"]), device=0)
    try:
        ib.upload_bergs(b)
        ib.set_store_environment(False)
        ib.run(6)
        acc, out, scal = ib.fetch()
        got = ib.download_bergs()
        for f in P.TRAJ_FIELDS + P.SIZE_FIELDS:
            o1, o2 = np.argsort(ref[0]["id"]), np.argsort(got["id"])
            assert np.array_equal(ref[0][f][o1], got[f][o2]), f
        assert np.array_equal(acc, ref[1]) or P.rel_err(acc, ref[1]) < 1e-12
        assert np.all(got["uo"] == 123.0) and not np.all(ref[0]["uo"] == 123.0)
    finally:
        ib.close()
    # .not.old_interp_flds_order: the stored environment is an input of the phase-by-phase entry points, not of the fused step
    q = S.params_copy(p)
    q.old_interp_flds_order, q.Runge_not_Verlet, q.use_new_predictive_corrective = 0, 0, 1
    ref = P.run_hip(grid, q, b, 6)
    ib = Icebergs(grid, q, capacity=len(b["lon"]), device=0)
    try:
        ib.upload_bergs(b)
        ib.set_store_environment(False)
        ib.run(6)
        acc, out, scal = ib.fetch()
        got = ib.download_bergs()
        o1, o2 = np.argsort(ref[0]["id"]), np.argsort(got["id"])
        for f in P.TRAJ_FIELDS + P.SIZE_FIELDS:
            assert np.array_equal(ref[0][f][o1], got[f][o2]), f
        assert np.array_equal(acc, ref[1]) or P.rel_err(acc, ref[1]) < 1e-12
        assert np.all(got["uo"] == 123.0) and not np.all(ref[0]["uo"] == 123.0)
        with pytest.raises(KidError):
            ib.run_phases(1)     # kid_evolve_icebergs reads the stored environment
        ib.set_store_environment(True)
    finally:
        ib.close()


@pytest.mark.parametrize("case", ["hex_free", "hex_grounded", "square_free", "two_bergs", "thickness_jitter",
                                  "kid_explicit", "kid_implicit", "kid_two_bergs", "sts_kid", "sts_kid_contact"])
def test_c4_mts_dem(oracle, case):
    """BASELINE config 4 family at oracle size: bonded DEM elements under MTS velocity Verlet (200 explicit sub-steps
    per step), grounding on a seamount with stress fracture, square and hexagonal packing, a collision between two
    conglomerates, and elements of unequal thickness (where the side that evaluates a bond pair matters)."""
    kw = {"hex_free": dict(bump=(150e3, 150e3)), "hex_grounded": dict(),
          "square_free": dict(bump=(150e3, 150e3), hexagonal=False, nx=6, ny=6),
          "two_bergs": dict(bump=(150e3, 150e3), two_bergs=True, hexagonal=False, nx=4, ny=6),
          "thickness_jitter": dict(bump=(150e3, 150e3), thickness_jitter=0.2),
          # MTS without DEM (tests/collision_tests/input_MTS_KID.nml): KID springs on the bonds, explicit inner steps, or
          # the implicit inner iteration with force_convergence; then two conglomerates colliding
          "kid_explicit": dict(bump=(150e3, 150e3), dem=False, explicit_inner=True, spring_coef=1e-5, sub_steps=120, dt=3600.0),
          "kid_implicit": dict(bump=(150e3, 150e3), dem=False, explicit_inner=False, spring_coef=1e-5, sub_steps=20, dt=1800.0),
          "kid_two_bergs": dict(bump=(150e3, 150e3), dem=False, explicit_inner=True, spring_coef=1e-5, sub_steps=60, dt=3600.0,
                                two_bergs=True, hexagonal=False, nx=4, ny=6),
          # interacting bergs WITHOUT multiple time stepping (tests/collision_tests/input_KID.nml): springs and implicit
          # damping inside accel; Stern et al.'s original 3x3 search, then the contact-distance variant with two bergs
          "sts_kid": dict(bump=(150e3, 150e3), dem=False, mts=False, contact=False, spring_coef=1e-5, dt=60.0),
          "sts_kid_contact": dict(bump=(150e3, 150e3), dem=False, mts=False, contact=True, spring_coef=1e-5, dt=60.0,
                                  two_bergs=True, hexagonal=False, nx=4, ny=6)}[case]
    grid, p, b, bd = S.config_c4(**kw)
    S.set_diag_all(p)
    nsteps = 6 if p.mts else 100
    ref, refbd = P.run_oracle_mts(grid, p, b, bd, nsteps)
    got, gotbd = P.run_hip_mts(grid, p, b, bd, nsteps)
    rep = P.compare_mts(ref, refbd, got, gotbd, "C4/" + case)
    print(case, {k: "%.1e" % v for k, v in rep.items() if v > 0})
    if case in ("hex_grounded", "two_bergs"):
        assert (refbd["broken"] != 0).sum() > 0  # the case does fracture


@pytest.mark.parametrize("case", ["hex_grounded", "two_bergs"])
def test_c4_fused_substeps_match_three_launches(oracle, case, monkeypatch):
    """The sub-step loop in one cooperative launch (mts_substeps_kernel: records exchanged point to point, the pair force in
    pieces) against the same loop as three launches per sub-step (KID_MTS_NO_FUSED): the same bonds break in the same
    sub-steps, conglomerates come out the same, states agree to the DEM tolerances."""
    kw = {"hex_grounded": dict(), "two_bergs": dict(bump=(150e3, 150e3), two_bergs=True, hexagonal=False, nx=4, ny=6)}[case]
    grid, p, b, bd = S.config_c4(**kw)
    S.set_diag_all(p)
    fused, fusedbd = P.run_hip_mts(grid, p, b, bd, 6)
    monkeypatch.setenv("KID_MTS_NO_FUSED", "1")
    plain, plainbd = P.run_hip_mts(grid, p, b, bd, 6)
    P.compare_mts(plain, plainbd, fused, fusedbd, "C4 fused vs three launches/" + case)
    if case == "hex_grounded":
        assert (fusedbd["broken"] != 0).sum() > 0


def test_c4_fused_substeps_fall_back_when_they_do_not_fit(oracle, monkeypatch):
    """the fused sub-step kernel needs every berg's lane co-resident; a population beyond what fits (here: a cap of one workgroup,
    KID_MTS_FUSED_BLOCKS_CAP, on 400 elements = two workgroups) must take the three-launch path by itself and give the same answer"""
    grid, p, b, bd = S.config_c4(nx=20, ny=20, hexagonal=False, radius=1500.0, ni=45, nj=45, sub_steps=40, origin=(40137.0, 35211.0), bump=(150e3, 150e3))
    S.set_diag_all(p)   # (the seamount out of the way: a conglomerate that shatters amplifies rounding beyond the DEM tolerances)
    assert len(b["lon"]) == 400
    fused, fusedbd = P.run_hip_mts(grid, p, b, bd, 4)
    monkeypatch.setenv("KID_MTS_FUSED_BLOCKS_CAP", "1")
    capped, cappedbd = P.run_hip_mts(grid, p, b, bd, 4)
    P.compare_mts(capped, cappedbd, fused, fusedbd, "C4 fused vs capped (fall-back)")
    monkeypatch.delenv("KID_MTS_FUSED_BLOCKS_CAP")
    monkeypatch.setenv("KID_MTS_NO_FUSED", "1")
    plain, plainbd = P.run_hip_mts(grid, p, b, bd, 4)
    for f in ("lon", "lat", "uvel", "vvel", "rot"):   # the capped run IS the three-launch path: bit for bit
        assert np.array_equal(capped[0][f], plain[0][f]), f


def test_c4_fused_substeps_time_out_loudly(oracle, monkeypatch):
    """a lane of the fused kernel that gives up waiting for a neighbour's record (here: a poll limit of zero spins) must not
    leave the step half applied in silence: the rows hold a finite state and the next synchronisation returns an error that
    names the time-out"""
    from icebergs_amd.framework import Icebergs
    from icebergs_amd import lib as L
    grid, p, b, bd = S.config_c4(nx=20, ny=20, hexagonal=False, radius=1500.0, ni=45, nj=45, sub_steps=90, origin=(40137.0, 35211.0))
    monkeypatch.setenv("KID_MTS_POLL_LIMIT", "1")
    ib = Icebergs(grid, p, capacity=len(b["lon"]))
    try:
        ib.upload_bergs(b); ib.upload_bonds(bd)
        with pytest.raises(L.KidError, match="timed out"):
            ib.run(1)
            ib.sync()
            ib.run(1)      # (the time-out of a step is reported at the latest when the next one starts)
            ib.sync()
        got = ib.download_bergs()
        for f in ("lon", "lat", "uvel", "vvel", "rot", "ang_vel"):
            assert np.isfinite(got[f]).all(), f
    finally:
        ib.close()


@pytest.mark.parametrize("split_general", [False, True, "slow_lane", "slow_lane_diag", "slow_lane_verlet", "slow_lane_new_order"])
def test_pipelined_stepper_matches_plain(oracle, split_general):
    """PipelinedStepper (two accumulator blocks, exchange + gather on a second stream under the next step's kernels)
    must give what the plain sequence gives; run here on one GPU, with and without a (world-size-1) RCCL all-reduce."""
    import torch
    from icebergs_amd.framework import Icebergs
    from icebergs_amd.distributed import ShardedStepper, PipelinedStepper
    from icebergs_amd import types as T
    grid, p, b = S.config_c2(n=30000, seed=21, continents=True)
    slow_lane = isinstance(split_general, str)
    if split_general == "slow_lane_diag":   # every diagnostic plane on: the gather reads the forcing records too
        S.set_diag_all(p)
    if split_general == "slow_lane_verlet":
        p.Runge_not_Verlet = 0
    if split_general == "slow_lane_new_order":   # .not.old_interp_flds_order: the stored environment travels with the berg
        p.Runge_not_Verlet, p.old_interp_flds_order, p.use_new_predictive_corrective = 0, 0, 1
    nsteps = 37
    dev = torch.device("cuda", 0)
    forcing_dev = [torch.from_numpy(np.ascontiguousarray(grid["forcing"][name])).to(dev) for name in T.FORCING_NAMES]
    ptrs = [t.data_ptr() for t in forcing_dev]
    # a second forcing set, used on odd steps: launches of step k still in flight must keep seeing step k's records
    forcing_alt = [torch.from_numpy(np.ascontiguousarray(grid["forcing"][name] * (0.5 if name in ("uo", "vo", "ua", "va") else 1.0))).to(dev)
                   for name in T.FORCING_NAMES]
    ptrs_alt = [t.data_ptr() for t in forcing_alt]
    if os.environ.get("KID_TEST_CONST_FORCING"):
        ptrs_alt = ptrs

    def run(kind):
        ib = Icebergs(grid, p, capacity=len(b["lon"]), device=0)
        try:
            ib.set_stream(torch.cuda.current_stream().cuda_stream)
            ib.upload_bergs(b)
            if kind == "plain":
                _, count = ib.accum_device_ptr()
                acc_t = torch.zeros(count, dtype=torch.float64, device=dev)
                ib.bind_accum_buffer(acc_t.data_ptr(), count)
                st = ShardedStepper(ib, acc_t, ib.ncell, p.diag_mask, None, params=p)
                for k in range(nsteps):
                    st.set_forcing_device(ptrs_alt if (k & 1) else ptrs)
                    st.step()
            else:
                st = PipelinedStepper(ib, p, None, split_general=(split_general is True), slow_lane=slow_lane)
                for k in range(nsteps):
                    st.set_forcing_device(ptrs_alt if (k & 1) else ptrs)
                    st.step()
            st.flush()
            torch.cuda.synchronize()
            acc, out, scal = ib.fetch()
            return ib.download_bergs(), acc.copy(), out.copy(), scal.copy()
        finally:
            ib.close()
    ref, got = run("plain"), run("pipelined")
    P.compare(ref, got, "pipelined vs plain", params=p)
    # what a berg does never depends on the schedule: its own fields are bit-identical (only the atomically summed planes differ)
    ra, ga = ref[0]["alive"] != 0, got[0]["alive"] != 0
    o1, o2 = np.argsort(ref[0]["id"][ra]), np.argsort(got[0]["id"][ga])
    for f in ("lon", "lat", "uvel", "vvel", "mass", "thickness", "width", "length", "xi", "yj", "mass_of_bits"):
        assert np.array_equal(ref[0][f][ra][o1], got[0][f][ga][o2]), f


@pytest.mark.parametrize("verlet", [False, True])
def test_time_average_weight(oracle, verlet):
    """time_average_weight=T: the spreading moves into the integrator stages (IB:7264, 7395-7620), calculate_mass_on_ocean
    then zeroes those planes without refilling them (IB:4984-4997): spread_mass is identically zero in the reference,
    and everything else goes on as before"""
    from icebergs_amd import types as T
    grid, p, b = S.config_c2(n=4000, seed=31, continents=True)
    S.set_diag_all(p)
    p.time_average_weight = 1
    if verlet:
        p.Runge_not_Verlet = 0
    ref, got = _both(grid, p, b, 12, "fused")
    P.compare(ref, got, "time_average_weight/verlet=%s" % verlet, params=p)
    k = T.ENUMS["KID_O_SPREAD_MASS"]
    assert not ref[2][k].any() and not got[2][k].any()
    assert np.abs(got[1][T.ENUMS["KID_A_FLOATING_MELT"]]).max() > 0   # the melt fluxes are still there


@pytest.mark.parametrize("variant", ["rk4", "verlet_new_order", "cutoff", "without_decay", "without_decay_new_order"])
def test_find_melt_using_spread_mass(oracle, variant):
    """find_melt_using_spread_mass=T (IB:5490-5503, 3436-3445): the melt flux handed to the ocean is the gridded mass the
    step lost, max((spread_mass_old - spread_mass)/dt, 0), not the sum of the bergs' own melt terms"""
    from icebergs_amd import types as T
    grid, p, b = S.config_c2(n=4000, seed=33, continents=True)
    S.set_diag_all(p)
    p.find_melt_using_spread_mass = 1
    if variant == "verlet_new_order":
        p.Runge_not_Verlet, p.old_interp_flds_order = 0, 0
    if variant == "cutoff":
        p.apply_thickness_cutoff_to_gridded_melt, p.melt_cutoff = 1, 3800.0   # cells whose mean draught exceeds 200 m are cut
    if variant.startswith("without_decay"):   # the bergs keep their size; thermodynamics spreads what they WOULD weigh (IB:3219-3238)
        p.Iceberg_melt_without_decay = 1
        if variant.endswith("new_order"):
            p.Runge_not_Verlet, p.old_interp_flds_order = 0, 0
    ref, got = _both(grid, p, b, 6, "fused")
    P.compare(ref, got, "find_melt/" + variant, params=p)
    if variant.startswith("without_decay"):
        o = np.argsort(got[0]["id"])
        assert np.array_equal(got[0]["mass"][o], b["mass"][np.argsort(b["id"])])      # nothing decayed
    q = S.params_copy(p)
    q.find_melt_using_spread_mass = 0
    plain = P.run_oracle(grid, q, b, 6)
    k = T.ENUMS["KID_A_FLOATING_MELT"]
    assert np.abs(ref[1][k]).max() > 0 and not np.allclose(ref[1][k], plain[1][k], rtol=1e-3)   # it IS a different flux


@pytest.mark.parametrize("without_decay", [False, True])
def test_find_melt_using_spread_mass_sharded_stepper(oracle, without_decay):
    """the sharded path with find_melt_using_spread_mass: the two planes of grd%spread_mass_old / spread_mass_tmp live in the
    caller's tensor (kid_bind_spread_mass_old) where the ranks sum them between the local work and the gather; two shards on
    one GPU, summed by hand the way the all-reduce would, give the single-handle result"""
    import torch
    from icebergs_amd import types as T, distributed as D
    from icebergs_amd.framework import Icebergs
    grid, p, b = S.config_c2(n=6000, seed=35, continents=True)
    p.find_melt_using_spread_mass = 1
    p.Iceberg_melt_without_decay = 1 if without_decay else 0
    ref = P.run_hip(grid, p, b, 3)
    shards = [D.take_shard(b, r, 2) for r in range(2)]
    hs, accs, olds = [], [], []
    try:
        for sh in shards:
            ib = Icebergs(grid, p, capacity=len(sh["lon"]))
            ib.upload_bergs(sh)
            ib.set_resort_interval(0)
            _, count = ib.accum_device_ptr()
            acc = torch.zeros(count, dtype=torch.float64, device="cuda")
            old = torch.zeros(2 * ib.ncell, dtype=torch.float64, device="cuda")
            ib.bind_accum_buffer(acc.data_ptr(), count)
            ib.bind_spread_mass_old(old.data_ptr(), old.numel())
            hs.append(ib); accs.append(acc); olds.append(old)
        for step in range(3):
            for ib in hs:
                ib.step_local()
                ib.sync()
            tot_acc, tot_old = accs[0] + accs[1], olds[0] + olds[1]          # what the all-reduce leaves on every rank
            for ib, acc, old in zip(hs, accs, olds):
                acc.copy_(tot_acc); old.copy_(tot_old)
                torch.cuda.synchronize()
                ib.step_gather()
                ib.sync()
        acc0, out0, _ = hs[0].fetch()
        k = T.ENUMS["KID_A_FLOATING_MELT"]
        assert np.abs(ref[1][k]).max() > 0
        assert P.rel_err(acc0[k], ref[1][k]) <= P.TOL_GRID
        assert P.rel_err(out0[T.OUT_NAMES["spread_mass"]], ref[2][T.OUT_NAMES["spread_mass"]]) <= P.TOL_GRID
    finally:
        for ib in hs:
            ib.close()


@pytest.mark.parametrize("verlet", [False, True])
def test_periodic_reentry(oracle, verlet):
    """periodic_reentry=1 on the zonally cyclic config-2 grid: bergs driven across the seam in both directions come back on
    the other side (cell index one period away, lon unchanged, xi / yj recomputed) and the 9-point gather reads across the
    seam; without the switch the same bergs are removed"""
    from icebergs_amd import types as T
    grid = S.c2_forcing(S.latlon_grid())
    grid["forcing"]["uo"][:] = np.where(grid["static"]["lat"] > 0, 1.5, -1.5)    # eastward north of the equator, westward south of it
    grid["forcing"]["vo"][:] = 0.0
    p = S.default_params()
    p.dt = 1800.0
    S.set_diag_all(p)
    if verlet:
        p.Runge_not_Verlet = 0
    b1 = S.place_bergs(grid, 1500, 61, (356, 360), (120, 180))     # next to the east edge, northern hemisphere
    b2 = S.place_bergs(grid, 1500, 62, (1, 5), (20, 80))           # next to the west edge, southern hemisphere
    b = {k: np.concatenate([b1[k], b2[k]]) for k in b1}
    b["id"] = np.arange(1, 3001, dtype=np.int64)
    q = S.params_copy(p)
    q.periodic_reentry = 1
    nsteps = 60
    ref, got = _both(grid, q, b, nsteps, "fused")
    P.compare(ref, got, "periodic re-entry/verlet=%s" % verlet, params=q)
    gb = got[0]
    alive = gb["alive"] != 0
    assert alive.sum() == 3000                                     # nobody is lost
    o = np.argsort(gb["id"])
    east_start = (b["ine"] > 300)
    crossed_e = east_start & (gb["ine"][o] < 100) & (gb["lon"][o] > 360.0)   # back in the low columns, longitude still counting up
    crossed_w = (~east_start) & (gb["ine"][o] > 300)
    assert crossed_e.sum() > 200 and crossed_w.sum() > 200
    sm = got[2][T.ENUMS["KID_O_SPREAD_MASS"]]
    d = grid["desc"]
    assert sm[:, d.isc - d.isd].max() > 0 and sm[:, d.iec - d.isd].max() > 0   # mass on both sides of the seam
    lost = P.run_hip(grid, p, b, nsteps)                                        # the default: removed at the edge
    assert (lost[0]["alive"] != 0).sum() < 3000 - 400


def test_slow_lane_with_collective_matches_plain(oracle):
    """the N>1 code path of the slow-lane schedule (RCCL all-reduce + gather on a third stream), rehearsed with one
    rank: must give what the serial single-stream sequence gives"""
    import torch
    import torch.distributed as dist
    from icebergs_amd.framework import Icebergs
    from icebergs_amd.distributed import ShardedStepper, PipelinedStepper
    from icebergs_amd import types as T
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        grid, p, b = S.config_c2(n=30000, seed=23, continents=True)
        S.set_diag_all(p)
        nsteps = 35
        dev = torch.device("cuda", 0)
        forcing_dev = [torch.from_numpy(np.ascontiguousarray(grid["forcing"][name])).to(dev) for name in T.FORCING_NAMES]
        ptrs = [t.data_ptr() for t in forcing_dev]

        def run(kind):
            ib = Icebergs(grid, p, capacity=len(b["lon"]), device=0)
            try:
                ib.set_stream(torch.cuda.current_stream().cuda_stream)
                ib.upload_bergs(b)
                if kind == "plain":
                    _, count = ib.accum_device_ptr()
                    acc_t = torch.zeros(count, dtype=torch.float64, device=dev)
                    ib.bind_accum_buffer(acc_t.data_ptr(), count)
                    st = ShardedStepper(ib, acc_t, ib.ncell, p.diag_mask, None, params=p)
                else:
                    st = PipelinedStepper(ib, p, dist, force_collective=True, slow_lane=True)
                    assert st.three
                for _ in range(nsteps):
                    st.set_forcing_device(ptrs)
                    st.step()
                st.flush()
                torch.cuda.synchronize()
                acc, out, scal = ib.fetch()
                return ib.download_bergs(), acc.copy(), out.copy(), scal.copy()
            finally:
                ib.close()
        ref, got = run("plain"), run("slow_lane+collective")
        P.compare(ref, got, "slow lane + collective vs plain", params=p)
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("verlet", [False, True])
def test_polar_cap(oracle, verlet):
    """Bergs between 85N and the pole on a lat-lon grid whose top row of cells touches 90N: the tangent-plane branch
    of both integrators above 89N (IB:7347, 7296), the polar variant of pos_within_cell (FW:6370-6391) with the 5-point
    point-in-cell test (FW:6226-6296), and bergs that cross the pole row and leave."""
    grid = S.c2_forcing(S.latlon_grid(ni=360, nj=50, lat0=50.0, dlat=0.8))
    p = S.default_params()
    p.dt = 1800.0
    if verlet:
        p.Runge_not_Verlet = 0
    S.set_diag_all(p)
    b = S.place_bergs(grid, 3000, 7, (3, 358), (44, 50))
    ref, got = _both(grid, p, b, 24, "fused")
    P.compare(ref, got, "polar/verlet=%s" % verlet, params=p)
    assert (ref[0]["lat"][ref[0]["alive"] != 0] > 89.0).sum() > 100


def test_periodic_seams(oracle):
    """Bergs hugging the zonal seam of the periodic lat-lon grid (Lx=360: the modulo branches of the point-in-cell
    tests and of calc_xiyj, FW:6163-6296, 6439-6534) and of a periodic Cartesian grid (Lx=20 km, the regular-grid
    branch FW:6320-6330); some cross the seam and leave the rank (FW:3024-3041)."""
    grid, p, _ = S.config_c2(n=10, seed=2)
    S.set_diag_all(p)
    east = S.place_bergs(grid, 1500, 31, (357, 360), (8, 192))
    west = S.place_bergs(grid, 1500, 32, (1, 3), (8, 192))
    b = {k: (np.concatenate([east[k], west[k]]) if hasattr(east[k], "dtype") else east[k]) for k in east}
    b["id"] = np.arange(1, len(b["lon"]) + 1, dtype=np.int64)
    b["uvel"][:1500], b["uvel"][1500:] = 0.4, -0.4          # towards the seam
    b = S.sort_reference_order(b)
    ref, got = _both(grid, p, b, 30, "fused")
    P.compare(ref, got, "seam/latlon", params=p)
    assert 0 < (ref[0]["alive"] == 0).sum() < len(b["lon"])   # some left, some are still there
    # Cartesian, periodic in x
    grid = S.c1_forcing(S.cartesian_grid(20, 20, 1000.0, Lx=20000.0))
    p = S.default_params()
    p.dt, p.lat_ref, p.use_f_plane = 600.0, -70.0, 1
    S.set_diag_all(p)
    b = S.place_bergs(grid, 400, 33, (1, 20), (4, 17), klass=np.arange(400) % 10)
    ref, got = _both(grid, p, b, 60, "fused")
    P.compare(ref, got, "seam/cartesian", params=p)


def test_accum_live_count_is_what_the_python_host_reduces():
    """kid_accum_live_count (what a Fortran / MPI host sums across ranks) and accumulator_views (what
    the RCCL path of bench.py sums) name the same prefix of the accumulator block, whatever the namelist asks for"""
    from icebergs_amd.framework import Icebergs
    from icebergs_amd.distributed import accumulator_views
    from icebergs_amd import types as T
    grid, p0, b = S.config_c2(n=2000, seed=3)
    E = T.ENUMS
    cases = [dict(), dict(pass_fields_to_ocean_model=1), dict(diag_mask=E["KID_DIAG_SPREAD_AREA"]), dict(diag_mask=E["KID_DIAG_MELT_BUOY"]),
             dict(diag_mask=E["KID_DIAG_MASS"] | E["KID_DIAG_USTAR_ICEBERG"])]
    for kw in cases:
        p = S.params_copy(p0)
        for k, v in kw.items():
            setattr(p, k, v)
        ib = Icebergs(grid, p, capacity=len(b["lon"]), device=0)
        try:
            _, count = ib.accum_device_ptr()
            assert count == T.NSCALAR + T.NACC * ib.ncell
            live = ib.accum_live_count()
            want = len(accumulator_views(np.zeros(count), ib.ncell, p.diag_mask, p)[0])
            assert live == want, (kw, live, want)
        finally:
            ib.close()

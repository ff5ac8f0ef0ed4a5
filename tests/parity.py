"""Shared helpers of the parity tests: run the same seeded inputs through the CPU oracle and through the HIP
library (via the C ABI) and compare with the tolerances of SURVEY.md section 8d."""
import numpy as np

from icebergs_amd import synthetic as S

TRAJ_FIELDS = ["lon", "lat", "uvel", "vvel", "axn", "ayn", "bxn", "byn", "xi", "yj"]
SIZE_FIELDS = ["mass", "thickness", "width", "length", "mass_of_bits", "mass_scaling", "mass_of_fl_bits",
               "mass_of_fl_bergy_bits", "fl_k"]
ENV_FIELDS = ["uo", "vo", "ui", "vi", "ua", "va", "ssh_x", "ssh_y", "sst", "sss", "cn", "hi"]
TOL_TRAJ = 1.0e-10   # BASELINE.json: trajectories within 1e-10 relative of the CPU reference
TOL_SIZE = 1.0e-10   # berg mass/size
TOL_GRID = 1.0e-9    # per-cell fields, relative to the field max (summation order)


def rel_err(a, b):
    scale = np.max(np.abs(b)) if b.size else 0.0
    if scale == 0.0:
        return float(np.max(np.abs(a))) if a.size else 0.0
    return float(np.max(np.abs(a - b)) / scale)


def acc_scales(acc):
    """Scale of every accumulator plane: its own max, except that the 9 slots of mass/area/Uvel/Vvel_on_ocean share
    the max of their field (a hexagon or rectangle corner that overlaps a neighbour cell by a rounding-sized sliver
    is the only entry of its slot plane)."""
    from icebergs_amd import types as T
    sc = np.array([np.max(np.abs(acc[k])) for k in range(acc.shape[0])])
    for base in ("mass_on_ocean", "area_on_ocean", "uvel_on_ocean", "vvel_on_ocean"):
        b0 = T.ACC_NAMES[base]
        sc[b0:b0 + 9] = sc[b0:b0 + 9].max()
    return sc


def acc_err(got, ref, k, scales):
    d = float(np.max(np.abs(got[k] - ref[k])))
    return d / scales[k] if scales[k] > 0 else d


def run_oracle(grid, params, bergs, nsteps):
    import oracle_lib
    o = oracle_lib.Oracle(grid, params)
    b = S.copy_bergs(bergs)
    o.run_step(b, nsteps)
    return b, o.acc.copy(), o.out.copy(), o.scalars.copy()


def run_hip(grid, params, bergs, nsteps, mode="fused", device=0):
    from icebergs_amd.framework import Icebergs
    ib = Icebergs(grid, params, capacity=max(len(bergs["lon"]), 1), device=device)  # spare rows = room for children
    try:
        ib.upload_bergs(bergs)
        if mode == "fused":
            ib.run(nsteps)
        else:
            ib.run_phases(nsteps)
        acc, out, scal = ib.fetch()
        b = ib.download_bergs()
        return b, acc.copy(), out.copy(), scal.copy()
    finally:
        ib.close()


def run_oracle_mts(grid, params, bergs, bonds, nsteps):
    import oracle_lib
    o = oracle_lib.Oracle(grid, params)
    b, bd = S.copy_bergs(bergs), S.copy_bonds(bonds)
    if params.mts:
        o.run_step_mts(b, bd, nsteps)
    else:
        o.run_step_interactive(b, bd, nsteps)   # interacting bergs under the single-time-step scheme
    return (b, o.acc.copy(), o.out.copy(), o.scalars.copy()), bd


def run_hip_mts(grid, params, bergs, bonds, nsteps, device=0):
    from icebergs_amd.framework import Icebergs
    ib = Icebergs(grid, params, capacity=max(len(bergs["lon"]), 1), device=device)
    try:
        ib.upload_bergs(bergs)
        ib.upload_bonds(bonds)
        ib.run(nsteps)
        acc, out, scal = ib.fetch()
        b = ib.download_bergs()
        bd = ib.download_bonds(bonds["max_bonds"])
        return (b, acc.copy(), out.copy(), scal.copy()), bd
    finally:
        ib.close()


MTS_FIELDS = ["uvel_old", "vvel_old", "lon_old", "lat_old", "axn_fast", "ayn_fast", "bxn_fast", "byn_fast", "ang_vel", "ang_accel", "rot", "od"]
BOND_STATE = ["length", "tangd1", "tangd2", "nstress", "sstress", "rel_rotation"]


def bond_set(bergs, bonds, only_unbroken=False):
    """{(id, other_id): slot}: the bonds as a set, independent of list order"""
    n = len(bergs["lon"])
    out = {}
    for k in range(n):
        for s in range(bonds["count"][k]):
            if only_unbroken and bonds["broken"][s * n + k] != 0:
                continue
            out[(int(bergs["id"][k]), int(bonds["other_id"][s * n + k]))] = s * n + k
    return out


def compare_mts(ref, refbd, got, gotbd, label="", tol=1.0e-9, stiff_tol=1.0e-4):
    """MTS/DEM: rows are stable (no re-binning), so bergs and bonds are compared row by row.  The set of bonds and the
    set of broken bonds must be identical; states within `tol` (relative to the field max)."""
    rb, gb = ref[0], got[0]
    assert np.array_equal(rb["id"], gb["id"]) and np.array_equal(rb["alive"], gb["alive"]), label + ": rows differ"
    for f in ("ine", "jne", "conglom_id", "n_bonds"):
        assert np.array_equal(rb[f], gb[f]), "%s: %s differs" % (label, f)
    rep = {}
    # The DEM springs are stiff (dem_spring_coef*T ~ 1e9 N/m): a bond force is a difference of positions that agree to
    # the last bit or two, so forces, stresses and the accelerations built from them carry a relative error of
    # ~1e-16 * |x| * k / |F| -- they are compared at `stiff_tol`, positions/velocities/rotations at `tol`.
    stiff = {"axn", "ayn", "bxn", "byn", "axn_fast", "ayn_fast", "bxn_fast", "byn_fast", "ang_accel"}
    for f in TRAJ_FIELDS + MTS_FIELDS + SIZE_FIELDS:
        e = rel_err(gb[f], rb[f])
        rep[f] = e
        t = stiff_tol if f in stiff else (1.0e-6 if f in ("ang_vel", "rot") else tol)  # rotation integrates the stiff torques
        assert e <= t, "%s: %s rel err %.3e > %.1e" % (label, f, e, t)
    assert np.array_equal(refbd["count"], gotbd["count"]), label + ": bond counts differ"
    n = len(rb["lon"])
    live = (np.arange(refbd["max_bonds"])[:, None] < refbd["count"][None, :]).ravel()  # slots past a list's end are stale
    assert np.array_equal(refbd["other_id"][live], gotbd["other_id"][live]), label + ": bond partners differ"
    assert np.array_equal(refbd["broken"][live], gotbd["broken"][live]), label + ": set of broken bonds differs"
    for f in BOND_STATE + ["f_x", "f_y", "fd_x", "fd_y", "t", "t_d"]:
        e = rel_err(gotbd[f][live], refbd[f][live])
        rep["bond_" + f] = e
        t = tol if f == "length" else stiff_tol
        assert e <= t, "%s: bond %s rel err %.3e > %.1e" % (label, f, e, t)
    sc = acc_scales(ref[1])
    for k in range(ref[1].shape[0]):
        e = acc_err(got[1], ref[1], k, sc)
        assert e <= 10 * TOL_GRID, "%s: accumulator plane %d rel err %.3e" % (label, k, e)  # velocity-weighted planes carry the DEM velocities
    from icebergs_amd import types as T
    # ustar_iceberg steps from 0 to its value where spread_area becomes non-zero: leave out cells that only hold a
    # rounding-sized sliver of a footprint
    solid = np.abs(ref[2][T.OUT_NAMES["spread_area"]]) > 1.0e-9
    for k in range(ref[2].shape[0]):
        a, b_ = (got[2][k][solid], ref[2][k][solid]) if k == T.OUT_NAMES["ustar_iceberg"] else (got[2][k], ref[2][k])
        e = rel_err(a, b_)
        assert e <= 10 * TOL_GRID, "%s: output plane %d rel err %.3e" % (label, k, e)
    from icebergs_amd import types as T
    for name in ("nbergs_melted", "nspeeding_tickets", "nbergs_alive", "error_count", "nbonds_broken"):
        k = T.SCALAR_NAMES[name]
        assert ref[3][k] == got[3][k], "%s: scalar %s %r != %r" % (label, name, got[3][k], ref[3][k])
    return rep


def compare(ref, got, label="", params=None):
    """params: when given and nothing reads the area/Uvel/Vvel footprints (distributed.needs_footprint_planes), the
    library does not produce those 27 planes: they must come back all zero and are not compared."""
    rb, racc, rout, rscal = ref
    gb, gacc, gout, gscal = got
    report = {}
    # bergs are matched by id: the library may have re-binned (sorted) the SoA and dropped dead bergs
    nr, ng = int(rb.get("_n", len(rb["lon"]))), len(gb["lon"])
    ra, ga = rb["alive"] != 0, gb["alive"] != 0
    ra[nr:] = False
    ga[ng:] = False

    def order(b, m):
        # by id: ids are exact on both sides -- a footloose child's too (the per-cell counter values are handed out in the
        # reference's traversal order by the library, whatever the schedule of its lanes)
        return np.argsort(b["id"][m], kind="stable")
    orr, org = order(rb, ra), order(gb, ga)
    assert np.array_equal(np.sort(rb["id"][ra]), np.sort(gb["id"][ga])), label + ": set of surviving bergs differs"

    def R(f):
        return rb[f][ra][orr]

    def G(f):
        return gb[f][ga][org]
    def where(f):
        bad = np.nonzero(R(f) != G(f))[0][:4]
        return "%s: %s differs at %s: ref %s got %s (lon ref %s got %s)" % (label, f, bad, R(f)[bad], G(f)[bad], R("lon")[bad], G("lon")[bad])
    assert np.array_equal(R("ine"), G("ine")), where("ine")
    assert np.array_equal(R("jne"), G("jne")), where("jne")
    for f in TRAJ_FIELDS:
        e = rel_err(G(f), R(f))
        report[f] = e
        assert e <= TOL_TRAJ, "%s: %s rel err %.3e > %.1e" % (label, f, e, TOL_TRAJ)
    for f in SIZE_FIELDS:
        e = rel_err(G(f), R(f))
        report[f] = e
        assert e <= TOL_SIZE, "%s: %s rel err %.3e > %.1e" % (label, f, e, TOL_SIZE)
    for f in ENV_FIELDS:
        e = rel_err(G(f), R(f))
        report[f] = e
        assert e <= TOL_TRAJ, "%s: env %s rel err %.3e" % (label, f, e)
    sc = acc_scales(racc)
    dead = set()
    if params is not None:
        from icebergs_amd import types as T
        from icebergs_amd.distributed import needs_footprint_planes
        if not needs_footprint_planes(params):
            a0 = T.ACC_NAMES["area_on_ocean"]
            dead = set(range(a0, a0 + 27))
            assert not gacc[a0:a0 + 27].any(), label + ": footprint planes must stay zero when nothing reads them"
    for k in range(racc.shape[0]):
        if k in dead:
            continue
        e = acc_err(gacc, racc, k, sc)
        report["acc%d" % k] = e
        assert e <= TOL_GRID, "%s: accumulator plane %d rel err %.3e > %.1e" % (label, k, e, TOL_GRID)
    for k in range(rout.shape[0]):
        e = rel_err(gout[k], rout[k])
        report["out%d" % k] = e
        assert e <= TOL_GRID, "%s: output plane %d rel err %.3e > %.1e" % (label, k, e, TOL_GRID)
    # scalars: counters exact, heat within summation-order tolerance
    from icebergs_amd import types as T
    for name in ("nbergs_melted", "nbergs_calved_fl", "nspeeding_tickets", "nbergs_alive", "error_count"):
        k = T.SCALAR_NAMES[name]
        assert rscal[k] == gscal[k], "%s: scalar %s %r != %r" % (label, name, gscal[k], rscal[k])
    k = T.SCALAR_NAMES["net_heat_to_ocean"]
    assert abs(gscal[k] - rscal[k]) <= 1e-9 * max(abs(rscal[k]), 1e-300) + 0.0 or rscal[k] == gscal[k]
    return report

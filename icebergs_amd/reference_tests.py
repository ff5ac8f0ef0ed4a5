"""The reference's own regression tests, restated: generator (makeberg), namelist, driver grid and run length of
/root/reference/tests/{dem_ssbeam_test, dem_cbeam_test, collision_tests, dem_ground_frac_test}, and the integers those tests
record (the `bergs_chksum` line printed by icebergs_save_restart and kept as a comment at the head of each input.nml).

Which of the recorded integers another implementation can reach.  chksum3 / chksum4 are mpp_chksum of
grd%tmp(i,j) = sum over the bergs of the cell of time_hash*pos_hash + log(mass) (icebergs_framework.F90:6943-6950).  Every
generator writes start_year = start_day = 0 into its restart (e.g. dem_ssbeam_test/makeberg/makeberg.py:65-69), so
time_hash = 0, and every namelist has set_melt_rates_to_zero=.true., so the masses are those of the generator: the number is
a function of log(mass) and of the per-cell occupancy at the end of the run -- no bit pattern of a position enters.  '#' is
the number of bergs.  chksum / chksum2 / chksum5 sum bit patterns of positions and velocities out of another compiler's
binary and are out of reach of any other implementation.

A population is returned "as read_restart_bergs leaves it" (icebergs_fms2io.F90:663-1049): ids of a 32-bit-era file from
generate_id in file order (FW:4165-4179, IO2:917), *_old = current values, start_* as in the file (zeros), the cell found from
the position (ignore_ij_restart), xi/yj by pos_within_cell; rows in the order the traversal visits them right after the read
(cells j outer / i inner; inside a cell bergs with identical `inorder` keys sit in reverse file order, because
insert_berg_into_list puts a tie at the head, FW:4279-4285); bonds from initialize_iceberg_bonds (IB:356-441) with
form_a_bond's insertion at the head (FW:4866-4877), i.e. a berg's partners in reverse traversal order.
"""
import numpy as np

from . import synthetic as S

RECORDED = {
    # test: (chksum3 = chksum4, '#')            where it is recorded
    "dem_ssbeam": (-1459704404, 29),          # tests/dem_ssbeam_test/input.nml:1
    "dem_cbeam": (-1504290914, 90),           # tests/dem_cbeam_test/input.nml:2,5 (both moment-of-inertia variants)
    "collision_KID": (1964715299, 16),        # tests/collision_tests/input_KID.nml:1, README:16
    "collision_MTS_KID": (1124700946, 16),    # tests/collision_tests/input_MTS_KID.nml:1, README:19
    "collision_iKID": (-1070230468, 16),      # tests/collision_tests/input_iKID.nml:1, README:22
    "dem_ground_frac": (946244516, 69),       # tests/dem_ground_frac_test/input.nml:7,10 (both contact variants)
}


def _i32(u):
    u = int(u) & 0xFFFFFFFF
    return u - (1 << 32) if u >= (1 << 31) else u


def chksum3_of_occupancy(cells, masses):
    """chksum3 as a function of what it depends on when time_hash = 0: for every cell the left-to-right sum of log(mass)
    over its bergs (list order), then mpp_chksum = the sum of the IEEE bit patterns modulo 2^64, narrowed to 32 bits.
    `cells`: one hashable cell key per berg, in traversal order; `masses` alongside."""
    tmp = {}
    for c, m in zip(cells, masses):
        tmp[c] = (tmp.get(c, 0.0) + 0.0) + float(np.log(np.float64(m)))
    return _i32(sum(int(np.float64(v).view(np.uint64)) for v in tmp.values()))


def driver_grid(ni, nj, gridres, Lx=-1.0, halo=3):
    """driver/icebergs_driver.F90:274-286 (all tests here are Cartesian): lon = gridres*i, lat = gridres*j at the NE corner,
    depth 1000, all wet.  The halo of this package's grids is S.HALO cells whatever the namelist's `halo`: nothing on the path
    reads further than two cells from a berg that sits on the computational domain."""
    return S.cartesian_grid(ni, nj, gridres, Lx=Lx)


def _cell_of(grid, gridres, x, y):
    """find_cell on the regular Cartesian grid (FW:5868-5900 picks the cell whose centre is nearest, then is_point_in_cell):
    the cell (i, j) with lon(i-1) < x <= lon(i) in the sense of is_point_in_cell's half-open test (FW:6117-6127: x >= xlo and
    x < xhi)."""
    i = int(np.floor(x / gridres)) + 1
    j = int(np.floor(y / gridres)) + 1
    return i, j


def as_read_from_restart(grid, gridres, iNg, x, y, thickness, width, mass, uvel=0.0, vvel=0.0, static=None, mass_scaling=1.0):
    """The population read_restart_bergs builds from a 32-bit-era file of a generator (file order = argument order)."""
    n = len(x)
    b = S.empty_bergs(n)
    counter = {}
    for k in range(n):
        i, j = _cell_of(grid, gridres, x[k], y[k])
        b["ine"][k], b["jne"][k] = i, j
        counter[(i, j)] = counter.get((i, j), 0) + 1                       # generate_id FW:4165-4179: counter first, then the hash
        b["id"][k] = (counter[(i, j)] << 32) + (i + iNg * (j - 1))
        x1, y1 = gridres * i - gridres / 2.0, gridres * j - gridres / 2.0     # pos_within_cell, regular grid FW:6344-6354
        b["xi"][k] = ((x[k] - x1) / gridres) + 0.5
        b["yj"][k] = ((y[k] - y1) / gridres) + 0.5
    b["lon"][:], b["lat"][:] = x, y
    b["thickness"][:], b["width"][:], b["length"][:], b["mass"][:] = thickness, width, width, mass
    b["mass_scaling"][:] = mass_scaling
    b["uvel"][:], b["vvel"][:] = uvel, vvel
    for f, g in (("uvel_old", "uvel"), ("vvel_old", "vvel"), ("lon_old", "lon"), ("lat_old", "lat"), ("uvel_prev", "uvel"), ("vvel_prev", "vvel")):
        b[f][:] = b[g]                                                      # IO2:905-908 (uvel_prev: verlet_stepping sets it before use)
    if static is not None:
        b["static_berg"][:] = static
    # start_year = start_day = start_mass = start_lon = start_lat = 0 (the generators' var[:]=0)
    file_row = np.arange(n)
    order = np.lexsort((-file_row, b["ine"], b["jne"]))                       # ties of `inorder` go to the head of the list
    for k in list(b.keys()):
        b[k] = np.ascontiguousarray(b[k][order])
    return b


def dem_tests_init(b, p):
    """FW:4687-4710"""
    b["start_lon"][:], b["start_lat"][:] = b["lon"], b["lat"]
    p.dem_tests_start_lon, p.dem_tests_end_lon = float(b["lon"].min()), float(b["lon"].max())


def initialize_iceberg_bonds(b, p, length=None):
    """IB:356-441 over the rows in traversal order: `length` = length_for_manually_initialize_bonds, None =
    manually_initialize_bonds_from_radii (r_dist < 1.25 (radius1 + radius2), radii from length*width)."""
    n = len(b["lon"])
    mb = int(p.max_bonds)
    bd = S.empty_bonds(n, mb)
    lists = [[] for _ in range(n)]
    rdenom = 1.0 / (2.0 * np.sqrt(3.0)) if p.hexagonal_icebergs else 1.0 / 4.0
    for k in range(n):
        for o in range(n):
            if b["id"][k] == b["id"][o] or b["id"][o] in lists[k]:
                continue
            rx, ry = b["lon"][k] - b["lon"][o], b["lat"][k] - b["lat"][o]
            r = np.sqrt((rx ** 2) + (ry ** 2))
            if length is None:
                r1 = np.sqrt(b["length"][k] * b["width"][k] * rdenom)
                r2 = np.sqrt(b["length"][o] * b["width"][o] * rdenom)
                bond = r < 1.25 * (r1 + r2)
            else:
                bond = r < length
            if bond:
                lists[k].insert(0, b["id"][o])                                # form_a_bond: new bond at the head
    for k in range(n):
        assert len(lists[k]) <= mb, (k, len(lists[k]))
        for s, o in enumerate(lists[k]):
            bd["other_id"][s * n + k] = o
        bd["count"][k] = len(lists[k])
        b["n_bonds"][k] = len(lists[k])                                     # assign_n_bonds IB:169
    return bd


def _common_namelist(p):
    """what the four namelists share (icebergs_nml of each input.nml)"""
    p.lat_ref, p.use_f_plane = 0.0, 0
    p.Runge_not_Verlet, p.use_new_predictive_corrective = 0, 1
    p.old_bug_bilin, p.use_old_spreading = 0, 0
    p.iceberg_bonds_on, p.interactive_icebergs_on = 1, 1
    p.set_melt_rates_to_zero = 1
    p.ustar_icebergs_bg, p.const_gamma, p.Use_three_equation_model = 0.0, 0, 0
    p.apply_thickness_cutoff_to_gridded_melt, p.apply_thickness_cutoff_to_bergs_melt, p.melt_cutoff = 1, 1, 10.0
    p.add_weight_to_ocean = 1
    return p


# ------------------------------------------------------------------------------------------------------------------------------
# tests/dem_ssbeam_test and tests/dem_cbeam_test
# ------------------------------------------------------------------------------------------------------------------------------
def dem_beam(kind):
    """kind 'ss': dem_ssbeam_test (29 elements, nmax = 10 steps of 1 s, 1e5 sub-steps); 'c': dem_cbeam_test (90 elements, nmax =
    300 steps of 100 s, 2000 sub-steps).  Generator: makeberg/makeberg.py main() of each test."""
    grid, p, _, _ = S.config_beam("supported" if kind == "ss" else "cantilever")      # namelist + the driver's 20x20 grid of 15 km cells
    gridres = 15000.0
    xs, ys, h = 101.0e3, 151.0e3, 1.0
    bx, by, st = [], [], []
    if kind == "ss":
        r, rho, nbergs, nsteps = 0.25, 800.0, 29, 10
        p.orig_dem_moment_of_inertia = 0                                      # not in dem_ssbeam_test/input.nml: the default, FW:60
        x, y, count = xs, ys, 0
        while count < nbergs:                                                 # makeberg.py:289-308
            count += 1
            if count == 1:
                y = y + 2 * r
            else:
                x = x + 2 * r
            bx.append(x), by.append(y), st.append(0.0)
    else:
        r, rho, nbergs, nsteps = 2500.0, 900.0, 90, 300
        x, y, count = xs, ys - 2 * r, 0
        while count < nbergs:                                                 # makeberg.py:291-311
            count += 1
            if count % 30 == 1:
                x, y, static = xs, y + 2 * r, 1.0
            else:
                static, x = 0.0, x + 2 * r
            bx.append(x), by.append(y), st.append(static)
    element_area = (2.0 * r) ** 2
    n = len(bx)
    b = as_read_from_restart(grid, gridres, 20, bx, by, [h] * n, [np.sqrt(element_area)] * n, [h * rho * element_area] * n, static=st)
    bd = initialize_iceberg_bonds(b, p)                                         # manually_initialize_bonds_from_radii
    dem_tests_init(b, p)
    w = float(np.sqrt(element_area))
    p.constant_length = p.constant_width = w
    return {"grid": grid, "params": p, "bergs": b, "bonds": bd, "nsteps": nsteps, "gridres": gridres, "ni": 20}


# ------------------------------------------------------------------------------------------------------------------------------
# tests/collision_tests
# ------------------------------------------------------------------------------------------------------------------------------
def collision_elements():
    """makeberg/makeberg.py (a disc of 300 m thick ice on a 20x20 grid of 1 km cells: the cells within 1 km of (4.5, 4.5) km)
    + initialize_bergs_in_pattern.py as makeberg/RUN calls it (Generic geometry -> Use_default_radius, hexagon elements,
    collision_test: the conglomerate mirrored about y = 10 km).  Returns x, y, thickness, width, mass in file order."""
    nx = ny = 20
    grid_res = 1.0e3
    thick = np.zeros((ny, nx))
    cx = cy = 4.5e3
    for i in range(nx):
        for j in range(ny):
            tx, ty = float(i) * grid_res, float(j) * grid_res
            if np.sqrt((tx - cx) * (tx - cx) + (ty - cy) * (ty - cy)) < 1.0e3:
                thick[i, j] = 300.0                                           # (the script's [i, j] on a (ny, nx) variable: the disc is symmetric)
    x = np.array([float(i) * grid_res for i in range(nx)])
    y = np.array([float(j) * grid_res for j in range(ny)])
    ice_mask = thick > 0.0
    dx = x[1] - x[0]
    Radius = (np.sqrt(3) / 2.0) * (0.45 * dx)                                 # Use_default_radius, hexagon (:1790-1796)
    rho_ice = 918.0                                                           # the script's default (:196), not the namelist's rho_bergs
    element_area = (3.0 * np.sqrt(3.0) / 2.0) * ((4.0 / 3.0) * (Radius) ** 2)
    X_min, X_max, Y_min, Y_max = np.min(x), np.max(x), np.min(y), np.max(y)
    N = 2 * int(np.ceil((X_max - X_min) / Radius))
    M = 2 * int(np.ceil((Y_max - Y_min) / Radius))
    dxb, dyb, width = [], [], []
    for i in range(N):                                                        # Create_icebergs :827-868
        y_start = Radius + ((i % 2) * Radius)
        x_start = (2 / np.sqrt(3)) * Radius
        x_val = x_start + (np.sqrt(3) * Radius * i)
        for j in range(M):
            y_val = y_start + (2 * j * Radius)
            if (x_val >= (X_max - X_min + dx)) or (x_val <= 0) or (y_val >= (Y_max - Y_min + dx)) or (y_val <= 0):
                continue
            if ice_mask[int(np.floor(y_val / dx)), int(np.floor(x_val / dx))]:
                dxb.append(x_val), dyb.append(y_val), width.append(np.sqrt(element_area))
    thickness = [thick[int(np.floor(yv / dx)), int(np.floor(xv / dx))] for xv, yv in zip(dxb, dyb)]
    mass = [t * rho_ice * (w) ** 2 for t, w in zip(thickness, width)]         # Define_iceberg_thickness_and_mass :641
    for i in range(len(dxb)):                                                 # collision_test :905-915
        dxb.append(dxb[i]), dyb.append(20000.0 - dyb[i]), width.append(width[i]), thickness.append(thickness[i]), mass.append(mass[i])
    return dxb, dyb, thickness, width, mass


def collision(kind, periods=3):
    """tests/collision_tests with input_{KID, MTS_KID, iKID}.nml: 16 elements in two bonded conglomerates of 8 driven against each
    other (vo = +-0.2 m/s for 0 < x <= 10 km, DRV:313-326) and east (uo = 0.2 m/s) for ibhrs = 48 hours.

    The reference runs it on 4 PEs with a zonally periodic 20 km domain (iflags = CYCLIC_GLOBAL_DOMAIN, Lx = 20000): the
    conglomerates drift ~30 km and cross the seam, where the PE-local coordinates stay continuous (halo copies are shifted
    by Lx, FW:2251-2265, 2368-2374; update_latlon FW:5128-5169).  Here the channel is unrolled: `periods` copies of the 20 km
    period side by side on one non-periodic grid, the forcing repeated with the period -- the same continuous coordinates
    everywhere, without the seam.  Occupancy is then counted per cell of the period (`wrap_to_period`)."""
    gridres, ni0, nj = 1000.0, 20, 20
    grid = driver_grid(ni0 * periods, nj, gridres, Lx=-1.0)
    d = grid["desc"]
    ii, jj = S._ij(d)
    lon = gridres * (((ii - 1) % ni0) + 1) * np.ones_like(jj)                  # the corner's longitude within its own period: 1000 .. 20000
    lat = gridres * jj * np.ones_like(ii)
    F = grid["forcing"]
    ibuo = ibvo = 0.2
    F["uo"][:] = ibuo
    mid = 10.0e3
    # DRV:313-326 on the driver's own lon(i,j) = gridres*i, then (inside icebergs_run) the cyclic halo update: corner i = 20 holds
    # lon = 20000 > mid -> 0, which is also the periodic image of corner 0 (lon <= 0 -> 0)
    vo = np.where((lon > mid) | (lat == mid), 0.0, np.where(lat > mid, -ibvo, ibvo))
    F["vo"][:] = vo
    F["sst"][:] = -2.0                                                        # the driver's default sst (DRV:80)
    p = _common_namelist(S.default_params())
    p.rho_bergs, p.hexagonal_icebergs, p.max_bonds = 850.0, 1, 6
    p.spring_coef, p.radial_damping_coef, p.tangental_damping_coef = 1.0e-5, 1.0e-4, 2.0e-5
    p.critical_interaction_damping_on, p.scale_damping_by_pmag, p.tang_crit_int_damp_on = 1, 1, 1
    p.coastal_drift = 0.4
    p.allow_bergs_to_roll, p.use_updated_rolling_scheme = 1, 1
    p.rotate_icebergs_for_mass_spreading = 1
    p.internal_bergs_for_drag = 0
    if kind == "KID":
        p.dt, nsteps = 60.0, 48 * 60
        p.mts, p.dem, p.explicit_inner_mts, p.force_convergence, p.mts_sub_steps = 0, 0, 0, 0, 1
        p.old_interp_flds_order = 1                                           # FW:1483: neither mts, dem nor footloose
        p.contact_distance, p.contact_spring_coef = 0.0, p.spring_coef        # FW:1311
        p.contact_cells_lon = p.contact_cells_lat = 1
    else:
        p.dt, nsteps = 3600.0, 48
        p.mts, p.explicit_inner_mts, p.mts_sub_steps = 1, 1, 60
        p.old_interp_flds_order = 0
        p.force_convergence, p.convergence_tolerance = 1, 1.0e-8
        p.contact_distance, p.contact_spring_coef = 1.75e3, 1.0e-7
        p.contact_cells_lon, p.contact_cells_lat = 2, 2                       # FW:1492-1515 with 1 km cells
        if kind == "iKID":
            p.dem, p.poisson, p.dem_damping_coef, p.dem_spring_coef = 1, 0.3, 1.0, 4471.94
        else:
            assert kind == "MTS_KID", kind
    x, y, th, w, m = collision_elements()
    b = as_read_from_restart(grid, gridres, ni0, x, y, th, w, m)
    bd = initialize_iceberg_bonds(b, p, length=800.0)
    return {"grid": grid, "params": p, "bergs": b, "bonds": bd, "nsteps": nsteps, "gridres": gridres, "ni": ni0}


# ------------------------------------------------------------------------------------------------------------------------------
# tests/dem_ground_frac_test
# ------------------------------------------------------------------------------------------------------------------------------
def dem_ground_frac():
    """tests/dem_ground_frac_test: 69 hexagonally packed elements drifting onto a Gaussian seamount (big_grounding_test,
    DRV:288-307: the grid shifted by 0.45 m, land rows south of -5 km and north of 220 km, bump_depth = 50), 72 hours of
    dt = 1800 s with 200 sub-steps."""
    grid, p, _, _ = S.config_c4(reference_pattern=True, sub_steps=200)
    gridres = 5000.0
    st = grid["static"]
    d = grid["desc"]
    for f in ("lon", "lat", "lonc", "latc"):
        st[f][:] = st[f] - 0.45                                                # DRV:289-290
    st["msk"][(st["lat"] <= -5.0e3) | (st["lat"] >= 220.0e3)] = 0.0              # DRV:293-295
    a, cw, bx, by = 1000.0 - 50.0, 5.0e3, 63.0e3, 60.0e3
    xc, yc = st["lon"] - (gridres / 2.0), st["lat"] - (gridres / 2.0)
    st["ocean_depth"][:] = 1000.0 - a * np.exp(-((xc - bx) * (xc - bx) / (2.0 * cw * cw) + (yc - by) * (yc - by) / (2.0 * cw * cw)))
    p.coastal_drift = 0.1
    # the generator, makeberg/makeberg.py:241-339
    radius, rho_ice, hmax, hmin = 1.5e3, 850.0, 200.0, 200.0
    CBxc, CByc, CBxl, CByl = 50000.0, 50000.0, 15000.0, 35000.0
    xmin, xmax, ymin, ymax = CBxc - (0.5 * CBxl), CBxc + (0.5 * CBxl), CByc - (0.5 * CByl), CByc + (0.5 * CByl)
    x_start = min(xmin + (radius * 2.0 / np.sqrt(3)), xmax)
    y_start0 = min(ymin + radius, ymax)
    element_area = (3.0 * np.sqrt(3.0) / 2.0) * ((4.0 / 3.0) * (radius) ** 2)
    cdistb = np.sqrt((xmin - CBxc) ** 2 + (ymin - CByc) ** 2)
    bxs, bys, hs = [], [], []
    j, x_val = 0, x_start
    while x_val <= xmax and x_val >= xmin:
        y_start = y_start0 + ((j % 2) * radius) + 0.0
        k, y_val = 0, y_start
        while y_val <= (ymax + 0.0):
            bdistc = np.sqrt((x_val - CBxc) ** 2 + (y_val - CByc) ** 2)
            hs.append(hmin * bdistc / cdistb + hmax * (1 - bdistc / cdistb))
            bxs.append(x_val), bys.append(y_val)
            k += 1
            y_val = y_start + (2 * k * radius)
        j += 1
        x_val = x_start + (np.sqrt(3) * radius * j)
    n = len(bxs)
    w = np.sqrt(element_area)
    # the cell of a position on the shifted grid: corner lon(i) = gridres*i - 0.45
    shifted = {"x": [v + 0.45 for v in bxs], "y": [v + 0.45 for v in bys]}
    b = as_read_from_restart(grid, gridres, 45, shifted["x"], shifted["y"], hs, [w] * n, [h * rho_ice * element_area for h in hs], uvel=0.1, vvel=0.0)
    b["lon"][:], b["lat"][:] = b["lon"] - 0.45, b["lat"] - 0.45              # (positions are the generator's; only the cell search saw the shift)
    b["lon_old"][:], b["lat_old"][:] = b["lon"], b["lat"]
    x1 = (gridres * b["ine"] - 0.45) - gridres / 2.0
    y1 = (gridres * b["jne"] - 0.45) - gridres / 2.0
    b["xi"][:], b["yj"][:] = ((b["lon"] - x1) / gridres) + 0.5, ((b["lat"] - y1) / gridres) + 0.5
    bd = initialize_iceberg_bonds(b, p)                                         # manually_initialize_bonds_from_radii
    p.constant_length = p.constant_width = float(np.max(b["length"]))          # set_constant_interaction_length_and_width IB:195
    return {"grid": grid, "params": p, "bergs": b, "bonds": bd, "nsteps": 72 * 2, "gridres": gridres, "ni": 45}


def wrap_to_period(b, ni):
    """cell index of the unrolled channel -> cell index within the period (a copy of the bergs)"""
    w = S.copy_bergs(b)
    w["ine"] = (((b["ine"].astype(np.int64) - 1) % ni) + 1).astype(np.int32)
    return w


def occupancy_chksum3(b):
    """chksum3 from a berg dict (alive rows), cells in traversal order, ties in row order"""
    rows = [k for k in range(len(b["lon"])) if b["alive"][k]]
    rows.sort(key=lambda k: (b["jne"][k], b["ine"][k], b["start_year"][k], b["start_day"][k], b["start_mass"][k], b["start_lon"][k], b["start_lat"][k], k))
    return chksum3_of_occupancy([(int(b["jne"][k]), int(b["ine"][k])) for k in rows], [b["mass"][k] for k in rows])


def occupancy_partition(b):
    from collections import Counter
    c = Counter((int(j), int(i)) for i, j, a in zip(b["ine"], b["jne"], b["alive"]) if a)
    return sorted(c.values(), reverse=True)

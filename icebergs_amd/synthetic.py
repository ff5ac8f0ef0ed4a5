"""Synthetic grids, forcing and berg populations for the BASELINE.json configs.

All gridded arrays are numpy float64 of shape (nj, ni), C-contiguous, indexed [j - jsd, i - isd]: the
same bytes as the Fortran array f(isd:ied, jsd:jed) the reference owns (icebergs_framework.F90:950-1004).
The halo of the static grid is filled the way ice_bergs_framework_init does for a non-cyclic domain:
corner lon/lat by linear extrapolation (icebergs_framework.F90:1069-1094).

The population generator reproduces the reference's calving-class tables: mass / thickness /
mass_scaling per class (icebergs_framework.F90:787-790) and width = sqrt(M / (LoW * rho * T)),
length = LoW * width (icebergs_framework.F90:1540-1541).
"""
import ctypes as C

import numpy as np

from . import types as T

INITIAL_MASS = np.array([8.8e7, 4.1e8, 3.3e9, 1.8e10, 3.8e10, 7.5e10, 1.2e11, 2.2e11, 3.9e11, 7.4e11])
INITIAL_THICKNESS = np.array([40., 67., 133., 175., 250., 250., 250., 250., 250., 250.])
MASS_SCALING = np.array([2000., 200., 50., 20., 10., 5., 2., 1., 1., 1.])
INITIAL_MASS_N = np.array([4.58e8, 3.61e9, 1.22e10, 2.91e10, 5.09e10, 7.34e10, 1.15e11, 1.65e11, 2.94e11, 5.59e11])
LOW_RATIO = 1.5
RHO_BERGS = 850.0
HALO = 4  # icebergs_framework.F90:686


def default_params():
    """Namelist defaults (icebergs_framework.F90:686-822) + pinned FMS constants."""
    p = T.Params()
    p.pi, p.omega, p.HLF = 3.14159265358979323846, 7.292e-5, 3.34e5
    p.dt, p.current_year, p.current_yearday = 1800.0, 1, 0.0
    p.Rearth, p.rho_bergs, p.lat_ref = 6360000.0, 850.0, 0.0
    p.cdrag_grounding, p.h_to_init_grounding, p.ocean_drag_scale = 0.0, 100.0, 1.0
    p.speed_limit = p.sicn_shift = p.bergy_bit_erosion_fraction = p.tip_parameter = 0.0
    p.grounding_fraction = p.clipping_depth = p.coastal_drift = p.tidal_drift = 0.0
    p.initial_orientation, p.melt_cutoff = 0.0, -1.0
    p.cdrag_icebergs, p.utide_icebergs, p.ustar_icebergs_bg, p.Gamma_T_3EQ = 1.5e-3, 0.0, 0.001, 0.022
    p.fl_youngs, p.fl_strength, p.new_berg_from_fl_bits_mass_thres = 1.0e7, 250.0, 1.0e12
    for k in range(10):
        p.initial_mass_s[k] = INITIAL_MASS[k]
        p.initial_mass_n[k] = INITIAL_MASS_N[k]
    p.Runge_not_Verlet = 1
    p.old_interp_flds_order = 1
    p.old_bug_bilin = 1
    p.use_operator_splitting = 1
    p.add_weight_to_ocean = 1
    p.use_old_spreading = 1
    p.allow_bergs_to_roll = 1
    p.Use_three_equation_model = 1
    p.const_gamma = 1
    p.use_roundoff_fix = 1
    p.fl_style = T.ENUMS["KID_FL_STYLE_NEW_BERGS"]
    p.fl_bits_erosion_to_bergy_bits = 1
    p.displace_fl_bergs = 1
    # interactions / MTS / DEM (FW:693-705, 772-812)
    p.spring_coef = p.contact_spring_coef = 1.0e-8
    p.contact_distance, p.radial_damping_coef, p.tangental_damping_coef = 0.0, 1.0e-4, 2.0e-5
    p.convergence_tolerance, p.dem_damping_coef, p.poisson = 1.0e-8, 0.1, 0.3
    p.scale_damping_by_pmag = p.critical_interaction_damping_on = p.tang_crit_int_damp_on = 1
    p.contact_cells_lon = p.contact_cells_lat = 1
    p.max_bonds, p.mts_sub_steps = 6, 1
    p.rotate_icebergs_for_mass_spreading = 1
    return p


def _desc(ni_c, nj_c, latlon, regular, Lx):
    d = T.GridDesc()
    d.isc, d.iec, d.jsc, d.jec = 1, ni_c, 1, nj_c
    d.isd, d.ied, d.jsd, d.jed = 1 - HALO, ni_c + HALO, 1 - HALO, nj_c + HALO
    d.grid_is_latlon, d.grid_is_regular, d.Lx = int(latlon), int(regular), float(Lx)
    return d


def _ij(d):
    i = np.arange(d.isd, d.ied + 1, dtype=np.float64)[None, :]
    j = np.arange(d.jsd, d.jed + 1, dtype=np.float64)[:, None]
    return i, j


def zeros(d):
    return np.zeros((d.jed - d.jsd + 1, d.ied - d.isd + 1))


def cartesian_grid(ni=20, nj=20, gridres=1000.0, Lx=-1.0, depth=1000.0):
    """The stand-alone driver's synthetic Cartesian grid (driver/icebergs_driver.F90:274-286)."""
    d = _desc(ni, nj, latlon=False, regular=True, Lx=Lx)
    i, j = _ij(d)
    one = zeros(d) + 1.0
    st = {"lon": gridres * i * one, "lat": gridres * j * one}
    st["lonc"] = st["lon"] - 0.5 * gridres
    st["latc"] = st["lat"] - 0.5 * gridres
    st["dx"] = one * gridres
    st["dy"] = one * gridres
    st["area"] = one * gridres * gridres
    st["msk"] = one.copy()
    st["cos"] = one.copy()
    st["sin"] = zeros(d)
    st["ocean_depth"] = one * depth
    return {"desc": d, "static": st, "forcing": {k: zeros(d) for k in T.FORCING_NAMES}}


def c1_forcing(grid, P=20000.0):
    """SURVEY 8d config C1 analytic forcing; velocities on B-grid corners, tracers at cell centres."""
    st, f = grid["static"], grid["forcing"]
    x, y, xc, yc = st["lon"], st["lat"], st["lonc"], st["latc"]
    w = 2.0 * np.pi / P
    f["uo"][:] = 0.2 * np.sin(w * y)
    f["vo"][:] = 0.1 * np.cos(w * x)
    f["ua"][:] = 5.0
    f["va"][:] = -3.0
    f["ssh"][:] = 0.05 * np.sin(w * xc) * np.sin(w * yc)
    f["sst"][:] = -1.0 + 3.0 * np.sin(w * xc) * np.cos(w * yc)
    f["sss"][:] = -1.0  # icebergs.F90:5357 when the coupler passes no salinity
    return grid


def latlon_grid(ni=360, nj=200, lon0=0.0, dlon=1.0, lat0=-80.0, dlat=0.8, Rearth=6.36e6, continents=False):
    """SURVEY 8d config C2 grid: corners lon=i, lat=-80+0.8 j; calc_xiyj path (grid_is_regular=F)."""
    d = _desc(ni, nj, latlon=True, regular=False, Lx=360.0)
    i, j = _ij(d)
    one = zeros(d) + 1.0
    rad = np.pi / 180.0
    st = {"lon": (lon0 + dlon * i) * one, "lat": (lat0 + dlat * j) * one}
    st["lonc"] = st["lon"] - 0.5 * dlon
    st["latc"] = st["lat"] - 0.5 * dlat
    st["dx"] = Rearth * np.cos(st["lat"] * rad) * (dlon * rad)
    st["dy"] = one * Rearth * (dlat * rad)
    st["area"] = Rearth ** 2 * (dlon * rad) * np.abs(np.sin(st["lat"] * rad) - np.sin((st["lat"] - dlat) * rad))
    msk = one.copy()
    if continents:  # "blocky continents": three rectangles of land, to exercise the coast bounce
        for (i0, i1, j0, j1) in ((60, 110, 60, 150), (200, 230, 30, 90), (280, 330, 110, 170)):
            msk[(j >= j0) & (j <= j1) & (i >= i0) & (i <= i1) & (one > 0)] = 0.0
    st["msk"] = msk
    st["cos"] = one.copy()
    st["sin"] = zeros(d)
    st["ocean_depth"] = one * 4000.0
    return {"desc": d, "static": st, "forcing": {k: zeros(d) for k in T.FORCING_NAMES}}


def c2_forcing(grid):
    """zonal jet, weak meridional flow, westerly wind band, SST; no sea ice."""
    st, f = grid["static"], grid["forcing"]
    rad = np.pi / 180.0
    lon, lat, lonc, latc = st["lon"], st["lat"], st["lonc"], st["latc"]
    f["uo"][:] = 0.3 * np.cos(3.0 * lat * rad)
    f["vo"][:] = 0.05 * np.sin(2.0 * lon * rad)
    f["ua"][:] = 8.0 * np.exp(-((np.abs(lat) - 50.0) / 15.0) ** 2)
    f["va"][:] = 1.0 * np.sin(lon * rad)
    f["ssh"][:] = 0.3 * np.sin(2.0 * lonc * rad) * np.cos(2.0 * latc * rad)
    f["sst"][:] = np.maximum(-1.8, 10.0 * np.cos(latc * rad) * (1.0 + 0.2 * np.sin(3.0 * lonc * rad)) - 4.0)
    f["sss"][:] = -1.0
    land = st["msk"] < 0.5  # icebergs.F90:5364-5372 scrub
    for k in ("ua", "va", "uo", "vo", "ui", "vi", "sst", "sss", "cn", "hi"):
        f[k][land] = 0.0
    return grid


def calving_params(p, LoW_ratio=1.5, tau_calving=0.0, restarted=False):
    """kid_calving_params with the namelist defaults (icebergs_framework.F90:706, 788-796) and the derived initial
    width / length of FW:1540-1541, 1549-1550."""
    cp = T.CalvingParams()
    dist_s = (0.24, 0.12, 0.15, 0.18, 0.12, 0.07, 0.03, 0.03, 0.03, 0.02)
    dist_n = (0.14, 0.15, 0.20, 0.15, 0.08, 0.07, 0.05, 0.05, 0.05, 0.05)
    scal_s = (2000, 200, 50, 20, 10, 5, 2, 1, 1, 1)
    scal_n = (200, 50, 25, 13, 8, 5, 2, 1, 1, 1)
    thick_s = (40., 67., 133., 175., 250., 250., 250., 250., 250., 250.)
    thick_n = (80.4, 159.5, 240., 320., 360., 360., 360., 360., 360., 360.)
    for k in range(10):
        cp.distribution_s[k], cp.distribution_n[k] = dist_s[k], dist_n[k]
        cp.mass_scaling_s[k], cp.mass_scaling_n[k] = scal_s[k], scal_n[k]
        cp.initial_thickness_s[k], cp.initial_thickness_n[k] = thick_s[k], thick_n[k]
        cp.initial_width_s[k] = np.sqrt(p.initial_mass_s[k] / (LoW_ratio * p.rho_bergs * thick_s[k]))
        cp.initial_width_n[k] = np.sqrt(p.initial_mass_n[k] / (LoW_ratio * p.rho_bergs * thick_n[k]))
        cp.initial_length_s[k] = LoW_ratio * cp.initial_width_s[k]
        cp.initial_length_n[k] = LoW_ratio * cp.initial_width_n[k]
    cp.tau_calving, cp.restarted = tau_calving, int(restarted)
    return cp


def coupler_calving(grid, seed=0, frac=0.05, buckets=1.5, dt=1800.0):
    """Synthetic calving / calving_hflx arguments of icebergs_run (kg/m2/s, W/m2) over (isc:iec, jsc:jec): a fraction
    `frac` of the cells calves, each delivering up to `buckets` class-1 buckets (8.8e7 kg x 2000 / 0.24) per step of
    length dt, so that buckets of several classes overflow within a few steps."""
    d = grid["desc"]
    nic, njc = d.iec - d.isc + 1, d.jec - d.jsc + 1
    area = grid["static"]["area"][d.jsc - d.jsd:d.jec - d.jsd + 1, d.isc - d.isd:d.iec - d.isd + 1]
    rng = np.random.default_rng(seed)
    on = rng.random((njc, nic)) < frac
    calving = np.where(on, rng.uniform(0.2, 1.0, (njc, nic)) * buckets * (8.8e7 * 2000 / 0.24) / (dt * area), 0.0)
    hflx = np.where(on, -rng.uniform(0.5, 1.5, (njc, nic)) * calving * 2.0e4, 0.0)   # cold ice: a negative heat flux
    return calving, hflx


def coupler_forcing(grid, seed=0, vel_stagger="B", stress_stagger="B", symmetric=False, kelvin=False, sss=True, nans=True):
    """Synthetic arguments of icebergs_run (icebergs.F90:5074-5096) for a grid: what the coupler hands over BEFORE the
    ingest block, with the extents each stagger implies (symmetric memory adds the extra C-grid row / column).
    Arrays are (n2, n1) numpy = Fortran a(n1, n2)."""
    d = grid["desc"]
    nic, njc = d.iec - d.isc + 1, d.jec - d.jsc + 1
    rng = np.random.default_rng(seed)
    f = lambda n2, n1, s: rng.normal(0.0, s, (n2, n1))
    if vel_stagger == "B":
        ushape = vshape = (njc + 2, nic + 2)
    else:   # C-grid: data-domain sized arrays, one more u column / v row with symmetric memory (IB:5245)
        ushape = (njc + 2, nic + 2 + (1 if symmetric else 0))
        vshape = (njc + 2 + (1 if symmetric else 0), nic + 2)
    if stress_stagger == "C":
        txshape = (njc + 2, nic + 2 + (1 if symmetric else 0))
        tyshape = (njc + 2 + (1 if symmetric else 0), nic + 2)
    else:
        txshape = tyshape = (njc, nic)
    a = {"uo": f(*ushape, 0.3), "ui": f(*ushape, 0.1), "vo": f(*vshape, 0.3), "vi": f(*vshape, 0.1),
         "tauxa": f(*txshape, 0.2), "tauya": f(*tyshape, 0.2),
         "ssh": f(njc + 2, nic + 2, 0.5), "cn": rng.uniform(0, 1, (njc + 2, nic + 2)), "hi": rng.uniform(0, 2, (njc + 2, nic + 2)),
         "sst": rng.uniform(-1.5, 25.0, (njc, nic)) + (273.15 if kelvin else 0.0)}
    if sss:
        a["sss"] = rng.uniform(30.0, 36.0, (njc, nic))
    if nans:   # what the scrub (IB:5373-5382) is there for; a zero stress exercises the cddvmod == 0 branch
        for name in ("uo", "vi", "cn", "tauxa"):
            a[name][rng.integers(1, a[name].shape[0] - 1), rng.integers(1, a[name].shape[1] - 1)] = np.nan
        a["tauxa"][2, 3] = 0.0
        a["tauya"][2, 3] = 0.0
    return a


def empty_bergs(n):
    b = {name: np.zeros(n) for name in T.BERG_F64_NAMES}
    for name in T.BERG_I32_NAMES:
        b[name] = np.zeros(n, dtype=np.int32)
    b["alive"][:] = 1
    b["id"] = np.arange(1, n + 1, dtype=np.int64)
    return b


def _fill_classes(b, klass, jitter=None):
    M = INITIAL_MASS[klass].copy()
    Tk = INITIAL_THICKNESS[klass].copy()
    if jitter is not None:
        M *= jitter
    W = np.sqrt(M / (LOW_RATIO * RHO_BERGS * Tk))
    b["mass"][:] = M
    b["thickness"][:] = Tk
    b["width"][:] = W
    b["length"][:] = LOW_RATIO * W
    b["mass_scaling"][:] = MASS_SCALING[klass]
    b["start_mass"][:] = M
    return b


def sort_reference_order(b):
    """cells j-outer / i-inner (icebergs.F90:7106), `inorder` inside a cell
    (icebergs_framework.F90:4318-4359)."""
    order = np.lexsort((b["start_lat"], b["start_lon"], b["start_mass"], b["start_day"], b["start_year"],
                        b["ine"], b["jne"]))
    for k in list(b.keys()):
        b[k] = np.ascontiguousarray(b[k][order])
    return b


def place_bergs(grid, n, seed, i_range, j_range, klass=None, wet_only=True):
    """Uniform random positions inside cells [i_range] x [j_range]; (xi, yj) from the cell's own corners."""
    d, st = grid["desc"], grid["static"]
    rng = np.random.Generator(np.random.PCG64(seed))
    b = empty_bergs(n)
    ine = rng.integers(i_range[0], i_range[1] + 1, size=n)
    jne = rng.integers(j_range[0], j_range[1] + 1, size=n)
    if wet_only:
        for _ in range(64):
            bad = st["msk"][jne - d.jsd, ine - d.isd] < 0.5
            if not bad.any():
                break
            ine[bad] = rng.integers(i_range[0], i_range[1] + 1, size=int(bad.sum()))
            jne[bad] = rng.integers(j_range[0], j_range[1] + 1, size=int(bad.sum()))
    xi = rng.uniform(0.02, 0.98, size=n)
    yj = rng.uniform(0.02, 0.98, size=n)
    lon0 = st["lon"][jne - d.jsd, ine - 1 - d.isd]
    lon1 = st["lon"][jne - d.jsd, ine - d.isd]
    lat0 = st["lat"][jne - 1 - d.jsd, ine - d.isd]
    lat1 = st["lat"][jne - d.jsd, ine - d.isd]
    b["lon"][:] = lon0 + xi * (lon1 - lon0)
    b["lat"][:] = lat0 + yj * (lat1 - lat0)
    # in-cell coordinates as pos_within_cell would return them for a rectangular cell
    b["xi"][:] = (b["lon"] - lon0) / (lon1 - lon0)
    b["yj"][:] = (b["lat"] - lat0) / (lat1 - lat0)
    b["ine"][:] = ine
    b["jne"][:] = jne
    if klass is None:
        klass = rng.integers(0, 10, size=n)
    _fill_classes(b, klass)
    b["start_lon"][:] = b["lon"]
    b["start_lat"][:] = b["lat"]
    b["start_year"][:] = 0
    b["start_day"][:] = 1.0e-6 * np.arange(n)  # makes the `inorder` sort total
    b["lon_old"][:] = b["lon"]
    b["lat_old"][:] = b["lat"]
    return sort_reference_order(b)


def config_c1(n=10, seed=1):
    """BASELINE config 1: 10 point bergs, 20x20 f-plane Cartesian grid, analytic forcing."""
    grid = c1_forcing(cartesian_grid(20, 20, 1000.0, Lx=-1.0))
    p = default_params()
    p.dt, p.lat_ref, p.use_f_plane = 600.0, -70.0, 1
    b = place_bergs(grid, n, seed, (4, 17), (4, 17), klass=np.arange(n) % 10)
    return grid, p, b


def config_c2(n=1_000_000, seed=2, continents=False):
    """BASELINE config 2: n bergs (random mass classes), 360x200 lat-lon grid, drag+Coriolis+melt."""
    grid = c2_forcing(latlon_grid(continents=continents))
    p = default_params()
    p.dt = 1800.0
    # |lat| < 75 -> j in [7, 193]; keep clear of the periodic seam (berg migration is out of scope, SURVEY #13)
    b = place_bergs(grid, n, seed, (6, 355), (8, 192))
    return grid, p, b


def config_c3(n=100, seed=3, ni=64, nj=20, fl_style="fl_bits", capacity_factor=3, dt=600.0, spread=False, displace=False, periodic=False,
              by_pe=False):
    """BASELINE config 3, the footloose profile of tests/footloose_tests/input.nml: Cartesian 1 km grid, Verlet,
    footloose calving into FL bits (or child bergs), 300 m thick tabular bergs with +-20 % jitter, ocean 1 m/s east,
    wind stress -1 (about -25.8 m/s as a wind), SST -0.5.  The arrays carry spare rows for children (b["_n"] live).
    fl_k starts at a random fraction of a foot so that calving starts within tens of steps.
    displace: displace_fl_bergs=T as in the profile's own namelist (children placed on the parent's perimeter with the
    counter-based generator, include/kid_rng.h); periodic: the channel is zonally periodic (Lx = ni km) as in the profile."""
    grid = cartesian_grid(ni, nj, 1000.0, Lx=(ni * 1000.0 if periodic else -1.0))
    F = grid["forcing"]
    F["uo"][:] = 1.0
    jmid = F["vo"].shape[0] // 2
    F["vo"][:jmid] = 0.1
    F["vo"][jmid:] = -0.1
    F["ua"][:] = -25.8
    F["sst"][:] = -0.5
    F["sss"][:] = 34.0
    p = default_params()
    p.dt, p.lat_ref, p.use_f_plane = dt, -70.0, 1
    p.Runge_not_Verlet = 0
    p.footloose, p.displace_fl_bergs = 1, (1 if displace else 0)
    p.fl_init_child_xy_by_pe, p.fl_rng_seed = (1 if by_pe else 0), 20240807
    p.periodic_reentry = 1 if periodic else 0
    p.fl_style = T.ENUMS["KID_FL_STYLE_FL_BITS" if fl_style == "fl_bits" else "KID_FL_STYLE_NEW_BERGS"]
    p.fl_youngs, p.fl_strength = 1.0e8, 250.0
    p.new_berg_from_fl_bits_mass_thres = 3.0e11
    p.bergy_bit_erosion_fraction = 1.0
    p.use_updated_rolling_scheme, p.allow_bergs_to_roll = 1, 1
    p.old_bug_bilin, p.use_old_spreading, p.add_weight_to_ocean = 0, 0, 0
    p.use_new_predictive_corrective, p.const_gamma = 1, 0
    p.ustar_icebergs_bg = 0.0
    p.apply_thickness_cutoff_to_gridded_melt, p.apply_thickness_cutoff_to_bergs_melt, p.melt_cutoff = 1, 1, 10.0
    p.old_interp_flds_order = 0
    rng = np.random.default_rng(seed)
    cap = int(max(capacity_factor * n, n + 64))
    b = empty_bergs(cap)
    b["alive"][n:] = 0
    d = grid["desc"]
    if spread:  # throughput runs: the whole interior, in reference (cell) order
        i = rng.integers(d.isc + 3, d.iec - 2, size=n).astype(np.int32)
    else:
        i = rng.integers(d.isc + 3, d.isc + max(4, (d.iec - d.isc) // 3), size=n).astype(np.int32)  # western third: they drift east
    j = rng.integers(d.jsc + 4, d.jec - 3, size=n).astype(np.int32)
    xi, yj = rng.uniform(0.05, 0.95, n), rng.uniform(0.05, 0.95, n)
    b["ine"][:n], b["jne"][:n], b["xi"][:n], b["yj"][:n] = i, j, xi, yj
    b["lon"][:n] = (i - 1 + xi) * 1000.0
    b["lat"][:n] = (j - 1 + yj) * 1000.0
    Tk = 300.0 * rng.uniform(0.8, 1.2, n)
    W = rng.uniform(3000.0, 6000.0, n)
    L = W * rng.uniform(1.0, 1.5, n)
    b["thickness"][:n], b["width"][:n], b["length"][:n] = Tk, W, L
    b["mass"][:n] = Tk * W * L * RHO_BERGS
    b["start_mass"][:n] = b["mass"][:n]
    b["mass_scaling"][:n] = 1.0
    b["start_lon"][:n], b["start_lat"][:n] = b["lon"][:n], b["lat"][:n]
    b["lon_old"][:n], b["lat_old"][:n] = b["lon"][:n], b["lat"][:n]
    b["start_year"][:n] = 1
    b["start_day"][:n] = rng.uniform(0.0, 10.0, n)
    b["fl_k"][:n] = rng.uniform(0.0, 6.0e4, n)
    if fl_style == "fl_bits":  # some bergs already carry FL bits close to the new-berg threshold
        b["mass_of_fl_bits"][:n] = np.where(rng.uniform(size=n) < 0.3, rng.uniform(1.0e11, 2.9e11, n), 0.0)
        b["mass_of_fl_bergy_bits"][:n] = 0.01 * b["mass_of_fl_bits"][:n]
    b["uvel"][:n] = 0.5
    b["uvel_old"][:n] = 0.5
    b["_n"] = n
    if spread:
        o = np.lexsort((b["ine"][:n], b["jne"][:n]))
        for k, v in b.items():
            if hasattr(v, "dtype"):
                v[:n] = v[:n][o]
    return grid, p, b


def empty_bonds(n, max_bonds=6):
    """Bond lists of n bergs flattened slot-major (include/kid_types.h kid_bond_soa)."""
    bd = {"max_bonds": max_bonds, "count": np.zeros(n, dtype=np.int32),
          "other_id": np.zeros(max_bonds * n, dtype=np.int64), "broken": np.zeros(max_bonds * n, dtype=np.int32)}
    for name in T.BOND_F64_NAMES:
        bd[name] = np.zeros(max_bonds * n)
    return bd


def bond_neighbours(b, n, reach, max_bonds=6):
    """Bond every pair of bergs closer than `reach` (what initialize_iceberg_bonds does for a packed conglomerate,
    IB:356-441); a berg's bonds are listed by increasing partner row."""
    bd = empty_bonds(len(b["lon"]), max_bonds)
    N = len(b["lon"])
    x, y = b["lon"][:n], b["lat"][:n]
    from scipy.spatial import cKDTree
    tree = cKDTree(np.column_stack([x, y]))
    pairs = tree.query_ball_point(np.column_stack([x, y]), reach * (1.0 - 1e-12))
    for k in range(n):
        near = np.array(sorted(o for o in pairs[k] if o != k), dtype=np.int64)
        assert len(near) <= max_bonds, (k, len(near))
        for s, o in enumerate(near):
            bd["other_id"][s * N + k] = b["id"][o]
        bd["count"][k] = len(near)
        b["n_bonds"][k] = len(near)
    return bd


def dem_ground_frac_elements(radius=1.5e3, xc=50000.0, yc=50000.0, xl=15000.0, yl=35000.0):
    """Element centres of the reference's tests/dem_ground_frac_test: its generator's parameters (makeberg/makeberg.py:241-257:
    one conglomerate, a 15 km x 35 km rectangle centred on (50 km, 50 km), element radius 1.5 km) and its hexagonal packing
    loop (:284-325: columns sqrt(3) r apart starting 2r/sqrt(3) inside the western edge, every other column shifted north by r,
    elements 2r apart from r inside the southern edge up to the northern edge) restated -- 69 elements, the '#=69' of the
    regression line recorded in that test's namelist (input.nml:7, 10)."""
    xmin, xmax, ymin, ymax = xc - 0.5 * xl, xc + 0.5 * xl, yc - 0.5 * yl, yc + 0.5 * yl
    xs, ys = [], []
    x_start, y_start0 = min(xmin + radius * 2.0 / np.sqrt(3.0), xmax), min(ymin + radius, ymax)
    j, x_val = 0, x_start
    while xmin <= x_val <= xmax:
        y_start = y_start0 + (j % 2) * radius
        k, y_val = 0, y_start
        while y_val <= ymax:
            xs.append(x_val)
            ys.append(y_val)
            k += 1
            y_val = y_start + 2 * k * radius
        j += 1
        x_val = x_start + np.sqrt(3.0) * radius * j
    return xs, ys


def config_c4(nx=5, ny=11, hexagonal=True, radius=1500.0, thickness=200.0, ni=45, nj=45, gridres=5000.0, sub_steps=200,
              bump=(58.0e3, 60.0e3), bump_depth=50.0, origin=(44137.0, 35211.0), frac=(1850.0, 1000.0), thickness_jitter=0.0,
              seed=4, two_bergs=False, dem=True, explicit_inner=True, spring_coef=None, dt=1800.0, mts=True, contact=True,
              reference_pattern=False):
    """BASELINE config 4 family: a tabular berg made of bonded DEM elements (hexagonal or square packing) drifting at
    0.1 m/s onto a Gaussian seamount on the Cartesian grid of tests/dem_ground_frac_test (driver DRV:288-307,
    namelist tests/dem_ground_frac_test/input.nml): MTS velocity Verlet with explicit DEM sub-steps, stress fracture
    on the sub-steps, broken bonds kept for contact, grounding drag on the sub-steps.
    Differences from that namelist: coastal_drift=0 and no land rows; melt rates are computed (set_melt_rates_to_zero=T
    there)."""
    grid = cartesian_grid(ni, nj, gridres, Lx=-1.0)
    d = grid["desc"]
    ii, jj = _ij(d)
    xc, yc = gridres * ii - gridres / 2.0, gridres * jj - gridres / 2.0
    if reference_pattern:
        bump = (63.0e3, 60.0e3)   # the driver's own seamount for this test (DRV:299), 5.5 km east of the conglomerate's edge
    a, cw = 1000.0 - bump_depth, 5.0e3
    grid["static"]["ocean_depth"][:] = 1000.0 - a * np.exp(-((xc - bump[0]) ** 2 / (2 * cw * cw) + (yc - bump[1]) ** 2 / (2 * cw * cw)))
    F = grid["forcing"]
    F["uo"][:] = 0.1
    F["vo"][:] = 0.1
    F["sst"][:] = -1.0
    F["sss"][:] = 34.0
    p = default_params()
    p.dt, p.lat_ref, p.use_f_plane = 1800.0, 0.0, 0
    p.Runge_not_Verlet, p.old_interp_flds_order, p.use_new_predictive_corrective = 0, 0, 1
    p.old_bug_bilin, p.use_old_spreading = 0, 0
    p.mts, p.dem, p.explicit_inner_mts, p.mts_sub_steps = 1, 1, 1, sub_steps
    p.iceberg_bonds_on, p.interactive_icebergs_on, p.internal_bergs_for_drag = 1, 1, 1
    p.use_broken_bonds_for_substep_contact, p.break_bonds_on_sub_steps, p.short_step_mts_grounding = 1, 1, 1
    p.constant_interaction_LW, p.force_convergence, p.convergence_tolerance = 1, 1, 1.0e-2
    p.hexagonal_icebergs = 1 if hexagonal else 0
    p.max_bonds = 6 if hexagonal else 4
    p.poisson, p.dem_damping_coef, p.dem_spring_coef = 0.3, 1.0, 5.0e6
    p.spring_coef = 0.0007547062342348049
    p.contact_distance, p.contact_spring_coef = 4.0e3, 5.0e-8
    p.contact_cells_lon = p.contact_cells_lat = 1  # ceil(4e3 / 5e3)
    p.cdrag_grounding, p.h_to_init_grounding = 1.0e4, 0.0
    p.fracture_criterion_stress, p.frac_thres_n, p.frac_thres_t = 1, frac[0], frac[1]
    p.radial_damping_coef = p.tangental_damping_coef = 0.0
    p.scale_damping_by_pmag, p.critical_interaction_damping_on, p.tang_crit_int_damp_on = 0, 1, 0
    p.use_updated_rolling_scheme, p.allow_bergs_to_roll = 1, 1
    p.ustar_icebergs_bg, p.const_gamma = 0.0, 0
    p.apply_thickness_cutoff_to_gridded_melt, p.apply_thickness_cutoff_to_bergs_melt, p.melt_cutoff = 1, 1, 10.0
    # elements
    xs, ys = [], []
    if reference_pattern:   # the element pattern of tests/dem_ground_frac_test itself (69 elements); melt off as in its namelist (:128)
        p.set_melt_rates_to_zero = 1
        xs, ys = dem_ground_frac_elements(radius)
        area = (3.0 * np.sqrt(3.0) / 2.0) * ((4.0 / 3.0) * radius ** 2)
        nx, ny = len(xs), 1
    elif hexagonal:
        for jcol in range(nx):
            for krow in range(ny):
                xs.append(origin[0] + radius * 2.0 / np.sqrt(3.0) + np.sqrt(3.0) * radius * jcol)
                ys.append(origin[1] + radius + (jcol % 2) * radius + 2 * krow * radius)
        area = (3.0 * np.sqrt(3.0) / 2.0) * ((4.0 / 3.0) * radius ** 2)
    else:
        for jcol in range(nx):
            for krow in range(ny):
                xs.append(origin[0] + radius + 2 * radius * jcol)
                ys.append(origin[1] + radius + 2 * radius * krow)
        area = (2.0 * radius) ** 2
    if two_bergs:  # a second, smaller conglomerate just east of the first: collisions between conglomerates
        x0 = max(xs) + 3.2 * radius
        for jcol in range(2):
            for krow in range(3):
                xs.append(x0 + 2 * radius * jcol)
                ys.append(origin[1] + radius + 2 * radius * (krow + ny // 2 - 1))
    n = len(xs)
    first = nx * ny
    rng = np.random.default_rng(seed)
    b = empty_bergs(n)
    b["lon"][:], b["lat"][:] = xs, ys
    b["ine"][:] = np.floor(b["lon"] / gridres).astype(np.int32) + 1
    b["jne"][:] = np.floor(b["lat"] / gridres).astype(np.int32) + 1
    b["xi"][:] = b["lon"] / gridres - (b["ine"] - 1)
    b["yj"][:] = b["lat"] / gridres - (b["jne"] - 1)
    w = np.sqrt(area)
    Tk = thickness * (1.0 + thickness_jitter * rng.uniform(-1.0, 1.0, n))
    b["thickness"][:], b["width"][:], b["length"][:] = Tk, w, w
    b["mass"][:] = Tk * RHO_BERGS * area
    b["start_mass"][:] = b["mass"]
    b["mass_scaling"][:] = 1.0
    b["uvel"][:first] = 0.1
    b["uvel"][first:] = -0.05
    for f in ("uvel_old", "uvel_prev"):
        b[f][:] = b["uvel"]
    b["lon_old"][:], b["lat_old"][:] = b["lon"], b["lat"]
    b["start_lon"][:], b["start_lat"][:] = b["lon"], b["lat"]
    b["start_year"][:] = 1
    b["start_day"][:] = 1.0e-6 * np.arange(n)
    p.constant_length = p.constant_width = float(w)
    if not dem:   # Stern et al. (2017) KID springs between bonded elements instead of the DEM bonds (tests/collision_tests/input_MTS_KID.nml)
        p.dem, p.explicit_inner_mts = 0, 1 if explicit_inner else 0
        p.use_broken_bonds_for_substep_contact = p.break_bonds_on_sub_steps = p.short_step_mts_grounding = 0
        p.fracture_criterion_stress, p.frac_thres_n, p.frac_thres_t = 0, 0.0, 0.0
        p.radial_damping_coef, p.tangental_damping_coef = 1.0e-4, 2.0e-5
        p.convergence_tolerance = 1.0e-8
    if spring_coef is not None:
        p.spring_coef = spring_coef
    if not mts:   # single-time-step KID (tests/collision_tests/input_KID.nml): Verlet, springs + implicit damping inside accel
        p.mts, p.dem, p.explicit_inner_mts, p.force_convergence, p.mts_sub_steps = 0, 0, 0, 0, 1
        p.old_interp_flds_order = 1
    if not contact:   # Stern et al.'s original interaction: 3x3 cells, every berg, no separate contact spring
        p.contact_distance, p.contact_spring_coef = 0.0, p.spring_coef
    p.dt = dt
    b = sort_reference_order(b)
    bd = bond_neighbours(b, n, 2.0 * radius * 1.05, p.max_bonds)
    return grid, p, b, bd


def config_beam(kind="cantilever"):
    """The reference's DEM beam tests (Wang 2020, sections 3.1 and 3.2; tests/dem_ssbeam_test, tests/dem_cbeam_test): the
    generator's parameters (makeberg/makeberg.py main()) and the namelists (input.nml) restated.
      cantilever: 3 rows of 30 square-packed elements 5 km apart (90, the '#=90' of input.nml:2,5), thickness 1 m, the first
        element of every row static, the load 1.5e10/3 N on the last element of every row (dem_beam_test=2, IB:1870-1877);
        dt = 100 s, 2000 sub-steps, dem_damping_coef 0.7.
      supported:  one row of 29 elements 0.5 m apart ('#=29', input.nml:1), the two ends carry no vertical load and the
        centre element is loaded with 1.5e5 N (dem_beam_test=1, IB:1862-1869); dt = 1 s, 1e5 sub-steps, damping 0.1.
    Both: only_interactive_forces, dem_spring_coef 1e9 (Young's modulus), poisson 0.3, bonds from the radii (4 neighbours),
    melt rates zero, a 20x20 grid of 15 km cells with the ocean at rest.  dem_tests_init (FW:4687-4710: start_lon = lon for
    every berg, the west-most and east-most start_lon) is done here, on the host, as icebergs_init does it."""
    gridres, ni, nj = 15000.0, 20, 20
    grid = cartesian_grid(ni, nj, gridres, Lx=-1.0)
    F = grid["forcing"]
    F["sst"][:] = -1.0
    F["sss"][:] = 34.0
    p = default_params()
    p.lat_ref, p.use_f_plane = 0.0, 0
    p.Runge_not_Verlet, p.old_interp_flds_order, p.use_new_predictive_corrective = 0, 0, 1
    p.old_bug_bilin, p.use_old_spreading = 0, 0
    p.mts, p.dem, p.explicit_inner_mts = 1, 1, 1
    p.iceberg_bonds_on, p.interactive_icebergs_on, p.internal_bergs_for_drag, p.only_interactive_forces = 1, 1, 1, 1
    p.use_broken_bonds_for_substep_contact, p.break_bonds_on_sub_steps, p.short_step_mts_grounding = 0, 0, 0
    p.constant_interaction_LW, p.force_convergence, p.convergence_tolerance = 0, 1, 1.0e-8
    p.orig_dem_moment_of_inertia = 1
    p.hexagonal_icebergs, p.max_bonds = 0, 4
    p.poisson, p.dem_spring_coef = 0.3, 1.0e9
    p.spring_coef = 1.0e-5
    p.contact_distance, p.contact_spring_coef = 2000.0, 1.0e-8
    p.contact_cells_lon = p.contact_cells_lat = 1
    p.cdrag_grounding, p.h_to_init_grounding = 3.16e6, 200.0
    p.fracture_criterion_stress, p.frac_thres_n, p.frac_thres_t = 0, 0.0, 0.0
    p.radial_damping_coef = p.tangental_damping_coef = 0.0
    p.scale_damping_by_pmag, p.critical_interaction_damping_on, p.tang_crit_int_damp_on = 0, 0, 0
    p.use_updated_rolling_scheme, p.allow_bergs_to_roll = 0, 0
    p.set_melt_rates_to_zero = 1
    p.ustar_icebergs_bg, p.const_gamma = 0.0, 0
    p.apply_thickness_cutoff_to_gridded_melt, p.apply_thickness_cutoff_to_bergs_melt, p.melt_cutoff = 1, 1, 10.0
    xs0, ys0, h = 101.0e3, 151.0e3, 1.0
    xs, ys, static = [], [], []
    if kind == "cantilever":
        r, rho, nrow, ncol = 2500.0, 900.0, 3, 30
        p.dt, p.mts_sub_steps, p.dem_damping_coef, p.dem_beam_test = 100.0, 2000, 0.7, 2
        for row in range(nrow):
            for col in range(ncol):
                xs.append(xs0 + 2 * r * col)
                ys.append(ys0 + 2 * r * row)
                static.append(1.0 if col == 0 else 0.0)
    else:
        r, rho, ncol = 0.25, 800.0, 29
        p.dt, p.mts_sub_steps, p.dem_damping_coef, p.dem_beam_test = 1.0, 100000, 0.1, 1
        for col in range(ncol):
            xs.append(xs0 + 2 * r * col)
            ys.append(ys0 + 2 * r)
            static.append(0.0)
    p.rho_bergs = rho
    n = len(xs)
    area = (2.0 * r) ** 2
    b = empty_bergs(n)
    b["lon"][:], b["lat"][:] = xs, ys
    b["ine"][:] = np.floor(b["lon"] / gridres).astype(np.int32) + 1
    b["jne"][:] = np.floor(b["lat"] / gridres).astype(np.int32) + 1
    b["xi"][:] = b["lon"] / gridres - (b["ine"] - 1)
    b["yj"][:] = b["lat"] / gridres - (b["jne"] - 1)
    w = np.sqrt(area)
    b["thickness"][:], b["width"][:], b["length"][:] = h, w, w
    b["mass"][:] = h * rho * area
    b["start_mass"][:] = b["mass"]
    b["mass_scaling"][:] = 1.0
    b["static_berg"][:] = static
    b["lon_old"][:], b["lat_old"][:] = b["lon"], b["lat"]
    b["start_lon"][:], b["start_lat"][:] = b["lon"], b["lat"]      # dem_tests_init
    p.dem_tests_start_lon, p.dem_tests_end_lon = float(min(xs)), float(max(xs))
    b["start_year"][:] = 1
    b["start_day"][:] = 1.0e-6 * np.arange(n)
    p.constant_length = p.constant_width = float(w)
    b = sort_reference_order(b)
    bd = bond_neighbours(b, n, 2.0 * r * 1.05, p.max_bonds)
    return grid, p, b, bd


def copy_bonds(bd):
    return {k: (v.copy() if hasattr(v, "copy") else v) for k, v in bd.items()}


def set_diag_all(p):
    p.diag_mask = (1 << 20) - 1
    return p


def copy_bergs(b):
    return {k: (v.copy() if hasattr(v, "copy") else v) for k, v in b.items()}


def params_copy(p):
    q = T.Params()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(T.Params))
    return q

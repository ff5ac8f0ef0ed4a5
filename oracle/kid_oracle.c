/* kid_oracle.c -- CPU restatement (ORACLE) of the NOAA-GFDL/icebergs per-berg evolve loop.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
 * checker / the timed CPU baseline.  The product (icebergs_amd/) never includes, links or calls this file.
 *
 * What it is: a plain scalar C restatement, in the reference's own operation order, of
 *   src/icebergs.F90            (IB)  accel, interp_flds, Runge_Kutta_stepping, verlet_stepping,
 *                                     update_verlet_position, adjust_index_and_ground, thermodynamics, rolling,
 *                                     fl_bits_dimensions, find_basal_melt, spread_mass_across_ocean_cells,
 *                                     hexagon geometry, calculate_mass_on_ocean, sum_up_spread_fields,
 *                                     create_gridded_icebergs_fields, footloose_calving
 *   src/icebergs_framework.F90  (FW)  bilin, apply_modulo_around_point, is_point_in_cell, sum_sign_dot_prod4/5,
 *                                     calc_xiyj, pos_within_cell, inorder
 * Every function cites the reference lines it follows.  Build with -ffp-contract=off (no FMA fusion) so that
 * the arithmetic is the plain IEEE sequence the Fortran source spells out.
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   - The reference itself cannot be built here: both source files `use` FMS (mpp_mod, fms_mod,
 *     mpp_domains_mod, time_manager_mod, diag_manager_mod, random_numbers_mod, constants_mod), which is not in
 *     this image and not vendored under /root/reference; writing stand-in modules for a missing library is
 *     not allowed, so there is no oracle/_ref and no reference-generated golden vectors.
 *   - Pinned against the reference's OWN known-answer tests (tests/test_oracle_pins.py):
 *       hexagon_test            IB:247-353  (all 7 cases, tol 1e-10)        -> ko_hexagon_into_quadrants
 *       point_in_triangle_test  IB:226-244                                   -> ko_point_in_triangle
 *       basal_melt_test         IB:205-223  inputs; printed values recorded in SURVEY.md section 4
 *                               (4.33063180897577E-06, 7.090487055660092E-06) -> ko_find_basal_melt
 *       unit_tests              FW:7299-7327 bilin corner identities          -> ko_bilin
 *   - PARITY UNPINNED for everything the reference has no stored vector for: accel / RK4 / Verlet
 *     trajectories, adjust_index_and_ground, thermodynamics melt masses, rectangular mass spreading.  For
 *     those the oracle is a line-by-line restatement checked only by code reading and by internal
 *     consistency properties (tests/test_oracle_properties.py).
 *   - FMS constants (pi, omega, HLF) are pinned to FMS's published values in ko_default_params().
 *   - footloose child placement uses FMS's Mersenne-Twister stream in the reference (IB:2548-2550, 2631);
 *     it is not restated: the oracle and the product support displace_fl_bergs=.false. only.
 */
#include "kid_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* reference module constants, IB:68-80 */
#define RHO_ICE      916.7
#define RHO_WATER    999.8
#define RHO_AIR      1.1
#define RHO_SEAWATER 1025.0
#define GRAVITY      9.8
#define CD_AV 1.3
#define CD_AH 0.0055
#define CD_WV 0.9
#define CD_WH 0.0012
#define CD_IV 0.9

#define NI(g) ((g)->d.ied - (g)->d.isd + 1)
#define GIDX(g, i, j) ((size_t)((i) - (g)->d.isd) + (size_t)((j) - (g)->d.jsd) * (size_t)NI(g))
#define GS(g, F, i, j) ((g)->stat[F][GIDX(g, i, j)])
#define GF(g, F, i, j) ((g)->forc[F][GIDX(g, i, j)])

static inline double dmin(double a, double b) { return a < b ? a : b; }
static inline double dmax(double a, double b) { return a > b ? a : b; }
static inline double fsign1(double b) { return copysign(1.0, b); } /* Fortran sign(1.,b) */

/* ------------------------------------------------------------------------------------------------
 * constants_mod (FMS, not in the reference tree) + namelist defaults FW:686-822
 * ---------------------------------------------------------------------------------------------- */
void ko_default_params(kid_params *p) {
  static const double im_s[10] = {8.8e7, 4.1e8, 3.3e9, 1.8e10, 3.8e10, 7.5e10, 1.2e11, 2.2e11, 3.9e11, 7.4e11};
  static const double im_n[10] = {4.58e8, 3.61e9, 1.22e10, 2.91e10, 5.09e10, 7.34e10, 1.15e11, 1.65e11, 2.94e11, 5.59e11};
  memset(p, 0, sizeof(*p));
  p->pi = 3.14159265358979323846; p->omega = 7.292e-5; p->HLF = 3.34e5; /* FMS constants_mod */
  p->dt = 1800.0; p->current_year = 1; p->current_yearday = 0.0;
  p->Rearth = 6360000.0; p->rho_bergs = 850.0; p->lat_ref = 0.0;
  p->cdrag_grounding = 0.0; p->h_to_init_grounding = 100.0; p->ocean_drag_scale = 1.0;
  p->speed_limit = 0.0; p->sicn_shift = 0.0; p->bergy_bit_erosion_fraction = 0.0; p->tip_parameter = 0.0;
  p->grounding_fraction = 0.0; p->clipping_depth = 0.0; p->coastal_drift = 0.0; p->tidal_drift = 0.0;
  p->initial_orientation = 0.0; p->melt_cutoff = -1.0; p->cdrag_icebergs = 1.5e-3; p->utide_icebergs = 0.0;
  p->ustar_icebergs_bg = 0.001; p->Gamma_T_3EQ = 0.022; p->fl_youngs = 1.e7; p->fl_strength = 250.0;
  p->new_berg_from_fl_bits_mass_thres = 1.e12;
  memcpy(p->initial_mass_s, im_s, sizeof(im_s)); memcpy(p->initial_mass_n, im_n, sizeof(im_n));
  p->Runge_not_Verlet = 1; p->use_new_predictive_corrective = 0; p->old_interp_flds_order = 1; p->old_bug_bilin = 1;
  p->use_f_plane = 0; p->use_operator_splitting = 1; p->add_weight_to_ocean = 1; p->time_average_weight = 0;
  p->use_old_spreading = 1; p->hexagonal_icebergs = 0; p->allow_bergs_to_roll = 1; p->use_updated_rolling_scheme = 0;
  p->Use_three_equation_model = 1; p->const_gamma = 1; p->use_roundoff_fix = 1;
  p->fl_style = KID_FL_STYLE_NEW_BERGS; p->fl_bits_erosion_to_bergy_bits = 1; p->displace_fl_bergs = 1;
  p->diag_mask = 0;
  /* interactions / MTS / DEM, FW:693-705, 772-812 */
  p->spring_coef = 1.e-8; p->contact_spring_coef = 1.e-8; p->contact_distance = 0.; p->radial_damping_coef = 1.e-4;
  p->tangental_damping_coef = 2.e-5; p->convergence_tolerance = 1.e-8; p->dem_damping_coef = 0.1; p->poisson = 0.3;
  p->scale_damping_by_pmag = 1; p->critical_interaction_damping_on = 1; p->tang_crit_int_damp_on = 1;
  p->contact_cells_lon = 1; p->contact_cells_lat = 1; p->max_bonds = 6; p->mts_sub_steps = 1;
  p->rotate_icebergs_for_mass_spreading = 1;
}

/* ------------------------------------------------------------------------------------------------
 * FW:6558-6573 apply_modulo_around_point; Fortran MODULO(a,p) = a - floor(a/p)*p with the sign of p
 * ---------------------------------------------------------------------------------------------- */
double ko_modulo(double a, double p) {
  double r = fmod(a, p);
  if (r != 0.0 && ((r < 0.0) != (p < 0.0))) r += p;
  return r;
}
double ko_apply_modulo_around_point(double x, double y, double Lx) {
  if (Lx > 0.) {
    double Lx_2 = Lx / 2.;
    return ko_modulo(x - (y - Lx_2), Lx) + (y - Lx_2);
  }
  return x;
}

/* FW:7071-7088 bilin (old_bug_bilin selects the inverted weights) */
double ko_bilin(const ko_grid *g, const kid_params *p, const double *fld, int i, int j, double xi, double yj) {
  double f11 = fld[GIDX(g, i, j)], f01 = fld[GIDX(g, i - 1, j)];
  double f10 = fld[GIDX(g, i, j - 1)], f00 = fld[GIDX(g, i - 1, j - 1)];
  if (p->old_bug_bilin)
    return (f11 * (1. - xi) + f01 * xi) * (1. - yj) + (f10 * (1. - xi) + f00 * xi) * yj;
  return (f11 * xi + f01 * (1. - xi)) * yj + (f10 * xi + f00 * (1. - xi)) * (1. - yj);
}

/* FW:6163-6228 */
int ko_sum_sign_dot_prod4(double x0, double y0, double x1, double y1, double x2, double y2,
                          double x3, double y3, double x, double y, double Lx) {
  double xx = ko_apply_modulo_around_point(x, x0, Lx);
  double xx0 = ko_apply_modulo_around_point(x0, x0, Lx);
  double xx1 = ko_apply_modulo_around_point(x1, x0, Lx);
  double xx2 = ko_apply_modulo_around_point(x2, x0, Lx);
  double xx3 = ko_apply_modulo_around_point(x3, x0, Lx);
  double l0 = (xx - xx0) * (y1 - y0) - (y - y0) * (xx1 - xx0);
  double l1 = (xx - xx1) * (y2 - y1) - (y - y1) * (xx2 - xx1);
  double l2 = (xx - xx2) * (y3 - y2) - (y - y2) * (xx3 - xx2);
  double l3 = (xx - xx3) * (y0 - y3) - (y - y3) * (xx0 - xx3);
  double p0 = fsign1(l0); if (l0 == 0.) p0 = -0.5;
  double p1 = fsign1(l1); if (l1 == 0.) p1 = 0.5;
  double p2 = fsign1(l2); if (l2 == 0.) p2 = 0.5;
  double p3 = fsign1(l3); if (l3 == 0.) p3 = -0.5;
  return ((fabs(p0) + fabs(p2)) + (fabs(p1) + fabs(p3))) == fabs((p0 + p2) + (p1 + p3));
}
/* FW:6231-6296 */
int ko_sum_sign_dot_prod5(double x0, double y0, double x1, double y1, double x2, double y2,
                          double x3, double y3, double x4, double y4, double x, double y, double Lx) {
  double xx = ko_apply_modulo_around_point(x, x0, Lx);
  double xx0 = ko_apply_modulo_around_point(x0, x0, Lx);
  double xx1 = ko_apply_modulo_around_point(x1, x0, Lx);
  double xx2 = ko_apply_modulo_around_point(x2, x0, Lx);
  double xx3 = ko_apply_modulo_around_point(x3, x0, Lx);
  double xx4 = ko_apply_modulo_around_point(x4, x0, Lx);
  double l0 = (xx - xx0) * (y1 - y0) - (y - y0) * (xx1 - xx0);
  double l1 = (xx - xx1) * (y2 - y1) - (y - y1) * (xx2 - xx1);
  double l2 = (xx - xx2) * (y3 - y2) - (y - y2) * (xx3 - xx2);
  double l3 = (xx - xx3) * (y4 - y3) - (y - y3) * (xx4 - xx3);
  double l4 = (xx - xx4) * (y0 - y4) - (y - y4) * (xx0 - xx4);
  double p0 = fsign1(l0); if (l0 == 0.) p0 = 0.;
  double p1 = fsign1(l1); if (l1 == 0.) p1 = 0.;
  double p2 = fsign1(l2); if (l2 == 0.) p2 = 0.;
  double p3 = fsign1(l3); if (l3 == 0.) p3 = 0.;
  double p4 = fsign1(l4); if (l4 == 0.) p4 = 0.;
  return (((fabs(p0) + fabs(p2)) + (fabs(p1) + fabs(p3))) + fabs(p4) - fabs(((p0 + p2) + (p1 + p3)) + p4)) < 0.5;
}

/* FW:6076-6160 */
int ko_is_point_in_cell(const ko_grid *g, double x, double y, int i, int j) {
  const double Lx = g->d.Lx;
  if (i - 1 < g->d.isd || i > g->d.ied || j - 1 < g->d.jsd || j > g->d.jed) return 0; /* FATAL in the reference */
  const double lon00 = GS(g, KID_G_LON, i - 1, j - 1), lon10 = GS(g, KID_G_LON, i, j - 1);
  const double lon01 = GS(g, KID_G_LON, i - 1, j), lon11 = GS(g, KID_G_LON, i, j);
  const double lat00 = GS(g, KID_G_LAT, i - 1, j - 1), lat10 = GS(g, KID_G_LAT, i, j - 1);
  const double lat01 = GS(g, KID_G_LAT, i - 1, j), lat11 = GS(g, KID_G_LAT, i, j);
  double a = ko_apply_modulo_around_point(lon00, x, Lx), b = ko_apply_modulo_around_point(lon10, x, Lx);
  double c = ko_apply_modulo_around_point(lon01, x, Lx), d = ko_apply_modulo_around_point(lon11, x, Lx);
  double xlo = dmin(dmin(dmin(a, b), c), d);
  double xhi = dmax(dmax(dmax(a, b), c), d);
  const double tol = 0.1;
  if (x < (xlo - tol) || x > (xhi + tol)) return 0;
  double ylo = dmin(dmin(dmin(lat00, lat10), lat01), lat11);
  double yhi = dmax(dmax(dmax(lat00, lat10), lat01), lat11);
  if (y < ylo || y > yhi) return 0;
  const int ll = g->d.grid_is_latlon;
  if (lat11 > 89.999 && ll)
    return ko_sum_sign_dot_prod5(lon00, lat00, lon10, lat10, lon10, lat11, lon01, lat11, lon01, lat01, x, y, Lx);
  else if (lat01 > 89.999 && ll)
    return ko_sum_sign_dot_prod5(lon00, lat00, lon10, lat10, lon11, lat11, lon11, lat01, lon00, lat01, x, y, Lx);
  else if (lat00 > 89.999 && ll)
    return ko_sum_sign_dot_prod5(lon01, lat00, lon10, lat00, lon10, lat10, lon11, lat11, lon01, lat01, x, y, Lx);
  else if (lat10 > 89.999 && ll)
    return ko_sum_sign_dot_prod5(lon00, lat00, lon00, lat10, lon11, lat10, lon11, lat11, lon01, lat01, x, y, Lx);
  return ko_sum_sign_dot_prod4(lon00, lat00, lon10, lat10, lon11, lat11, lon01, lat01, x, y, Lx);
}

/* FW:6439-6534; returns non-zero on the reference's FATAL paths (complex roots / singular) */
int ko_calc_xiyj(double x1, double x2, double x3, double x4, double y1, double y2, double y3, double y4,
                 double x, double y, double *xi, double *yj, double Lx) {
  double alpha = x2 - x1, delta = y2 - y1, beta = x4 - x1, epsilon = y4 - y1;
  double gamma = (x3 - x1) - (alpha + beta);
  double kappa = (y3 - y1) - (delta + epsilon);
  double a = (kappa * beta - gamma * epsilon);
  double dx = ko_apply_modulo_around_point(x, x1, Lx) - x1;
  double dy = y - y1;
  double b = (delta * beta - alpha * epsilon) - (kappa * dx - gamma * dy);
  double c = (alpha * dy - delta * dx);
  int err = 0;
  if (fabs(a) > 1.e-12) {
    double d = 0.25 * (b * b) - a * c;
    if (d >= 0.) {
      double yy1 = -(0.5 * b + sqrt(d)) / a;
      double yy2 = -(0.5 * b - sqrt(d)) / a;
      if (fabs(yy1 - 0.5) < fabs(yy2 - 0.5)) *yj = yy1; else *yj = yy2;
    } else { err = 1; *yj = -999.; }
  } else {
    if (b != 0.) *yj = -c / b; else *yj = 0.;
  }
  a = (alpha + gamma * (*yj));
  b = (delta + kappa * (*yj));
  if (a != 0.) *xi = (dx - beta * (*yj)) / a;
  else if (b != 0.) *xi = (dy - epsilon * (*yj)) / b;
  else {
    c = (epsilon * alpha - beta * delta) + (epsilon * gamma - beta * kappa) * (*yj);
    if (c != 0.) *xi = (epsilon * dx - beta * dy) / c; else { err = 1; *xi = -999.; }
  }
  return err;
}

static int within_xi_yj_bounds(double xi, double yj) { /* FW:6540-6552 */
  return (xi >= 0. && xi < 1.) && (yj >= 0. && yj < 1.);
}

/* FW:6299-6436 */
int ko_pos_within_cell(const ko_grid *g, const kid_params *p, double x, double y, int i, int j,
                       double *xi, double *yj, int *err) {
  const double Lx = g->d.Lx;
  const double pi_180 = p->pi / 180.;
  *xi = -999.; *yj = -999.;
  if (i - 1 < g->d.isd) return 0;
  if (j - 1 < g->d.jsd) return 0;
  if (i > g->d.ied) return 0;
  if (j > g->d.jed) return 0;
  double x1 = GS(g, KID_G_LON, i - 1, j - 1), y1 = GS(g, KID_G_LAT, i - 1, j - 1);
  double x2 = GS(g, KID_G_LON, i, j - 1), y2 = GS(g, KID_G_LAT, i, j - 1);
  double x3 = GS(g, KID_G_LON, i, j), y3 = GS(g, KID_G_LAT, i, j);
  double x4 = GS(g, KID_G_LON, i - 1, j), y4 = GS(g, KID_G_LAT, i - 1, j);
  if (!g->d.grid_is_latlon && g->d.grid_is_regular) {
    double dx = fabs(GS(g, KID_G_LON, i, j) - GS(g, KID_G_LON, i - 1, j));
    double dy = fabs(GS(g, KID_G_LAT, i, j) - GS(g, KID_G_LAT, i, j - 1));
    x1 = GS(g, KID_G_LON, i, j) - (dx / 2);
    y1 = GS(g, KID_G_LAT, i, j) - (dy / 2);
    double Delta_x = ko_apply_modulo_around_point(x, x1, Lx) - x1;
    *xi = ((Delta_x) / dx) + 0.5;
    *yj = ((y - y1) / dy) + 0.5;
  } else if (dmax(dmax(dmax(y1, y2), y3), y4) < 89.999 || !g->d.grid_is_latlon) {
    if (ko_calc_xiyj(x1, x2, x3, x4, y1, y2, y3, y4, x, y, xi, yj, Lx) && err) *err = 1;
  } else {
    double xx = (90. - y) * cos(x * pi_180), yy = (90. - y) * sin(x * pi_180);
    double l00 = GS(g, KID_G_LON, i - 1, j - 1), l10 = GS(g, KID_G_LON, i, j - 1);
    double l11 = GS(g, KID_G_LON, i, j), l01 = GS(g, KID_G_LON, i - 1, j);
    x1 = (90. - y1) * cos(l00 * pi_180); y1 = (90. - y1) * sin(l00 * pi_180);
    x2 = (90. - y2) * cos(l10 * pi_180); y2 = (90. - y2) * sin(l10 * pi_180);
    x3 = (90. - y3) * cos(l11 * pi_180); y3 = (90. - y3) * sin(l11 * pi_180);
    x4 = (90. - y4) * cos(l01 * pi_180); y4 = (90. - y4) * sin(l01 * pi_180);
    if (ko_calc_xiyj(x1, x2, x3, x4, y1, y2, y3, y4, xx, yy, xi, yj, Lx) && err) *err = 1;
    if (ko_is_point_in_cell(g, x, y, i, j)) {
      if (!within_xi_yj_bounds(*xi, *yj)) {
        double fac = 2.1 * dmax(fabs(*xi - 0.5), fabs(*yj - 0.5)); fac = dmax(1., fac);
        *xi = 0.5 + (*xi - 0.5) / fac;
        *yj = 0.5 + (*yj - 0.5) / fac;
      }
    } else {
      if (fabs(*xi - 0.5) < 0.5 && fabs(*yj - 0.5) < 0.5) { if (err) *err = 1; } /* FATAL FW:6402 */
    }
  }
  return ko_is_point_in_cell(g, x, y, i, j);
}

/* ------------------------------------------------------------------------------------------------
 * IB:4903-4926 ddx_ssh / ddy_ssh
 * ---------------------------------------------------------------------------------------------- */
static double ddx_ssh(const ko_grid *g, int i, int j) {
  double dxp = 0.5 * (GS(g, KID_G_DX, i + 1, j) + GS(g, KID_G_DX, i + 1, j - 1));
  double dx0 = 0.5 * (GS(g, KID_G_DX, i, j) + GS(g, KID_G_DX, i, j - 1));
  return 2. * (GF(g, KID_F_SSH, i + 1, j) - GF(g, KID_F_SSH, i, j)) / (dx0 + dxp) * GS(g, KID_G_MSK, i + 1, j) * GS(g, KID_G_MSK, i, j);
}
static double ddy_ssh(const ko_grid *g, int i, int j) {
  double dyp = 0.5 * (GS(g, KID_G_DY, i, j + 1) + GS(g, KID_G_DY, i - 1, j + 1));
  double dy0 = 0.5 * (GS(g, KID_G_DY, i, j) + GS(g, KID_G_DY, i - 1, j));
  return 2. * (GF(g, KID_F_SSH, i, j + 1) - GF(g, KID_F_SSH, i, j)) / (dy0 + dyp) * GS(g, KID_G_MSK, i, j + 1) * GS(g, KID_G_MSK, i, j);
}
static void rotate(double *u, double *v, double cos_rot, double sin_rot) { /* IB:4953-4967 */
  double u_old = *u, v_old = *v;
  *u = cos_rot * u_old + sin_rot * v_old;
  *v = cos_rot * v_old - sin_rot * u_old;
}

/* env order: uo,vo,ui,vi,ua,va,ssh_x,ssh_y,sst,sss,cn,hi,od  (== KID_B_UO..KID_B_OD) */
enum { E_UO = 0, E_VO, E_UI, E_VI, E_UA, E_VA, E_SSHX, E_SSHY, E_SST, E_SSS, E_CN, E_HI, E_OD };

/* IB:4718-4900 interp_flds (non-MTS: od = ocean_depth+ssh PCM; tidal_drift=0 (rx=ry=0)) */
void ko_interp_flds(const ko_grid *g, const kid_params *p, double x, double y, int i, int j, double xi, double yj,
                    double env[13]) {
  double cos_rot = ko_bilin(g, p, g->stat[KID_G_COS], i, j, xi, yj);
  double sin_rot = ko_bilin(g, p, g->stat[KID_G_SIN], i, j, xi, yj);
  double uo = ko_bilin(g, p, g->forc[KID_F_UO], i, j, xi, yj);
  double vo = ko_bilin(g, p, g->forc[KID_F_VO], i, j, xi, yj);
  double ui = ko_bilin(g, p, g->forc[KID_F_UI], i, j, xi, yj);
  double vi = ko_bilin(g, p, g->forc[KID_F_VI], i, j, xi, yj);
  double ua = ko_bilin(g, p, g->forc[KID_F_UA], i, j, xi, yj);
  double va = ko_bilin(g, p, g->forc[KID_F_VA], i, j, xi, yj);
  if (p->coastal_drift > 0.) {
    double mE = GS(g, KID_G_MSK, i + 1, j), mW = GS(g, KID_G_MSK, i - 1, j), m0 = GS(g, KID_G_MSK, i, j);
    double mN = GS(g, KID_G_MSK, i, j + 1), mS = GS(g, KID_G_MSK, i, j - 1);
    uo = uo + p->coastal_drift * (mE - mW) * m0;
    ui = ui + p->coastal_drift * (mE - mW) * m0;
    vo = vo + p->coastal_drift * (mN - mS) * m0;
    vi = vi + p->coastal_drift * (mN - mS) * m0;
  }
  double sst = GF(g, KID_F_SST, i, j), sss = GF(g, KID_F_SSS, i, j);
  double cn = GF(g, KID_F_CN, i, j), hi = GF(g, KID_F_HI, i, j);
  double hxp, hxm;
  if (yj >= 0.5) {
    hxp = (yj - 0.5) * ddx_ssh(g, i, j + 1) + (1.5 - yj) * ddx_ssh(g, i, j);
    hxm = (yj - 0.5) * ddx_ssh(g, i - 1, j + 1) + (1.5 - yj) * ddx_ssh(g, i - 1, j);
  } else {
    hxp = (yj + 0.5) * ddx_ssh(g, i, j) + (0.5 - yj) * ddx_ssh(g, i, j - 1);
    hxm = (yj + 0.5) * ddx_ssh(g, i - 1, j) + (0.5 - yj) * ddx_ssh(g, i - 1, j - 1);
  }
  double ssh_x = xi * hxp + (1. - xi) * hxm;
  if (xi >= 0.5) {
    hxp = (xi - 0.5) * ddy_ssh(g, i + 1, j) + (1.5 - xi) * ddy_ssh(g, i, j);
    hxm = (xi - 0.5) * ddy_ssh(g, i + 1, j - 1) + (1.5 - xi) * ddy_ssh(g, i, j - 1);
  } else {
    hxp = (xi + 0.5) * ddy_ssh(g, i, j) + (0.5 - xi) * ddy_ssh(g, i - 1, j);
    hxm = (xi + 0.5) * ddy_ssh(g, i, j - 1) + (0.5 - xi) * ddy_ssh(g, i - 1, j - 1);
  }
  double ssh_y = yj * hxp + (1. - yj) * hxm;
  rotate(&uo, &vo, cos_rot, sin_rot);
  rotate(&ui, &vi, cos_rot, sin_rot);
  rotate(&ua, &va, cos_rot, sin_rot);
  rotate(&ssh_x, &ssh_y, cos_rot, sin_rot);
  if (ssh_x != ssh_x) ssh_x = 0.;
  if (ssh_y != ssh_y) ssh_y = 0.;
  env[E_UO] = uo; env[E_VO] = vo; env[E_UI] = ui; env[E_VI] = vi; env[E_UA] = ua; env[E_VA] = va;
  env[E_SSHX] = ssh_x; env[E_SSHY] = ssh_y; env[E_SST] = sst; env[E_SSS] = sss; env[E_CN] = cn; env[E_HI] = hi;
  if (p->mts) env[E_OD] = ko_quad_interp_depth(g, p, x, y, i, j, xi, yj); /* IB:4894 */
  else env[E_OD] = GS(g, KID_G_OCEAN_DEPTH, i, j) + GF(g, KID_F_SSH, i, j); /* IB:4897 */
}

/* ------------------------------------------------------------------------------------------------
 * IB:1950-2442 accel.  bergstate[] is indexed by KID_B_* (mass, thickness, width, length and, when
 * .not.old_interp_flds_order, the stored environment).  Non-interactive bergs only.
 * ---------------------------------------------------------------------------------------------- */
void ko_accel(const ko_grid *g, const kid_params *p, const double bs[], int n_bonds,
              int i, int j, double xi, double yj, double lat, double uvel, double vvel, double uvel0, double vvel0,
              double dt, double *ax_o, double *ay_o, double *axn_io, double *ayn_io, double *bxn_o, double *byn_o,
              int64_t *ntickets) {
  const double pi_180 = p->pi / 180.;
  const int RK = p->Runge_not_Verlet;
  int use_new_pc = p->use_new_predictive_corrective;
  double alpha = 0.0, beta = 1.0, C_N = 0.0;
  if (!RK) { alpha = 1.0; C_N = 1.0; beta = 1.0; use_new_pc = 1; }
  double axn = *axn_io, ayn = *ayn_io, bxn, byn, ax = 0., ay = 0.;
  const double u_star = uvel0 + (axn * (dt / 2.));
  const double v_star = vvel0 + (ayn * (dt / 2.));
  double env[13];
  if (p->old_interp_flds_order) ko_interp_flds(g, p, bs[KID_B_LON], bs[KID_B_LAT], i, j, xi, yj, env);
  else for (int k = 0; k < 13; ++k) env[k] = bs[KID_B_UO + k];
  double uo = env[E_UO], vo = env[E_VO], ui = env[E_UI], vi = env[E_VI], ua = env[E_UA], va = env[E_VA];
  double ssh_x = env[E_SSHX], ssh_y = env[E_SSHY], hi = env[E_HI], od = env[E_OD];
  double f_cori;
  if (g->d.grid_is_latlon && !p->use_f_plane) f_cori = (2. * p->omega) * sin(pi_180 * lat);
  else f_cori = (2. * p->omega) * sin(pi_180 * p->lat_ref);
  const double M = bs[KID_B_MASS], T = bs[KID_B_THICKNESS];
  const double D = (p->rho_bergs / RHO_SEAWATER) * T;
  const double F = T - D;
  const double W = bs[KID_B_WIDTH], L = bs[KID_B_LENGTH];
  axn = 0.; ayn = 0.; bxn = 0.; byn = 0.;
  hi = dmin(hi, D);
  const double D_hi = dmax(0., D - hi);
  double groundfrac, c_gnd;
  if (p->h_to_init_grounding > 0.0) {
    groundfrac = 1.0 - (od - D) / p->h_to_init_grounding;
    groundfrac = dmax(groundfrac, 0.0); groundfrac = dmin(groundfrac, 1.0);
  } else groundfrac = (D > od) ? 1.0 : 0.0;
  if (groundfrac > 0.0) c_gnd = (p->cdrag_grounding * W * L * groundfrac) / M; else c_gnd = 0.0;
  /* wave radiation IB:2085-2102 */
  double uwave = ua - uo, vwave = va - vo;
  double wmod = uwave * uwave + vwave * vwave;
  const double ampl = 0.5 * 0.02025 * wmod;
  const double Lwavelength = 0.32 * wmod;
  const double Lcutoff = 0.125 * Lwavelength, Ltop = 0.25 * Lwavelength;
  const double Cr0 = 0.06;
  const double Cr = Cr0 * dmin(dmax(0., (L - Lcutoff) / ((Ltop - Lcutoff) + 1.e-30)), 1.);
  double wave_rad = 0.5 * RHO_SEAWATER / M * Cr * GRAVITY * ampl * dmin(ampl, F) * (2. * W * L) / (W + L);
  wmod = sqrt(ua * ua + va * va);
  if (wmod != 0.) { uwave = ua / wmod; vwave = va / wmod; }
  else { uwave = 0.; vwave = 0.; wave_rad = 0.; }
  double dragfrac = 1.0;
  if (p->iceberg_bonds_on && p->internal_bergs_for_drag) {
    double N_max = p->hexagonal_icebergs ? 6.0 : 4.0;
    dragfrac = ((N_max - (double)n_bonds) / N_max);
  }
  const double c_ocn = RHO_SEAWATER / M * p->ocean_drag_scale * (0.5 * CD_WV * dragfrac * W * (D_hi) + CD_WH * W * L);
  const double c_atm = RHO_AIR / M * (0.5 * CD_AV * dragfrac * W * F + CD_AH * W * L);
  double c_ice;
  if (fabs(hi) == 0.) c_ice = 0.; else c_ice = RHO_ICE / M * (0.5 * CD_IV * dragfrac * W * hi);
  if (fabs(ui) + fabs(vi) == 0.) c_ice = 0.;
  if (!RK) { axn = -GRAVITY * ssh_x + wave_rad * uwave; ayn = -GRAVITY * ssh_y + wave_rad * vwave; }
  else     { bxn = -GRAVITY * ssh_x + wave_rad * uwave; byn = -GRAVITY * ssh_y + wave_rad * vwave; }
  if (alpha > 0.) {
    if (C_N > 0.) { axn = axn + f_cori * v_star; ayn = ayn - f_cori * u_star; }
    else          { bxn = bxn + f_cori * v_star; byn = byn - f_cori * u_star; }
  } else { bxn = bxn + f_cori * vvel; byn = byn - f_cori * uvel; }
  double uveln, vveln;
  if (use_new_pc) { uveln = uvel0; vveln = vvel0; } else { uveln = uvel; vveln = vvel; }
  double us, vs, drag_ocn, drag_atm, drag_ice, drag_gnd, RHS_x, RHS_y;
  for (int itloop = 1; itloop <= 2; ++itloop) {
    if (use_new_pc) {
      drag_ocn = c_ocn * 0.5 * (sqrt((uveln - uo) * (uveln - uo) + (vveln - vo) * (vveln - vo)) + sqrt((uvel0 - uo) * (uvel0 - uo) + (vvel0 - vo) * (vvel0 - vo)));
      drag_atm = c_atm * 0.5 * (sqrt((uveln - ua) * (uveln - ua) + (vveln - va) * (vveln - va)) + sqrt((uvel0 - ua) * (uvel0 - ua) + (vvel0 - va) * (vvel0 - va)));
      drag_ice = c_ice * 0.5 * (sqrt((uveln - ui) * (uveln - ui) + (vveln - vi) * (vveln - vi)) + sqrt((uvel0 - ui) * (uvel0 - ui) + (vvel0 - vi) * (vvel0 - vi)));
      drag_gnd = c_gnd;
    } else {
      us = 0.5 * (uveln + uvel); vs = 0.5 * (vveln + vvel);
      drag_ocn = c_ocn * sqrt((us - uo) * (us - uo) + (vs - vo) * (vs - vo));
      drag_atm = c_atm * sqrt((us - ua) * (us - ua) + (vs - va) * (vs - va));
      drag_ice = c_ice * sqrt((us - ui) * (us - ui) + (vs - vi) * (vs - vi));
      drag_gnd = c_gnd;
    }
    RHS_x = (axn / 2) + bxn;
    RHS_y = (ayn / 2) + byn;
    if (beta > 0.) {
      RHS_x = RHS_x - drag_ocn * (u_star - uo) - drag_atm * (u_star - ua) - drag_ice * (u_star - ui) - drag_gnd * u_star;
      RHS_y = RHS_y - drag_ocn * (v_star - vo) - drag_atm * (v_star - va) - drag_ice * (v_star - vi) - drag_gnd * v_star;
    } else {
      RHS_x = RHS_x - drag_ocn * (uvel - uo) - drag_atm * (uvel - ua) - drag_ice * (uvel - ui) - drag_gnd * uvel;
      RHS_y = RHS_y - drag_ocn * (vvel - vo) - drag_atm * (vvel - va) - drag_ice * (vvel - vi) - drag_gnd * vvel;
    }
    if (alpha + beta > 0.) {
      double lambda = drag_ocn + drag_atm + drag_ice + drag_gnd;
      double A11 = 1. + beta * dt * lambda, A22 = 1. + beta * dt * lambda;
      double A12 = -alpha * dt * f_cori, A21 = alpha * dt * f_cori;
      if (C_N > 0.) { A12 = A12 / 2.; A21 = A21 / 2.; }
      double detA = 1. / ((A11 * A22) - (A12 * A21));
      ax = detA * (A22 * RHS_x - A12 * RHS_y);
      ay = detA * (A11 * RHS_y - A21 * RHS_x);
    } else { ax = RHS_x; ay = RHS_y; }
    uveln = u_star + dt * ax;
    vveln = v_star + dt * ay;
  }
  axn = 0.; ayn = 0.;
  if (!RK) { axn = -GRAVITY * ssh_x + wave_rad * uwave; ayn = -GRAVITY * ssh_y + wave_rad * vwave; }
  if (C_N > 0.) { axn = axn + f_cori * vveln; ayn = ayn - f_cori * uveln; }
  bxn = ax - (axn / 2); byn = ay - (ayn / 2);
  /* CFL limiter: only the ticket counter has an effect (uveln/vveln are locals) IB:2304-2323 */
  if (p->speed_limit > 0. || p->speed_limit == -1.) {
    double speed = sqrt(uveln * uveln + vveln * vveln);
    if (speed > 0.) {
      double loc_dx = dmin(0.5 * (GS(g, KID_G_DX, i, j) + GS(g, KID_G_DX, i, j - 1)), 0.5 * (GS(g, KID_G_DY, i, j) + GS(g, KID_G_DY, i - 1, j)));
      double new_speed = loc_dx / dt * p->speed_limit;
      if (new_speed < speed && p->speed_limit > 0. && ntickets) *ntickets += 1;
    }
  }
  if (p->override_iceberg_velocities) { ax = 0.; ay = 0.; axn = 0.; ayn = 0.; bxn = 0.; byn = 0.; }
  *ax_o = ax; *ay_o = ay; *axn_io = axn; *ayn_io = ayn; *bxn_o = bxn; *byn_o = byn;
}

/* ------------------------------------------------------------------------------------------------
 * IB:7819-8063 adjust_index_and_ground (debug=.false.)
 * ---------------------------------------------------------------------------------------------- */
void ko_adjust_index_and_ground(const ko_grid *g, const kid_params *p, double *lon, double *lat,
                                int *ip, int *jp, double *xi, double *yj, int *bounced_o, int *err) {
  const double posn_eps = 0.05;
  int bounced = 0;
  int i = *ip, j = *jp;
  const int i0 = i, j0 = j;
  int lret = ko_pos_within_cell(g, p, *lon, *lat, i, j, xi, yj, err);
  if (lret) { *bounced_o = 0; return; }
  int icount = 0;
  lret = ko_pos_within_cell(g, p, *lon, *lat, i0, j0, xi, yj, err);
  while (!lret && icount < 4) {
    icount++;
    if (*xi < 0.) {
      if (i > g->d.isd) {
        if (GS(g, KID_G_MSK, i - 1, j) > 0.) { if (i > g->d.isd + 1) i = i - 1; }
        else bounced = 1;
      }
    } else if (*xi >= 1.) {
      if (i < g->d.ied) {
        if (GS(g, KID_G_MSK, i + 1, j) > 0.) { if (i < g->d.ied) i = i + 1; }
        else bounced = 1;
      }
    }
    if (*yj < 0.) {
      if (j > g->d.jsd) {
        if (GS(g, KID_G_MSK, i, j - 1) > 0.) { if (j > g->d.jsd + 1) j = j - 1; }
        else bounced = 1;
      }
    } else if (*yj >= 1.) {
      if (j < g->d.jed) {
        if (GS(g, KID_G_MSK, i, j + 1) > 0.) { if (j < g->d.jed) j = j + 1; }
        else bounced = 1;
      }
    }
    if (bounced) {
      if (*xi >= 1.) *xi = 1. - posn_eps;
      if (*xi < 0.) *xi = posn_eps;
      if (*yj >= 1.) *yj = 1. - posn_eps;
      if (*yj < 0.) *yj = posn_eps;
      *lon = ko_bilin(g, p, g->stat[KID_G_LON], i, j, *xi, *yj);
      *lat = ko_bilin(g, p, g->stat[KID_G_LAT], i, j, *xi, *yj);
    }
    lret = ko_pos_within_cell(g, p, *lon, *lat, i, j, xi, yj, err);
  }
  *ip = i; *jp = j; *bounced_o = bounced;
  if (!bounced && lret && GS(g, KID_G_MSK, i, j) > 0.) return;
  if (!bounced && !lret) {
    if (abs(i - i0) + abs(j - j0) == 0) {
      if (p->use_roundoff_fix) {
        *xi = (*xi - 0.5) * (1. - posn_eps) + 0.5;
        *yj = (*yj - 0.5) * (1. - posn_eps) + 0.5;
      }
    }
  }
  if (*xi >= 1.) *xi = 1. - posn_eps;
  if (*xi < 0.) *xi = posn_eps;
  if (*yj > 1.) *yj = 1. - posn_eps;
  if (*yj <= 0.) *yj = posn_eps;
  *lon = ko_bilin(g, p, g->stat[KID_G_LON], i, j, *xi, *yj);
  *lat = ko_bilin(g, p, g->stat[KID_G_LAT], i, j, *xi, *yj);
  lret = ko_pos_within_cell(g, p, *lon, *lat, i, j, xi, yj, err);
  (void)lret;
}

/* IB:462-477 */
void ko_meters_to_grid(const ko_grid *g, const kid_params *p, double lat_ref, double *dlon_dx, double *dlat_dy) {
  if (g->d.grid_is_latlon) {
    *dlon_dx = (180. / p->pi) / (p->Rearth * cos((lat_ref) * (p->pi / 180.)));
    *dlat_dy = (180. / p->pi) / p->Rearth;
  } else { *dlon_dx = 1.; *dlat_dy = 1.; }
}
/* tangent plane helpers IB:7767-7816, 8066-8099 */
void ko_rotpos_to_tang(const kid_params *p, double lon, double lat, double *x, double *y) {
  const double pi_180 = p->pi / 180.;
  double colat = 90. - lat;
  double r = p->Rearth * (colat * pi_180);
  *x = r * cos(lon * pi_180); *y = r * sin(lon * pi_180);
}
void ko_rotpos_from_tang(const kid_params *p, double x, double y, double *lon, double *lat) {
  const double r180_pi = 180. / p->pi;
  double r = sqrt(x * x + y * y);
  *lat = 90. - (r180_pi * r / p->Rearth);
  *lon = r180_pi * acos(x / r) * fsign1(y);
}
void ko_rotvec_to_tang(const kid_params *p, double lon, double uvel, double vvel, double *xdot, double *ydot) {
  const double pi_180 = p->pi / 180.;
  double clon = cos(lon * pi_180), slon = sin(lon * pi_180);
  *xdot = -slon * uvel - clon * vvel;
  *ydot = clon * uvel - slon * vvel;
}
void ko_rotvec_from_tang(const kid_params *p, double lon, double xdot, double ydot, double *uvel, double *vvel) {
  const double pi_180 = p->pi / 180.;
  double clon = cos(lon * pi_180), slon = sin(lon * pi_180);
  *uvel = -slon * xdot + clon * ydot;
  *vvel = -clon * xdot - slon * ydot;
}

/* local AoS view of one berg */
static void load_berg(const kid_berg_soa *b, int64_t k, double bs[KID_NB_F64]) {
  for (int f = 0; f < KID_NB_F64; ++f) bs[f] = b->f64[f] ? b->f64[f][k] : 0.0;
}
static int berg_alive(const kid_berg_soa *b, int64_t k) { return b->i32[KID_BI_ALIVE] ? b->i32[KID_BI_ALIVE][k] != 0 : 1; }
static int berg_nbonds(const kid_berg_soa *b, int64_t k) { return b->i32[KID_BI_N_BONDS] ? b->i32[KID_BI_N_BONDS][k] : 0; }
#define PUT(b, F, k, v) do { if ((b)->f64[F]) (b)->f64[F][k] = (v); } while (0)

/* IB:7331-7679 Runge_Kutta_stepping */
static void rk4_step(const ko_grid *g, const kid_params *p, const double bs[], int nb, int ine, int jne,
                     double *axn_o, double *ayn_o, double *bxn_o, double *byn_o, double *uveln, double *vveln,
                     double *lonn, double *latn, int *io, int *jo, double *xio, double *yjo, int64_t *tick, int *err) {
  const double dt = p->dt, dt_2 = 0.5 * dt, dt_6 = dt / 6.;
  int i = ine, j = jne; double xi = bs[KID_B_XI], yj = bs[KID_B_YJ];
  int bounced = 0;
  const int on_tang = (bs[KID_B_LAT] > 89.) && g->d.grid_is_latlon;
  const int i1 = i, j1 = j;
  double axn = bs[KID_B_AXN], ayn = bs[KID_B_AYN], bxn = 0., byn = 0.;
  double axn1 = axn, axn2 = axn, axn3 = axn, axn4 = axn, ayn1 = ayn, ayn2 = ayn, ayn3 = ayn, ayn4 = ayn;
  double lon1 = bs[KID_B_LON], lat1 = bs[KID_B_LAT], x1 = 0, y1 = 0;
  if (on_tang) ko_rotpos_to_tang(p, lon1, lat1, &x1, &y1);
  double dxdl1, dydl; ko_meters_to_grid(g, p, lat1, &dxdl1, &dydl);
  double uvel1 = bs[KID_B_UVEL], vvel1 = bs[KID_B_VVEL], xdot1 = 0, ydot1 = 0;
  if (on_tang) ko_rotvec_to_tang(p, lon1, uvel1, vvel1, &xdot1, &ydot1);
  double u1 = uvel1 * dxdl1, v1 = vvel1 * dydl;
  double ax1, ay1, xddot1 = 0, yddot1 = 0, xddot1n = 0, yddot1n = 0;
  ko_accel(g, p, bs, nb, i, j, xi, yj, lat1, uvel1, vvel1, uvel1, vvel1, dt_2, &ax1, &ay1, &axn1, &ayn1, &bxn, &byn, tick);
  if (on_tang) { ko_rotvec_to_tang(p, lon1, ax1, ay1, &xddot1, &yddot1); ko_rotvec_to_tang(p, lon1, axn1, ayn1, &xddot1n, &yddot1n); }
  /* stage 2 */
  double lon2, lat2, uvel2, vvel2, x2, y2, xdot2 = 0, ydot2 = 0;
  if (on_tang) {
    x2 = x1 + dt_2 * xdot1; y2 = y1 + dt_2 * ydot1;
    xdot2 = xdot1 + dt_2 * xddot1; ydot2 = ydot1 + dt_2 * yddot1;
    ko_rotpos_from_tang(p, x2, y2, &lon2, &lat2); ko_rotvec_from_tang(p, lon2, xdot2, ydot2, &uvel2, &vvel2);
  } else { lon2 = lon1 + dt_2 * u1; lat2 = lat1 + dt_2 * v1; uvel2 = uvel1 + dt_2 * ax1; vvel2 = vvel1 + dt_2 * ay1; }
  i = i1; j = j1; xi = bs[KID_B_XI]; yj = bs[KID_B_YJ];
  ko_adjust_index_and_ground(g, p, &lon2, &lat2, &i, &j, &xi, &yj, &bounced, err);
  double dxdl2; ko_meters_to_grid(g, p, lat2, &dxdl2, &dydl);
  double u2 = uvel2 * dxdl2, v2 = vvel2 * dydl;
  double ax2, ay2, xddot2 = 0, yddot2 = 0, xddot2n = 0, yddot2n = 0;
  ko_accel(g, p, bs, nb, i, j, xi, yj, lat2, uvel2, vvel2, uvel1, vvel1, dt_2, &ax2, &ay2, &axn2, &ayn2, &bxn, &byn, tick);
  if (on_tang) { ko_rotvec_to_tang(p, lon2, ax2, ay2, &xddot2, &yddot2); ko_rotvec_to_tang(p, lon2, axn2, ayn2, &xddot2n, &yddot2n); }
  /* stage 3 */
  double lon3, lat3, uvel3, vvel3, x3, y3, xdot3 = 0, ydot3 = 0;
  if (on_tang) {
    x3 = x1 + dt_2 * xdot2; y3 = y1 + dt_2 * ydot2;
    xdot3 = xdot1 + dt_2 * xddot2; ydot3 = ydot1 + dt_2 * yddot2;
    ko_rotpos_from_tang(p, x3, y3, &lon3, &lat3); ko_rotvec_from_tang(p, lon3, xdot3, ydot3, &uvel3, &vvel3);
  } else { lon3 = lon1 + dt_2 * u2; lat3 = lat1 + dt_2 * v2; uvel3 = uvel1 + dt_2 * ax2; vvel3 = vvel1 + dt_2 * ay2; }
  i = i1; j = j1; xi = bs[KID_B_XI]; yj = bs[KID_B_YJ];
  ko_adjust_index_and_ground(g, p, &lon3, &lat3, &i, &j, &xi, &yj, &bounced, err);
  double dxdl3; ko_meters_to_grid(g, p, lat3, &dxdl3, &dydl);
  double u3 = uvel3 * dxdl3, v3 = vvel3 * dydl;
  double ax3, ay3, xddot3 = 0, yddot3 = 0, xddot3n = 0, yddot3n = 0;
  ko_accel(g, p, bs, nb, i, j, xi, yj, lat3, uvel3, vvel3, uvel1, vvel1, dt, &ax3, &ay3, &axn3, &ayn3, &bxn, &byn, tick);
  if (on_tang) { ko_rotvec_to_tang(p, lon3, ax3, ay3, &xddot3, &yddot3); ko_rotvec_to_tang(p, lon3, axn3, ayn3, &xddot3n, &yddot3n); }
  /* stage 4 */
  double lon4, lat4, uvel4, vvel4, x4, y4, xdot4 = 0, ydot4 = 0;
  if (on_tang) {
    x4 = x1 + dt * xdot3; y4 = y1 + dt * ydot3;
    xdot4 = xdot1 + dt * xddot3; ydot4 = ydot1 + dt * yddot3;
    ko_rotpos_from_tang(p, x4, y4, &lon4, &lat4); ko_rotvec_from_tang(p, lon4, xdot4, ydot4, &uvel4, &vvel4);
  } else { lon4 = lon1 + dt * u3; lat4 = lat1 + dt * v3; uvel4 = uvel1 + dt * ax3; vvel4 = vvel1 + dt * ay3; }
  i = i1; j = j1; xi = bs[KID_B_XI]; yj = bs[KID_B_YJ];
  ko_adjust_index_and_ground(g, p, &lon4, &lat4, &i, &j, &xi, &yj, &bounced, err);
  double dxdl4; ko_meters_to_grid(g, p, lat4, &dxdl4, &dydl);
  double u4 = uvel4 * dxdl4, v4 = vvel4 * dydl;
  double ax4, ay4, xddot4 = 0, yddot4 = 0, xddot4n = 0, yddot4n = 0;
  ko_accel(g, p, bs, nb, i, j, xi, yj, lat4, uvel4, vvel4, uvel1, vvel1, dt, &ax4, &ay4, &axn4, &ayn4, &bxn, &byn, tick);
  if (on_tang) { ko_rotvec_to_tang(p, lon4, ax4, ay4, &xddot4, &yddot4); ko_rotvec_to_tang(p, lon4, axn4, ayn4, &xddot4n, &yddot4n); }
  /* combine IB:7597-7616 */
  if (on_tang) {
    double xn = x1 + dt_6 * ((xdot1 + xdot4) + 2. * (xdot2 + xdot3));
    double yn = y1 + dt_6 * ((ydot1 + ydot4) + 2. * (ydot2 + ydot3));
    double xdotn = xdot1 + dt_6 * ((xddot1 + xddot4) + 2. * (xddot2 + xddot3));
    double ydotn = ydot1 + dt_6 * ((yddot1 + yddot4) + 2. * (yddot2 + yddot3));
    double xddotn = ((xddot1n + xddot4n) + 2. * (xddot2n + xddot3n)) / 6.;
    double yddotn = ((yddot1n + yddot4n) + 2. * (yddot2n + yddot3n)) / 6.;
    ko_rotpos_from_tang(p, xn, yn, lonn, latn);
    ko_rotvec_from_tang(p, *lonn, xdotn, ydotn, uveln, vveln);
    ko_rotvec_from_tang(p, *lonn, xddotn, yddotn, &axn, &ayn);
    /* bxn, byn keep the values left by the 4th accel call (the reference does not recompute them here) */
  } else {
    *lonn = bs[KID_B_LON] + dt_6 * ((u1 + u4) + 2. * (u2 + u3));
    *latn = bs[KID_B_LAT] + dt_6 * ((v1 + v4) + 2. * (v2 + v3));
    *uveln = bs[KID_B_UVEL] + dt_6 * ((ax1 + ax4) + 2. * (ax2 + ax3));
    *vveln = bs[KID_B_VVEL] + dt_6 * ((ay1 + ay4) + 2. * (ay2 + ay3));
    axn = ((axn1 + axn4) + 2. * (axn2 + axn3)) / 6.;
    ayn = ((ayn1 + ayn4) + 2. * (ayn2 + ayn3)) / 6.;
    bxn = (((ax1 + ax4) + 2. * (ax2 + ax3)) / 6) - (axn / 2);
    byn = (((ay1 + ay4) + 2. * (ay2 + ay3)) / 6) - (ayn / 2);
  }
  i = i1; j = j1; xi = bs[KID_B_XI]; yj = bs[KID_B_YJ];
  ko_adjust_index_and_ground(g, p, lonn, latn, &i, &j, &xi, &yj, &bounced, err);
  *axn_o = axn; *ayn_o = ayn; *bxn_o = bxn; *byn_o = byn; *io = i; *jo = j; *xio = xi; *yjo = yj;
}

/* IB:7203-7328 verlet_stepping + IB:7684-7764 update_verlet_position (non-interactive: position updated
 * right after the velocity write-back, IB:7157-7169) */
static void verlet_step(const ko_grid *g, const kid_params *p, double bs[], int nb, int *ine, int *jne, int64_t *tick, int *err) {
  const double dt = p->dt, dt_2 = 0.5 * dt;
  double lonn = bs[KID_B_LON], latn = bs[KID_B_LAT];
  double axn = bs[KID_B_AXN], ayn = bs[KID_B_AYN], bxn = bs[KID_B_BXN], byn = bs[KID_B_BYN];
  double uvel1 = bs[KID_B_UVEL], vvel1 = bs[KID_B_VVEL];
  int i = *ine, j = *jne; double xi = bs[KID_B_XI], yj = bs[KID_B_YJ];
  bs[KID_B_UVEL_PREV] = bs[KID_B_UVEL] - dt_2 * bs[KID_B_BXN];
  bs[KID_B_VVEL_PREV] = bs[KID_B_VVEL] - dt_2 * bs[KID_B_BYN];
  double uvel3 = uvel1 + (dt_2 * axn), vvel3 = vvel1 + (dt_2 * ayn);
  double ax1, ay1, uveln, vveln;
  ko_accel(g, p, bs, nb, i, j, xi, yj, latn, uvel1, vvel1, uvel1, vvel1, dt, &ax1, &ay1, &axn, &ayn, &bxn, &byn, tick);
  const int on_tang = (bs[KID_B_LAT] > 89.) && g->d.grid_is_latlon;
  if (on_tang) {
    double xdot3, ydot3, xddot1, yddot1;
    ko_rotvec_to_tang(p, lonn, uvel3, vvel3, &xdot3, &ydot3);
    ko_rotvec_to_tang(p, lonn, ax1, ay1, &xddot1, &yddot1);
    double xdotn = xdot3 + (dt * xddot1), ydotn = ydot3 + (dt * yddot1);
    ko_rotvec_from_tang(p, lonn, xdotn, ydotn, &uveln, &vveln);
  } else { uveln = uvel3 + (dt * ax1); vveln = vvel3 + (dt * ay1); }
  if (p->override_iceberg_velocities) { uveln = p->u_override; vveln = p->v_override; }
  bs[KID_B_AXN] = axn; bs[KID_B_AYN] = ayn; bs[KID_B_BXN] = bxn; bs[KID_B_BYN] = byn;
  bs[KID_B_UVEL] = uveln; bs[KID_B_VVEL] = vveln;
  /* update_verlet_position: note it reads berg%uvel AFTER the write-back above */
  {
    double lon1 = bs[KID_B_LON], lat1 = bs[KID_B_LAT], x1 = 0, y1 = 0;
    const int tang2 = (bs[KID_B_LAT] > 89.) && g->d.grid_is_latlon;
    if (tang2) ko_rotpos_to_tang(p, lon1, lat1, &x1, &y1);
    double dxdl1, dydl; ko_meters_to_grid(g, p, lat1, &dxdl1, &dydl);
    double u1 = bs[KID_B_UVEL], v1 = bs[KID_B_VVEL];
    double uvel2 = u1 + (dt_2 * axn) + (dt_2 * bxn);
    double vvel2 = v1 + (dt_2 * ayn) + (dt_2 * byn);
    double xdot2 = 0, ydot2 = 0;
    if (tang2) ko_rotvec_to_tang(p, lon1, uvel2, vvel2, &xdot2, &ydot2);
    double u2 = uvel2 * dxdl1, v2 = vvel2 * dydl;
    double lo, la;
    if (tang2) { double xn = x1 + (dt * xdot2), yn = y1 + (dt * ydot2); ko_rotpos_from_tang(p, xn, yn, &lo, &la); }
    else { lo = lon1 + (dt * u2); la = lat1 + (dt * v2); }
    int bounced;
    ko_adjust_index_and_ground(g, p, &lo, &la, &i, &j, &xi, &yj, &bounced, err);
    bs[KID_B_LON] = lo; bs[KID_B_LAT] = la; bs[KID_B_XI] = xi; bs[KID_B_YJ] = yj; *ine = i; *jne = j;
  }
}

/* IB:4673-4715 */
void ko_interp_gridded_fields_to_bergs(const ko_grid *g, const kid_params *p, kid_berg_soa *b) {
  double env[13];
  for (int64_t k = 0; k < b->n; ++k) {
    if (!berg_alive(b, k)) continue;
    if (b->f64[KID_B_HALO_BERG] && b->f64[KID_B_HALO_BERG][k] >= 0.5) continue;
    ko_interp_flds(g, p, b->f64[KID_B_LON][k], b->f64[KID_B_LAT][k], b->i32[KID_BI_INE][k], b->i32[KID_BI_JNE][k],
                   b->f64[KID_B_XI][k], b->f64[KID_B_YJ][k], env);
    for (int e = 0; e < 13; ++e) PUT(b, KID_B_UO + e, k, env[e]);
  }
}

/* IB:7081-7200 evolve_icebergs, non-interactive bergs.  Bergs whose cell leaves the computational domain are
 * removed: on a single PE with no neighbour send_bergs_to_other_pes packs and deletes them (FW:3024-3041). */
void ko_evolve_icebergs(const ko_grid *g, const kid_params *p, kid_berg_soa *b, double *scalars) {
  double bs[KID_NB_F64];
  int64_t tickets = 0;
  int nerr = 0;
  for (int64_t k = 0; k < b->n; ++k) {
    if (!berg_alive(b, k)) continue;
    if (b->f64[KID_B_STATIC_BERG] && !(b->f64[KID_B_STATIC_BERG][k] < 0.5)) continue;
    load_berg(b, k, bs);
    int ine = b->i32[KID_BI_INE][k], jne = b->i32[KID_BI_JNE][k];
    int nb = berg_nbonds(b, k), err = 0;
    if (p->Runge_not_Verlet) {
      double axn, ayn, bxn, byn, uveln, vveln, lonn, latn, xi, yj; int i, j;
      rk4_step(g, p, bs, nb, ine, jne, &axn, &ayn, &bxn, &byn, &uveln, &vveln, &lonn, &latn, &i, &j, &xi, &yj, &tickets, &err);
      if (p->override_iceberg_velocities) { uveln = p->u_override; vveln = p->v_override; }
      PUT(b, KID_B_AXN, k, axn); PUT(b, KID_B_AYN, k, ayn); PUT(b, KID_B_BXN, k, bxn); PUT(b, KID_B_BYN, k, byn);
      PUT(b, KID_B_UVEL, k, uveln); PUT(b, KID_B_VVEL, k, vveln); PUT(b, KID_B_LON, k, lonn); PUT(b, KID_B_LAT, k, latn);
      PUT(b, KID_B_XI, k, xi); PUT(b, KID_B_YJ, k, yj);
      b->i32[KID_BI_INE][k] = i; b->i32[KID_BI_JNE][k] = j;
    } else {
      verlet_step(g, p, bs, nb, &ine, &jne, &tickets, &err);
      PUT(b, KID_B_AXN, k, bs[KID_B_AXN]); PUT(b, KID_B_AYN, k, bs[KID_B_AYN]);
      PUT(b, KID_B_BXN, k, bs[KID_B_BXN]); PUT(b, KID_B_BYN, k, bs[KID_B_BYN]);
      PUT(b, KID_B_UVEL, k, bs[KID_B_UVEL]); PUT(b, KID_B_VVEL, k, bs[KID_B_VVEL]);
      PUT(b, KID_B_UVEL_PREV, k, bs[KID_B_UVEL_PREV]); PUT(b, KID_B_VVEL_PREV, k, bs[KID_B_VVEL_PREV]);
      PUT(b, KID_B_LON, k, bs[KID_B_LON]); PUT(b, KID_B_LAT, k, bs[KID_B_LAT]);
      PUT(b, KID_B_XI, k, bs[KID_B_XI]); PUT(b, KID_B_YJ, k, bs[KID_B_YJ]);
      b->i32[KID_BI_INE][k] = ine; b->i32[KID_BI_JNE][k] = jne;
    }
    nerr += err;
    int i = b->i32[KID_BI_INE][k], j = b->i32[KID_BI_JNE][k];
    if (i < g->d.isc || i > g->d.iec || j < g->d.jsc || j > g->d.jec) {
      const int nic = g->d.iec - g->d.isc + 1;
      const int i2 = i > g->d.iec ? i - nic : (i < g->d.isc ? i + nic : i);
      if (p->periodic_reentry && g->d.Lx > 0. && j >= g->d.jsc && j <= g->d.jec && i2 >= g->d.isc && i2 <= g->d.iec) {
        /* the seam as a boundary between two PEs: the berg is sent east/west (FW:3024-3041) and unpacked on the other side
         * with its *_old fields reset (FW:3573-3577), its cell found there (check_and_find_cell, FW:3628: the modulo-aware
         * point-in-cell test accepts the cell one period away) and xi, yj recomputed (FW:3634); lon itself is not changed */
        const double lon = b->f64[KID_B_LON][k], lat = b->f64[KID_B_LAT][k];
        PUT(b, KID_B_UVEL_OLD, k, b->f64[KID_B_UVEL][k]); PUT(b, KID_B_VVEL_OLD, k, b->f64[KID_B_VVEL][k]);
        PUT(b, KID_B_LON_OLD, k, lon); PUT(b, KID_B_LAT_OLD, k, lat);
        double xi, yj; int perr = 0;
        if (!ko_is_point_in_cell(g, lon, lat, i2, j)) nerr += 1;   /* 'can not find a cell to place berg in!' FW:3660 */
        (void)ko_pos_within_cell(g, p, lon, lat, i2, j, &xi, &yj, &perr);
        b->i32[KID_BI_INE][k] = i2;
        PUT(b, KID_B_XI, k, xi); PUT(b, KID_B_YJ, k, yj);
      } else if (b->i32[KID_BI_ALIVE]) b->i32[KID_BI_ALIVE][k] = 0;
    }
  }
  if (scalars) { scalars[KID_S_NSPEEDING_TICKETS] += (double)tickets; scalars[KID_S_ERROR_COUNT] += (double)nerr; }
}

/* ------------------------------------------------------------------------------------------------
 * IB:3307-3364 rolling
 * ---------------------------------------------------------------------------------------------- */
static void swapd(double *x, double *y) { double t = *x; *x = *y; *y = t; }
void ko_rolling(const kid_params *p, double *Tn, double *Wn, double *Ln) {
  const double Delta = 6.0;
  double Dn = (p->rho_bergs / RHO_SEAWATER) * (*Tn);
  if (Dn > 0.) {
    if (!p->use_updated_rolling_scheme && p->tip_parameter < 999.) {
      if (dmax(*Wn, *Ln) < sqrt(0.92 * (Dn * Dn) + 58.32 * Dn)) {
        swapd(Tn, Wn);
        if (*Wn > *Ln) swapd(Wn, Ln);
      }
    } else {
      if (*Wn > *Ln) swapd(Ln, Wn);
      if (!p->use_updated_rolling_scheme && p->tip_parameter >= 999.) {
        double q = p->rho_bergs / RHO_SEAWATER;
        if (*Wn < sqrt((6.0 * q * (1 - q) * ((*Tn) * (*Tn))) - (12 * Delta * q * (*Tn)))) {
          swapd(Tn, Wn);
          if (*Wn > *Ln) swapd(Wn, Ln);
        }
      }
      if (p->use_updated_rolling_scheme) {
        double tip;
        if (p->tip_parameter > 0.) tip = p->tip_parameter;
        else tip = sqrt(6 * (p->rho_bergs / RHO_SEAWATER) * (1 - (p->rho_bergs / RHO_SEAWATER)));
        if ((tip * (*Tn)) > *Wn) {
          swapd(Tn, Wn);
          if (*Wn > *Ln) swapd(Wn, Ln);
        }
      }
    }
  }
}

/* IB:3370-3387 */
void ko_fl_bits_dimensions(const kid_params *p, double thickness, double *L_fl, double *W_fl, double *T_fl) {
  const double l_c = p->pi / (2. * sqrt(2.)), lw_c = 1. / (GRAVITY * RHO_SEAWATER);
  const double B_c = 1. / (12. * (1. - pow(0.3, 2.)));
  double l_w = pow(lw_c * p->fl_youngs * B_c * pow(thickness, 3.), 0.25);
  double l_b = l_c * l_w;
  *L_fl = 3. * l_b; *W_fl = l_b; *T_fl = thickness;
  ko_rolling(p, T_fl, W_fl, L_fl);
}

/* IB:3492-3785 find_basal_melt (+ calculate_TFreeze IB:3794, calculate_density IB:3816) */
static double calc_tfreeze(double S, double pres) {
  const double dTFr_dp = -7.53E-08, dTFr_dS = -0.0573, TFr_S0_P0 = 0.0832;
  return (TFr_S0_P0 + dTFr_dS * S) + dTFr_dp * pres;
}
double ko_find_basal_melt(const kid_grid_desc *gd, const kid_params *p, double dvo, double lat, double salt,
                          double temp, int use_three_eq, double thickness) {
  const double VK = 0.40, ZETA_N = 0.052, RC = 0.20, c2_3 = 2.0 / 3.0;
  const double dR0_dT = -0.038357, dR0_dS = 0.805876, RHO_T0_S0 = 999.910681, Salin_Ice = 0.0;
  const double kd_molec_salt = 8.02e-10, kd_molec_temp = 1.41e-7, kv_molec = 1.95e-6;
  const double Cp_ml = 3974.0, LF = 3.335e5, gamma_t = 0.0, p_atm = 101325;
  const double pi_180 = p->pi / 180.;
  const double density_ice = p->rho_bergs, Rho0 = RHO_SEAWATER, Hml = 10.;
  const double p_int = p_atm + (GRAVITY * thickness * density_ice);
  const double Rhoml = RHO_T0_S0 + dR0_dT * temp + dR0_dS * salt;
  const double I_ZETA_N = 1.0 / ZETA_N, I_LF = 1.0 / LF;
  const double SC = kv_molec / kd_molec_salt, PR = kv_molec / kd_molec_temp, I_VK = 1.0 / VK;
  const double RhoCp = Rho0 * Cp_ml;
  const double Gam_mol_t = 12.5 * pow(PR, c2_3) - 6, Gam_mol_s = 12.5 * pow(SC, c2_3) - 6;
  const double ustar = sqrt(p->cdrag_icebergs * (dvo * dvo + p->utide_icebergs * p->utide_icebergs));
  const double ustar_h = dmax(p->ustar_icebergs_bg, ustar);
  double f_cori;
  if (gd->grid_is_latlon && !p->use_f_plane) f_cori = (2. * p->omega) * sin(pi_180 * lat);
  else f_cori = (2. * p->omega) * sin(pi_180 * p->lat_ref);
  const double absf = fabs(f_cori);
  double hBL_neut;
  if ((absf * Hml <= VK * ustar_h) || (absf == 0.)) hBL_neut = Hml; else hBL_neut = (VK * ustar_h) / absf;
  const double hBL_neut_h_molec = ZETA_N * ((hBL_neut * ustar_h) / (5.0 * kv_molec));
  double ln_neut = 0.0; if (hBL_neut_h_molec > 1.0) ln_neut = log(hBL_neut_h_molec);
  double lprec = 0., t_flux, wT_flux, I_Gam_T = 0., I_Gam_S = 0., Gam_turb, tfreeze;
  int out_of_bounds = 0;
  (void)Gam_mol_s;
  if (use_three_eq) {
    double Sbdry = salt, Sb_max = 0, Sb_min = 0, dS_min = 0, dS_max = 0;
    int Sb_max_set = 0, Sb_min_set = 0;
    const double dB_dS = (GRAVITY / Rhoml) * dR0_dS, dB_dT = (GRAVITY / Rhoml) * dR0_dT;
    for (int it1 = 1; it1 <= 20; ++it1) {
      tfreeze = calc_tfreeze(Sbdry, p_int);
      double dT_ustar = (temp - tfreeze) * ustar_h, dS_ustar = (salt - Sbdry) * ustar_h;
      if (p->const_gamma) { I_Gam_T = p->Gamma_T_3EQ; I_Gam_S = p->Gamma_T_3EQ / 35.; }
      else {
        Gam_turb = I_VK * (ln_neut + (0.5 * I_ZETA_N - 1.0));
        I_Gam_T = 1.0 / (Gam_mol_t + Gam_turb); I_Gam_S = 1.0 / (Gam_mol_s + Gam_turb);
      }
      wT_flux = dT_ustar * I_Gam_T;
      double wB_flux = dB_dS * (dS_ustar * I_Gam_S) + dB_dT * wT_flux;
      if (wB_flux > 0.0) {
        double n_star_term = (ZETA_N / RC) * (hBL_neut * VK) / (ustar_h * ustar_h * ustar_h);
        for (int it3 = 1; it3 <= 30; ++it3) {
          double I_n_star = sqrt(1.0 + n_star_term * wB_flux);
          if (hBL_neut_h_molec > I_n_star * I_n_star)
            Gam_turb = I_VK * ((ln_neut - 2.0 * log(I_n_star)) + (0.5 * I_ZETA_N * I_n_star - 1.0));
          else
            Gam_turb = I_VK * (0.5 * I_ZETA_N * I_n_star - 1.0);
          if (p->const_gamma) { I_Gam_T = p->Gamma_T_3EQ; I_Gam_S = p->Gamma_T_3EQ / 35.; }
          else { I_Gam_T = 1.0 / (Gam_mol_t + Gam_turb); I_Gam_S = 1.0 / (Gam_mol_s + Gam_turb); }
          wT_flux = dT_ustar * I_Gam_T;
          double wB_flux_new = dB_dS * (dS_ustar * I_Gam_S) + dB_dT * wT_flux;
          if (fabs(wB_flux_new - wB_flux) < 1e-4 * (fabs(wB_flux_new) + fabs(wB_flux))) break;
          /* the Newton update of wB_flux_new is never fed back (IB:3694-3697): wB_flux stays fixed */
        }
      }
      t_flux = RhoCp * wT_flux;
      double exch_vel_s = ustar_h * I_Gam_S;
      lprec = I_LF * t_flux; /* both branches IB:3705-3721 */
      double mass_exch = exch_vel_s * Rho0;
      double Sbdry_it = (salt * mass_exch + Salin_Ice * lprec) / (mass_exch + lprec);
      double dS_it = Sbdry_it - Sbdry;
      if (fabs(dS_it) < 1e-4 * (0.5 * (salt + Sbdry + 1.e-10))) break;
      if (dS_it < 0.0) {
        if (Sb_max_set && (Sbdry > Sb_max)) { out_of_bounds = 1; break; }
        Sb_max = Sbdry; dS_max = dS_it; Sb_max_set = 1;
      } else {
        if (Sb_min_set && (Sbdry < Sb_min)) { out_of_bounds = 1; break; }
        Sb_min = Sbdry; dS_min = dS_it; Sb_min_set = 1;
      }
      (void)dS_min; (void)dS_max;
      Sbdry = Sbdry_it; /* IB:3755 overrides the false-position update */
    }
  }
  if (!use_three_eq || out_of_bounds) {
    tfreeze = calc_tfreeze(salt, p_int);
    Gam_turb = I_VK * (ln_neut + (0.5 * I_ZETA_N - 1.0));
    I_Gam_T = 1.0 / (Gam_mol_t + Gam_turb);
    double exch_vel_t = ustar_h * I_Gam_T;
    if (gamma_t > 0.0) exch_vel_t = gamma_t;
    wT_flux = exch_vel_t * (temp - tfreeze);
    t_flux = RhoCp * wT_flux;
    lprec = I_LF * t_flux;
  }
  return lprec / density_ice;
}

/* ------------------------------------------------------------------------------------------------
 * Hexagon geometry IB:4136-4670
 * ---------------------------------------------------------------------------------------------- */
static double area_of_triangle(double Ax, double Ay, double Bx, double By, double Cx, double Cy) {
  return fabs(0.5 * ((Ax * (By - Cy)) + (Bx * (Cy - Ay)) + (Cx * (Ay - By))));
}
static int point_in_interval(double Ax, double Ay, double Bx, double By, double px, double py) {
  if ((px <= dmax(Ax, Bx)) && (px >= dmin(Ax, Bx)))
    if ((py <= dmax(Ay, By)) && (py >= dmin(Ay, By))) return 1;
  return 0;
}
static int point_is_on_the_line(double Ax, double Ay, double Bx, double By, double qx, double qy) {
  double dxc = qx - Ax, dyc = qy - Ay, dxl = Bx - Ax, dyl = By - Ay;
  double cross = dxc * dyl - dyc * dxl;
  return fabs(cross) <= 0.0;
}
int ko_point_in_triangle(double Ax, double Ay, double Bx, double By, double Cx, double Cy, double qx, double qy) {
  if ((Ax == qx && Ay == qy) || (Bx == qx && By == qy) || (Cx == qx && Cy == qy)) return 0;
  if (point_is_on_the_line(Ax, Ay, Bx, By, qx, qy) || point_is_on_the_line(Ax, Ay, Cx, Cy, qx, qy) ||
      point_is_on_the_line(Bx, By, Cx, Cy, qx, qy)) return 0;
  double l0 = (qx - Ax) * (By - Ay) - (qy - Ay) * (Bx - Ax);
  double l1 = (qx - Bx) * (Cy - By) - (qy - By) * (Cx - Bx);
  double l2 = (qx - Cx) * (Ay - Cy) - (qy - Cy) * (Ax - Cx);
  double p0 = fsign1(l0); if (l0 == 0.) p0 = 0.;
  double p1 = fsign1(l1); if (l1 == 0.) p1 = 0.;
  double p2 = fsign1(l2); if (l2 == 0.) p2 = 0.;
  return ((fabs(p0) + fabs(p2)) + (fabs(p1))) == fabs((p0 + p2) + (p1));
}
static void intercept_of_a_line(double Ax, double Ay, double Bx, double By, char axes1, double *x0, double *y0) {
  const double No_intercept_val = 100000000000.;
  *x0 = No_intercept_val; *y0 = No_intercept_val;
  if (axes1 == 'x') { if (Ay != By) { *x0 = Ax - (((Ax - Bx) / (Ay - By)) * Ay); *y0 = 0.; } }
  if (axes1 == 'y') { if (Ax != Bx) { *x0 = 0.; *y0 = -(((Ay - By) / (Ax - Bx)) * Ax) + Ay; } }
}
static void area_of_triangle_across_axes(double Ax, double Ay, double Bx, double By, double Cx, double Cy, char axis1,
                                         double *Area_positive, double *Area_negative) {
  double A_triangle = area_of_triangle(Ax, Ay, Bx, By, Cx, Cy);
  double pABx, pABy, pACx, pACy, A0 = 0.;
  intercept_of_a_line(Ax, Ay, Bx, By, axis1, &pABx, &pABy);
  intercept_of_a_line(Ax, Ay, Cx, Cy, axis1, &pACx, &pACy);
  if (axis1 == 'x') A0 = Ay;
  if (axis1 == 'y') A0 = Ax;
  double A_half = area_of_triangle(Ax, Ay, pABx, pABy, pACx, pACy);
  if (A0 >= 0.) { *Area_positive = A_half; *Area_negative = A_triangle - A_half; }
  else { *Area_positive = A_triangle - A_half; *Area_negative = A_half; }
}
static void dividing_triangle_across_axes(double Ax, double Ay, double Bx, double By, double Cx, double Cy, char axes1,
                                          double *Ap, double *An) {
  double A0 = 0, B0 = 0, C0 = 0;
  if (axes1 == 'x') { A0 = Ay; B0 = By; C0 = Cy; }
  if (axes1 == 'y') { A0 = Ax; B0 = Bx; C0 = Cx; }
  double A_triangle = area_of_triangle(Ax, Ay, Bx, By, Cx, Cy);
  *Ap = 0.; *An = 0.;
  if ((B0 * C0) > 0.) {
    if ((A0 * B0) >= 0.) {
      if ((A0 > 0.) || ((A0 == 0.) && (B0 > 0.))) { *Ap = A_triangle; *An = 0.; }
      else { *Ap = 0.; *An = A_triangle; }
    } else area_of_triangle_across_axes(Ax, Ay, Bx, By, Cx, Cy, axes1, Ap, An);
  } else if ((B0 * C0) < 0.) {
    if ((A0 * B0) >= 0.) area_of_triangle_across_axes(Cx, Cy, Bx, By, Ax, Ay, axes1, Ap, An);
    else area_of_triangle_across_axes(Bx, By, Cx, Cy, Ax, Ay, axes1, Ap, An);
  } else {
    if (((A0 == 0.) && (B0 == 0.)) && (C0 == 0.)) { *Ap = 0.; *An = 0.; }
    else if ((A0 * B0 < 0.) || (A0 * C0 < 0.)) area_of_triangle_across_axes(Ax, Ay, Bx, By, Cx, Cy, axes1, Ap, An);
    else if (((A0 * B0 > 0.) || (A0 * C0 > 0.)) || (((fabs(A0) > 0.) && (B0 == 0.)) && (C0 == 0.))) {
      if (A0 > 0.) { *Ap = A_triangle; *An = 0.; } else { *Ap = 0.; *An = A_triangle; }
    } else if (A0 == 0.) {
      if ((B0 > 0.) || (C0 > 0.)) { *Ap = A_triangle; *An = 0.; }
      else if ((B0 < 0.) || (C0 < 0.)) { *Ap = 0.; *An = A_triangle; }
    }
  }
}
static void triangle_into_four_quadrants(double Ax, double Ay, double Bx, double By, double Cx, double Cy,
                                         double *Area_triangle, double *Q1, double *Q2, double *Q3, double *Q4) {
  double Area_Upper, Area_Lower, Area_Right, Area_Left, px = 0, py = 0, qx = 0, qy = 0, Area_key;
  int Key = 4;
  *Area_triangle = area_of_triangle(Ax, Ay, Bx, By, Cx, Cy);
  dividing_triangle_across_axes(Ax, Ay, Bx, By, Cx, Cy, 'x', &Area_Upper, &Area_Lower);
  dividing_triangle_across_axes(Ax, Ay, Bx, By, Cx, Cy, 'y', &Area_Right, &Area_Left);
  if (ko_point_in_triangle(Ax, Ay, Bx, By, Cx, Cy, 0., 0.)) {
    intercept_of_a_line(Ax, Ay, Bx, By, 'x', &px, &py);
    intercept_of_a_line(Ax, Ay, Bx, By, 'y', &qx, &qy);
    if (!(point_in_interval(Ax, Ay, Bx, By, px, py) && point_in_interval(Ax, Ay, Bx, By, qx, qy))) {
      intercept_of_a_line(Ax, Ay, Cx, Cy, 'x', &px, &py);
      intercept_of_a_line(Ax, Ay, Cx, Cy, 'y', &qx, &qy);
      if (!(point_in_interval(Ax, Ay, Cx, Cy, px, py) && point_in_interval(Ax, Ay, Cx, Cy, qx, qy))) {
        intercept_of_a_line(Bx, By, Cx, Cy, 'x', &px, &py);
        intercept_of_a_line(Bx, By, Cx, Cy, 'y', &qx, &qy);
      }
    }
    Area_key = area_of_triangle(px, py, qx, qy, 0., 0.);
    if ((px >= 0.) && (qy >= 0.)) Key = 1;
    else if ((px < 0.) && (qy >= 0.)) Key = 2;
    else if ((px < 0.) && (qy < 0.)) Key = 3;
    else if ((px >= 0.) && (qy < 0.)) Key = 4;
  } else {
    Area_key = 0;
    if ((!((((Ax > 0.) && (Ay > 0.)) || ((Bx > 0.) && (By > 0.))) || ((Cx > 0.) && (Cy > 0.)))) && ((Area_Upper + Area_Right) <= *Area_triangle)) Key = 1;
    else if ((!((((Ax < 0.) && (Ay > 0)) || ((Bx < 0.) && (By > 0.))) || ((Cx < 0.) && (Cy > 0.)))) && ((Area_Upper + Area_Left) <= *Area_triangle)) Key = 2;
    else if ((!((((Ax < 0.) && (Ay < 0.)) || ((Bx < 0.) && (By < 0.))) || ((Cx < 0.) && (Cy < 0.)))) && ((Area_Lower + Area_Left) <= *Area_triangle)) Key = 3;
    else Key = 4;
  }
  if (Key == 1) { *Q1 = Area_key; *Q2 = Area_Upper - *Q1; *Q4 = Area_Right - *Q1; *Q3 = *Area_triangle - (*Q1 + *Q2 + *Q4); }
  else if (Key == 2) { *Q2 = Area_key; *Q1 = Area_Upper - *Q2; *Q4 = Area_Right - *Q1; *Q3 = *Area_triangle - (*Q1 + *Q2 + *Q4); }
  else if (Key == 3) { *Q3 = Area_key; *Q2 = Area_Left - *Q3; *Q1 = Area_Upper - *Q2; *Q4 = *Area_triangle - (*Q1 + *Q2 + *Q3); }
  else { *Q4 = Area_key; *Q1 = Area_Right - *Q4; *Q2 = Area_Upper - *Q1; *Q3 = *Area_triangle - (*Q1 + *Q2 + *Q4); }
  *Q1 = dmax(*Q1, 0.); *Q2 = dmax(*Q2, 0.); *Q3 = dmax(*Q3, 0.); *Q4 = dmax(*Q4, 0.);
}
static const double KO_PI = 3.14159265358979323846; /* `pi` of constants_mod inside rotate_and_translate */
static void rotate_and_translate(double *px, double *py, double theta, double x0, double y0) {
  double px_temp = (cos(theta * KO_PI / 180) * (*px)) + (sin(theta * KO_PI / 180) * (*py));
  double py_temp = (-sin(theta * KO_PI / 180) * (*px)) + (cos(theta * KO_PI / 180) * (*py));
  *px = px_temp + x0; *py = py_temp + y0;
}
void ko_hexagon_into_quadrants(double x0, double y0, double H, double theta, double *Area_hex,
                               double *Area_Q1, double *Area_Q2, double *Area_Q3, double *Area_Q4) {
  double S = (2 / sqrt(3.)) * H;
  double Cx[6] = {S, H / sqrt(3.), -H / sqrt(3.), -S, -H / sqrt(3.), H / sqrt(3.)};
  double Cy[6] = {0., H, H, 0., -H, -H};
  for (int k = 0; k < 6; ++k) rotate_and_translate(&Cx[k], &Cy[k], theta, x0, y0);
  double TA[6], T1[6], T2[6], T3[6], T4[6];
  for (int k = 0; k < 6; ++k) {
    int k2 = (k + 1) % 6;
    triangle_into_four_quadrants(x0, y0, Cx[k], Cy[k], Cx[k2], Cy[k2], &TA[k], &T1[k], &T2[k], &T3[k], &T4[k]);
  }
  *Area_hex = TA[0] + TA[1] + TA[2] + TA[3] + TA[4] + TA[5];
  double Q1 = T1[0] + T1[1] + T1[2] + T1[3] + T1[4] + T1[5];
  double Q2 = T2[0] + T2[1] + T2[2] + T2[3] + T2[4] + T2[5];
  double Q3 = T3[0] + T3[1] + T3[2] + T3[3] + T3[4] + T3[5];
  double Q4 = T4[0] + T4[1] + T4[2] + T4[3] + T4[4] + T4[5];
  Q1 = dmax(Q1, 0.); Q2 = dmax(Q2, 0.); Q3 = dmax(Q3, 0.); Q4 = dmax(Q4, 0.);
  double Error = *Area_hex - (Q1 + Q2 + Q3 + Q4);
  if (((Q1 >= Q2) && (Q1 >= Q3)) && (Q1 >= Q4)) Q1 = Q1 + Error;
  else if (((Q2 >= Q1) && (Q2 >= Q3)) && (Q2 >= Q4)) Q2 = Q2 + Error;
  else if (((Q3 >= Q1) && (Q3 >= Q2)) && (Q3 >= Q4)) Q3 = Q3 + Error;
  else if (((Q4 >= Q1) && (Q4 >= Q2)) && (Q4 >= Q3)) Q4 = Q4 + Error;
  *Area_Q1 = Q1; *Area_Q2 = Q2; *Area_Q3 = Q3; *Area_Q4 = Q4;
}

/* ------------------------------------------------------------------------------------------------
 * IB:3959-4085: the nine footprint weights, order yDxL,yDxC,yDxR,yCxL,yCxC,yCxR,yUxL,yUxC,yUxR
 * ---------------------------------------------------------------------------------------------- */
/* hexagon orientation of the berg being spread: initial_orientation, or the bond-derived value when the caller has
 * installed a per-berg array (find_orientation_using_iceberg_bonds, IB:4003-4004; oracle/kid_oracle_mts.c) */
static const double *g_orient = NULL;
static double g_orient_now = 0.;
static int g_orient_use = 0;
void ko_set_orientation(const double *per_berg) { g_orient = per_berg; }
void ko_spread_weights(const ko_grid *g, const kid_params *p, int i, int j, double x, double y,
                       double Area, double static_berg, double w[9], double *I_fraction_used) {
  double yDxL = 0., yDxC = 0., yDxR = 0., yCxL = 0., yCxR = 0., yUxL = 0., yUxC = 0., yUxR = 0., yCxC = 1.;
  double fraction_used;
  const double a_ij = GS(g, KID_G_AREA, i, j);
#define M(di, dj) GS(g, KID_G_MSK, i + (di), j + (dj))
  if (!p->hexagonal_icebergs) {
    double L, xL, xR, xC, yD, yU, yC;
    if (a_ij > 0) L = dmin(sqrt(Area / a_ij), 1.0); else L = 1.;
    if (p->use_old_spreading) {
      xL = dmin(0.5, dmax(0., 0.5 - x)); xR = dmin(0.5, dmax(0., x - 0.5)); xC = dmax(0., 1. - (xL + xR));
      yD = dmin(0.5, dmax(0., 0.5 - y)); yU = dmin(0.5, dmax(0., y - 0.5)); yC = dmax(0., 1. - (yD + yU));
    } else {
      xL = dmin(0.5, dmax(0., 0.5 - (x / L))); xR = dmin(0.5, dmax(0., (x / L) + (0.5 - (1 / L)))); xC = dmax(0., 1. - (xL + xR));
      yD = dmin(0.5, dmax(0., 0.5 - (y / L))); yU = dmin(0.5, dmax(0., (y / L) + (0.5 - (1 / L)))); yC = dmax(0., 1. - (yD + yU));
    }
    yDxL = yD * xL * M(-1, -1); yDxC = yD * xC * M(0, -1); yDxR = yD * xR * M(1, -1);
    yCxL = yC * xL * M(-1, 0); yCxR = yC * xR * M(1, 0);
    yUxL = yU * xL * M(-1, 1); yUxC = yU * xC * M(0, 1); yUxR = yU * xR * M(1, 1);
    yCxC = 1. - (((yDxL + yUxR) + (yDxR + yUxL)) + ((yCxL + yCxR) + (yDxC + yUxC)));
    fraction_used = 1.;
  } else {
    double orientation = g_orient_use ? g_orient_now : p->initial_orientation; /* IB:4003-4004 */
    double H, S, origin_x = 1., origin_y = 1., x0, y0, Ah, Q1, Q2, Q3, Q4;
    if (a_ij > 0) H = dmin(((sqrt(Area / (2. * sqrt(3.))) / sqrt(a_ij))), 1.);
    else H = (sqrt(3.) / 2) * (0.49);
    S = (2 / sqrt(3.)) * H; (void)S;
    if (x < 0.5) origin_x = 0.;
    if (y < 0.5) origin_y = 0.;
    x0 = (x - origin_x); y0 = (y - origin_y);
    ko_hexagon_into_quadrants(x0, y0, H, orientation, &Ah, &Q1, &Q2, &Q3, &Q4);
    Q1 = Q1 / Ah; Q2 = Q2 / Ah; Q3 = Q3 / Ah; Q4 = Q4 / Ah;
    if ((x >= 0.5) && (y >= 0.5)) { yUxR = Q1; yUxC = Q2; yCxC = Q3; yCxR = Q4; }
    else if ((x < 0.5) && (y >= 0.5)) { yUxC = Q1; yUxL = Q2; yCxL = Q3; yCxC = Q4; }
    else if ((x < 0.5) && (y < 0.5)) { yCxC = Q1; yCxL = Q2; yDxL = Q3; yDxC = Q4; }
    else if ((x >= 0.5) && (y < 0.5)) { yCxR = Q1; yCxC = Q2; yDxC = Q3; yDxR = Q4; }
    fraction_used = ((yDxL * M(-1, -1)) + (yDxC * M(0, -1)) + (yDxR * M(1, -1)) + (yCxL * M(-1, 0)) + (yCxR * M(1, 0))
                     + (yUxL * M(-1, 1)) + (yUxC * M(0, 1)) + (yUxR * M(1, 1)) + (pow(yCxC, M(0, 0)))); /* IB:4081 `**` */
    if (static_berg == 1) fraction_used = 1.;
  }
#undef M
  *I_fraction_used = 1. / fraction_used;
  w[0] = yDxL; w[1] = yDxC; w[2] = yDxR; w[3] = yCxL; w[4] = yCxC; w[5] = yCxR; w[6] = yUxL; w[7] = yUxC; w[8] = yUxR;
}

/* IB:3895-4100 spread_mass_across_ocean_cells (+ IB:4103-4133) */
static void spread_mass(const ko_grid *g, const kid_params *p, double *acc, const double bs[], int i, int j,
                        double x, double y, double Mberg, double Mbits, double scaling, double Area, double Tn, int addfootloose) {
  const double rho_sw = 1035.; /* IB:3919: local parameter shadows the module value */
  const size_t ncell = (size_t)NI(g) * (size_t)(g->d.jed - g->d.jsd + 1);
  double Mass_berg = Mberg, Mfl, Mbits_fl;
  if (addfootloose) { Mfl = bs[KID_B_MASS_OF_FL_BITS]; Mbits_fl = bs[KID_B_MASS_OF_FL_BERGY_BITS]; } else { Mfl = 0.; Mbits_fl = 0.; }
  if (p->grounding_fraction > 0.) {
    double Hocean = p->grounding_fraction * (GS(g, KID_G_OCEAN_DEPTH, i, j) + GF(g, KID_F_SSH, i, j));
    double Dn = (p->rho_bergs / rho_sw) * Tn;
    if (Dn > Hocean) Mass_berg = Mass_berg * dmin(1., Hocean / Dn);
    if (Mfl > 0. && addfootloose) {
      double Lfl, Wfl, Tfl; ko_fl_bits_dimensions(p, bs[KID_B_THICKNESS], &Lfl, &Wfl, &Tfl);
      Dn = (p->rho_bergs / rho_sw) * Tfl;
      if (Dn > Hocean) Mfl = Mfl * dmin(1., Hocean / Dn);
    }
  }
  Mass_berg = Mass_berg + Mfl;
  double Mass = (Mass_berg + Mbits + Mbits_fl) * scaling;
  if (p->clipping_depth > 0.) Mass = dmin(Mass, p->clipping_depth * GS(g, KID_G_AREA, i, j) * rho_sw);
  double w[9], Ifu;
  ko_spread_weights(g, p, i, j, x, y, Area, bs[KID_B_STATIC_BERG], w, &Ifu);
  const size_t c = GIDX(g, i, j);
  const double vars[4] = {Mass, Area * scaling, bs[KID_B_UVEL] * Area * scaling, bs[KID_B_VVEL] * Area * scaling};
  const int base[4] = {KID_A_MASS_ON_OCEAN, KID_A_AREA_ON_OCEAN, KID_A_UVEL_ON_OCEAN, KID_A_VVEL_ON_OCEAN};
  for (int v = 0; v < 4; ++v)
    for (int s = 0; s < 9; ++s)
      acc[(size_t)(base[v] + s) * ncell + c] = acc[(size_t)(base[v] + s) * ncell + c] + (w[s] * vars[v] * Ifu);
}

/* ------------------------------------------------------------------------------------------------
 * order of traversal = cells j-outer/i-inner (IB:7106), inside a cell `inorder` (FW:4318-4359)
 * ---------------------------------------------------------------------------------------------- */
static const kid_berg_soa *g_sort_b;
static int cmp_ref_order(const void *pa, const void *pb) {
  int64_t a = *(const int64_t *)pa, b = *(const int64_t *)pb;
  const kid_berg_soa *s = g_sort_b;
  int ja = s->i32[KID_BI_JNE][a], jb = s->i32[KID_BI_JNE][b];
  if (ja != jb) return ja < jb ? -1 : 1;
  int ia = s->i32[KID_BI_INE][a], ib = s->i32[KID_BI_INE][b];
  if (ia != ib) return ia < ib ? -1 : 1;
  int ya = s->i32[KID_BI_START_YEAR] ? s->i32[KID_BI_START_YEAR][a] : 0, yb = s->i32[KID_BI_START_YEAR] ? s->i32[KID_BI_START_YEAR][b] : 0;
  if (ya != yb) return ya < yb ? -1 : 1;
  static const int keys[4] = {KID_B_START_DAY, KID_B_START_MASS, KID_B_START_LON, KID_B_START_LAT};
  for (int k = 0; k < 4; ++k) {
    if (!s->f64[keys[k]]) continue;
    double va = s->f64[keys[k]][a], vb = s->f64[keys[k]][b];
    if (va < vb) return -1;
    if (va > vb) return 1;
  }
  return a < b ? -1 : (a > b);
}
void ko_reference_order(const kid_berg_soa *b, int64_t *perm) {
  for (int64_t k = 0; k < b->n; ++k) perm[k] = k;
  g_sort_b = b;
  qsort(perm, (size_t)b->n, sizeof(int64_t), cmp_ref_order);
}

/* ------------------------------------------------------------------------------------------------
 * IB:2844-3300 thermodynamics
 * ---------------------------------------------------------------------------------------------- */
static int minloc_abs(const double *tab, double v) { /* Fortran minloc(abs(tab-v),1), 1-based -> 0-based */
  int k = 0; double best = fabs(tab[0] - v);
  for (int q = 1; q < 10; ++q) { double d = fabs(tab[q] - v); if (d < best) { best = d; k = q; } }
  return k;
}
void ko_thermodynamics(const ko_grid *g, const kid_params *p, kid_berg_soa *b, double *acc, double *scalars) {
  const double perday = 1. / 86400.;
  const double l_c = p->pi / (2. * sqrt(2.)), lw_c = 1. / (GRAVITY * RHO_SEAWATER), B_c = 1. / (12. * (1. - pow(0.3, 2.)));
  const size_t ncell = (size_t)NI(g) * (size_t)(g->d.jed - g->d.jsd + 1);
  const double dt = p->dt;
  int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(b->n > 0 ? b->n : 1));
  ko_reference_order(b, perm);
  /* IB:2872-2873 */
  for (int s = 0; s < 9; ++s) {
    memset(acc + (size_t)(KID_A_UVEL_ON_OCEAN + s) * ncell, 0, ncell * sizeof(double));
    memset(acc + (size_t)(KID_A_VVEL_ON_OCEAN + s) * ncell, 0, ncell * sizeof(double));
  }
  double N_max = p->hexagonal_icebergs ? 6.0 : 4.0;
  double bs[KID_NB_F64], env[13];
  for (int64_t kk = 0; kk < b->n; ++kk) {
    const int64_t k = perm[kk];
    if (!berg_alive(b, k)) continue;
    load_berg(b, k, bs);
    const int i = b->i32[KID_BI_INE][k], j = b->i32[KID_BI_JNE][k];
    if (p->old_interp_flds_order || (!p->mts && !p->dem && bs[KID_B_HALO_BERG] >= 0.5)) {
      ko_interp_flds(g, p, bs[KID_B_LON], bs[KID_B_LAT], i, j, bs[KID_B_XI], bs[KID_B_YJ], env);
      for (int e = 0; e < 12; ++e) { bs[KID_B_UO + e] = env[e]; PUT(b, KID_B_UO + e, k, env[e]); } /* od not passed IB:2891 */
    }
    double SST = bs[KID_B_SST], SSS = bs[KID_B_SSS];
    double IC = dmin(1., bs[KID_B_CN] + p->sicn_shift);
    const double M = bs[KID_B_MASS], T = bs[KID_B_THICKNESS], W = bs[KID_B_WIDTH], L = bs[KID_B_LENGTH];
    const double Vol = T * W * L;
    double du = bs[KID_B_UVEL] - bs[KID_B_UO], dv = bs[KID_B_VVEL] - bs[KID_B_VO];
    const double dvo = sqrt(du * du + dv * dv);
    du = bs[KID_B_UA] - bs[KID_B_UO]; dv = bs[KID_B_VA] - bs[KID_B_VO];
    const double dva = sqrt(du * du + dv * dv);
    const double Ss = 1.5 * pow(dva, 0.5) + 0.1 * dva;
    double Mv = dmax(7.62e-3 * SST + 1.29e-3 * (SST * SST), 0.) * perday;
    double Mb = dmax(0.58 * pow(dvo, 0.8) * (SST + 4.0) / pow(L, 0.2), 0.) * perday;
    double Me = dmax(1. / 12. * (SST + 2.) * Ss * (1 + cos(p->pi * (IC * IC * IC))), 0.) * perday;
    double Mv_fl = 0., Me_fl = 0.;
    if (bs[KID_B_MASS_OF_FL_BITS] > 0.) { Mv_fl = Mv; Me_fl = Me; }
    double N_bonds = 0.;
    if (p->use_mixed_melting || p->allow_bergs_to_roll) {
      N_bonds = 0.;
      if (p->iceberg_bonds_on) N_bonds = (double)berg_nbonds(b, k);
      if (bs[KID_B_STATIC_BERG] == 1) N_bonds = N_max;
    }
    if (p->melt_icebergs_as_ice_shelf || p->use_mixed_melting) {
      if (!p->use_mixed_layer_salinity_for_thermo) SSS = 35.0;
      double Ms = ko_find_basal_melt(&g->d, p, dvo, bs[KID_B_LAT], SSS, SST, p->Use_three_equation_model, T);
      Ms = dmax(Ms, 0.);
      if ((p->melt_cutoff >= 0.) && p->apply_thickness_cutoff_to_bergs_melt) {
        double Dn = (p->rho_bergs / RHO_SEAWATER) * bs[KID_B_THICKNESS];
        if ((GS(g, KID_G_OCEAN_DEPTH, i, j) - Dn) < p->melt_cutoff) Ms = 0.;
      }
      if (p->use_mixed_melting) {
        Me = ((N_max - N_bonds) / N_max) * (Mv + Me);
        Mv = 0.0;
        Mb = (((N_max - N_bonds) / N_max) * (Mb)) + (N_bonds / N_max) * Ms;
      } else { Mv = 0.0; Me = 0.0; Mb = Ms; }
    }
    if (p->set_melt_rates_to_zero) { Mv = 0.0; Mb = 0.0; Me = 0.0; }
    double Tn, nVol, Mnew, Mnew1 = 0, Mnew2 = 0, dMb, dMv, dMe, dM, Ln1 = 0, Wn1 = 0, Ln, Wn;
    if (p->use_operator_splitting) {
      Tn = dmax(T - Mb * dt, 0.);
      nVol = Tn * W * L; Mnew1 = (nVol / Vol) * M; dMb = M - Mnew1;
      Ln1 = dmax(L - Mv * dt, 0.); Wn1 = dmax(W - Mv * dt, 0.);
      nVol = Tn * Wn1 * Ln1; Mnew2 = (nVol / Vol) * M; dMv = Mnew1 - Mnew2;
      Ln = dmax(Ln1 - Me * dt, 0.); Wn = dmax(Wn1 - Me * dt, 0.);
      nVol = Tn * Wn * Ln; Mnew = (nVol / Vol) * M; dMe = Mnew2 - Mnew;
      dM = M - Mnew;
    } else {
      Ln = dmax(L - (Mv + Me) * (dt), 0.); Wn = dmax(W - (Mv + Me) * (dt), 0.); Tn = dmax(T - Mb * (dt), 0.);
      nVol = Tn * Wn * Ln; Mnew = (nVol / Vol) * M; dM = M - Mnew;
      dMb = (M / Vol) * (W * L) * Mb * dt;
      dMe = (M / Vol) * (T * (W + L)) * Me * dt;
      dMv = (M / Vol) * (T * (W + L)) * Mv * dt;
    }
    double fl_k = bs[KID_B_FL_K];
    if (p->footloose) {
      if (fl_k >= 0) {
        double l_b3 = 3. * l_c * pow(lw_c * p->fl_youngs * B_c * pow(Tn, 3.), 0.25);
        if (L > l_b3) {
          double fb = Tn * (1. - p->rho_bergs / RHO_SEAWATER);
          double kd = Tn - fb;
          if (W > l_b3) {
            fl_k = fl_k + (dMe / fb - dMv / kd) / p->rho_bergs;
            if (fl_k < 0) fl_k = 0;
          } else {
            double dMv_l = dMv * (Wn1 + W) / (2. * (Ln1 + W));
            double dMe_l = dMe * (Wn + Wn1) / (2. * (Ln + Wn1));
            fl_k = fl_k + (dMe_l / fb - dMv_l / kd) / p->rho_bergs;
            if (fl_k < 0) fl_k = 0;
          }
        }
      }
    }
    /* FL bits IB:3031-3068 */
    double Lfl = 0, Wfl = 0, Tfl = 0, Mfl, Volfl, Mb_fl, Tnfl = 0, Lnfl = 0, Wnfl = 0, nVolfl, Mnew_fl, dMfl, dMb_fl, dMv_fl, dMe_fl;
    if (bs[KID_B_MASS_OF_FL_BITS] > 0.) {
      ko_fl_bits_dimensions(p, bs[KID_B_THICKNESS], &Lfl, &Wfl, &Tfl);
      Mfl = bs[KID_B_MASS_OF_FL_BITS];
      Volfl = Lfl * Wfl * Tfl;
      Mb_fl = dmax(0.58 * pow(dvo, 0.8) * (SST + 4.0) / pow(Lfl, 0.2), 0.) * perday;
      Tnfl = dmax(Tfl - Mb_fl * dt, 0.);
      if (p->use_operator_splitting) {
        nVolfl = Tnfl * Wfl * Lfl; double Mnew1_fl = (nVolfl / Volfl) * Mfl; dMb_fl = Mfl - Mnew1_fl;
        Lnfl = dmax(Lfl - Mv_fl * dt, 0.); Wnfl = dmax(Wfl - Mv_fl * dt, 0.);
        nVolfl = Tnfl * Wnfl * Lnfl; double Mnew2_fl = (nVolfl / Volfl) * Mfl; dMv_fl = Mnew1_fl - Mnew2_fl;
        Lnfl = dmax(Lnfl - Me_fl * dt, 0.); Wnfl = dmax(Wnfl - Me_fl * dt, 0.);
        nVolfl = Tnfl * Wnfl * Lnfl; Mnew_fl = (nVolfl / Volfl) * Mfl; dMe_fl = Mnew2_fl - Mnew_fl;
      } else {
        Lnfl = dmax(Lfl - (Mv_fl + Me_fl) * dt, 0.); Wnfl = dmax(Wfl - (Mv_fl + Me_fl) * dt, 0.);
        nVolfl = Tnfl * Wnfl * Lnfl; Mnew_fl = (nVolfl / Volfl) * Mfl;
        dMb_fl = (Mfl / Volfl) * (Wfl * Lfl) * Mb_fl * dt;
        dMe_fl = (Mfl / Volfl) * (Tfl * (Wfl + Lfl)) * Me_fl * dt;
        dMv_fl = (Mfl / Volfl) * (Tfl * (Wfl + Lfl)) * Mv_fl * dt;
      }
      dMfl = Mfl - Mnew_fl;
    } else { dMfl = 0.; dMb_fl = 0.; dMv_fl = 0.; dMe_fl = 0.; Mnew_fl = bs[KID_B_MASS_OF_FL_BITS]; }
    /* bergy bits IB:3071-3111 */
    double dMbitsE, dMbitsM, nMbits, dMbitsE_fl, dMbitsM_fl, nMbits_fl;
    if (p->bergy_bit_erosion_fraction > 0.) {
      double Mbits = bs[KID_B_MASS_OF_BITS];
      dMbitsE = p->bergy_bit_erosion_fraction * dMe;
      nMbits = Mbits + dMbitsE;
      double Lbits = dmin(dmin(dmin(L, W), T), 40.);
      double Abits = (Mbits / p->rho_bergs) / Lbits;
      double Mbb = dmax(0.58 * pow(dvo, 0.8) * (SST + 2.0) / pow(Lbits, 0.2), 0.) * perday;
      Mbb = p->rho_bergs * Abits * Mbb;
      dMbitsM = dmin(Mbb * dt, nMbits);
      nMbits = nMbits - dMbitsM;
      if (Mnew == 0.) { dMbitsM = dMbitsM + nMbits; nMbits = 0.; }
      if (bs[KID_B_MASS_OF_FL_BITS] > 0.) {
        double Mbits_fl = bs[KID_B_MASS_OF_FL_BERGY_BITS];
        dMbitsE_fl = p->bergy_bit_erosion_fraction * dMe_fl;
        nMbits_fl = Mbits_fl + dMbitsE_fl;
        double Lbits_fl = dmin(dmin(dmin(Lfl, Wfl), Tfl), 40.);
        double Abits_fl = (Mbits_fl / p->rho_bergs) / Lbits_fl;
        double Mbb_fl = dmax(0.58 * pow(dvo, 0.8) * (SST + 2.0) / pow(Lbits_fl, 0.2), 0.) * perday;
        Mbb_fl = p->rho_bergs * Abits_fl * Mbb_fl;
        dMbitsM_fl = dmin(Mbb_fl * dt, nMbits_fl);
        nMbits_fl = nMbits_fl - dMbitsM_fl;
        if (Mnew_fl == 0.) { dMbitsM_fl = dMbitsM_fl + nMbits_fl; nMbits_fl = 0.; }
      } else { dMbitsE_fl = 0.; dMbitsM_fl = 0.; nMbits_fl = 0.; }
    } else {
      dMbitsE = 0.; dMbitsM = 0.; nMbits = bs[KID_B_MASS_OF_BITS];
      dMbitsE_fl = 0.; dMbitsM_fl = 0.; nMbits_fl = bs[KID_B_MASS_OF_FL_BERGY_BITS];
    }
    /* grid accumulation IB:3114-3208 */
    const size_t c = GIDX(g, i, j);
    const double area = GS(g, KID_G_AREA, i, j), ms = bs[KID_B_MASS_SCALING];
#define ACC(F, v) acc[(size_t)(F) * ncell + c] = acc[(size_t)(F) * ncell + c] + (v)
    if (area != 0.) {
      double melt = (dM - (dMbitsE - dMbitsM) + dMfl - (dMbitsE_fl - dMbitsM_fl)) / dt;
      ACC(KID_A_FLOATING_MELT, melt / area * ms);
      if (p->diag_mask & KID_DIAG_MELT_BY_CLASS) {
        int kc = (bs[KID_B_LAT] < 0.) ? minloc_abs(p->initial_mass_s, bs[KID_B_START_MASS]) : minloc_abs(p->initial_mass_n, bs[KID_B_START_MASS]);
        ACC(KID_A_MELT_BY_CLASS + kc, melt / area * ms);
      }
      melt = melt * bs[KID_B_HEAT_DENSITY];
      ACC(KID_A_CALVING_HFLX, melt / area * ms);
      scalars[KID_S_NET_HEAT_TO_OCEAN] = scalars[KID_S_NET_HEAT_TO_OCEAN] + melt * ms * dt;
      melt = dM / dt; ACC(KID_A_BERG_MELT, melt / area * ms);
      melt = (dMbitsE + dMbitsE_fl) / dt; ACC(KID_A_BERGY_SRC, melt / area * ms);
      melt = (dMbitsM + dMbitsM_fl) / dt; ACC(KID_A_BERGY_MELT, melt / area * ms);
      melt = dMfl / dt; ACC(KID_A_FL_BITS_MELT, melt / area * ms);
      if (fl_k >= 0) { /* this%fl_k as updated at IB:3018-3024; its sign cannot change there (floor at 0) */
        if (p->diag_mask & KID_DIAG_FL_PARENT_MELT) { melt = (dM - (dMbitsE - dMbitsM)) / dt; ACC(KID_A_FL_PARENT_MELT, melt / area * ms); }
        if (p->diag_mask & KID_DIAG_FL_CHILD_MELT) { melt = (dMfl - (dMbitsE_fl - dMbitsM_fl)) / dt; ACC(KID_A_FL_CHILD_MELT, melt / area * ms); }
        if (p->diag_mask & KID_DIAG_MELT_BUOY) { melt = dMb / dt; ACC(KID_A_MELT_BUOY, melt / area * ms); }
        if (p->diag_mask & KID_DIAG_MELT_EROS) { melt = dMe / dt; ACC(KID_A_MELT_EROS, melt / area * ms); }
        if (p->diag_mask & KID_DIAG_MELT_CONV) { melt = dMv / dt; ACC(KID_A_MELT_CONV, melt / area * ms); }
        if (dMfl > 0) {
          if (p->diag_mask & KID_DIAG_MELT_BUOY_FL) { melt = dMb_fl / dt; ACC(KID_A_MELT_BUOY_FL, melt / area * ms); }
          if (p->diag_mask & KID_DIAG_MELT_EROS_FL) { melt = dMe_fl / dt; ACC(KID_A_MELT_EROS_FL, melt / area * ms); }
          if (p->diag_mask & KID_DIAG_MELT_CONV_FL) { melt = dMv_fl / dt; ACC(KID_A_MELT_CONV_FL, melt / area * ms); }
        }
      } else {
        if (p->diag_mask & KID_DIAG_FL_CHILD_MELT) { melt = (dM - (dMbitsE - dMbitsM)) / dt; ACC(KID_A_FL_CHILD_MELT, melt / area * ms); }
        if (p->diag_mask & KID_DIAG_MELT_BUOY_FL) { melt = dMb / dt; ACC(KID_A_MELT_BUOY_FL, melt / area * ms); }
        if (p->diag_mask & KID_DIAG_MELT_EROS_FL) { melt = dMe / dt; ACC(KID_A_MELT_EROS_FL, melt / area * ms); }
        if (p->diag_mask & KID_DIAG_MELT_CONV_FL) { melt = dMv / dt; ACC(KID_A_MELT_CONV_FL, melt / area * ms); }
      }
    } else {
      scalars[KID_S_ERROR_COUNT] += 1.; /* FATAL 'berg appears to have grounded!' IB:3207 */
    }
    if (p->allow_bergs_to_roll && N_bonds == 0.) ko_rolling(p, &Tn, &Wn, &Ln);
    if (p->Iceberg_melt_without_decay) {
      /* IB:3214-3257: state is left unchanged; with find_melt_using_spread_mass the would-be masses are spread first (IB:3219-3238) */
      if (p->find_melt_using_spread_mass && (Mnew > 0. || Mnew_fl > 0.) && area > 0.) {
        double tb[KID_NB_F64];
        memcpy(tb, bs, sizeof(tb));
        tb[KID_B_MASS_OF_FL_BITS] = Mnew_fl; tb[KID_B_MASS_OF_FL_BERGY_BITS] = nMbits_fl;
        g_orient_use = (g_orient != NULL);
        if (g_orient) g_orient_now = g_orient[k];
        if (Mnew > 0.) spread_mass(g, p, acc, tb, i, j, bs[KID_B_XI], bs[KID_B_YJ], Mnew, nMbits, ms, Ln * Wn, Tn, 1);
        else {
          const double M_edit = Lnfl * Wnfl * Tnfl * p->rho_bergs, Mscale_edit = Mnew_fl * ms / M_edit;
          spread_mass(g, p, acc, tb, i, j, bs[KID_B_XI], bs[KID_B_YJ], M_edit, nMbits, Mscale_edit, Lnfl * Wnfl, Tnfl, 0);
        }
        g_orient_use = 0;
      }
      Mnew = bs[KID_B_MASS]; nMbits = bs[KID_B_MASS_OF_BITS];
      Mnew_fl = bs[KID_B_MASS_OF_FL_BITS]; nMbits_fl = bs[KID_B_MASS_OF_FL_BERGY_BITS];
      PUT(b, KID_B_FL_K, k, fl_k);
    } else {
      PUT(b, KID_B_MASS, k, Mnew); PUT(b, KID_B_MASS_OF_BITS, k, nMbits);
      PUT(b, KID_B_MASS_OF_FL_BITS, k, Mnew_fl); PUT(b, KID_B_MASS_OF_FL_BERGY_BITS, k, nMbits_fl);
      PUT(b, KID_B_THICKNESS, k, Tn); PUT(b, KID_B_WIDTH, k, dmin(Wn, Ln)); PUT(b, KID_B_LENGTH, k, dmax(Wn, Ln));
      PUT(b, KID_B_FL_K, k, fl_k);
    }
    if (Mnew <= 0.) {
      if (Mnew_fl > 0) {
        scalars[KID_S_NBERGS_CALVED_FL] += 1.;
        double mass = Lnfl * Wnfl * Tnfl * p->rho_bergs;
        PUT(b, KID_B_MASS, k, mass);
        PUT(b, KID_B_LENGTH, k, Lnfl); PUT(b, KID_B_WIDTH, k, Wnfl); PUT(b, KID_B_THICKNESS, k, Tnfl);
        nMbits_fl = nMbits_fl * ms;
        double new_ms = Mnew_fl * ms / mass;
        PUT(b, KID_B_MASS_SCALING, k, new_ms);
        PUT(b, KID_B_MASS_OF_BITS, k, nMbits_fl / new_ms);
        PUT(b, KID_B_MASS_OF_FL_BITS, k, 0.); PUT(b, KID_B_MASS_OF_FL_BERGY_BITS, k, 0.);
        PUT(b, KID_B_FL_K, k, -1.);
        if (b->i32[KID_BI_START_YEAR]) b->i32[KID_BI_START_YEAR][k] = p->current_year;
        PUT(b, KID_B_START_DAY, k, p->current_yearday);
        if (area != 0.) ACC(KID_A_FL_BITS_SRC, -(mass * new_ms / (dt * area)));
      } else {
        if (b->i32[KID_BI_ALIVE]) b->i32[KID_BI_ALIVE][k] = 0;
      }
      scalars[KID_S_NBERGS_MELTED] += 1.;
    }
#undef ACC
  }
  free(perm);
}

/* ------------------------------------------------------------------------------------------------
 * IB:3390-3489 create_gridded_icebergs_fields = calculate_mass_on_ocean (IB:4970-5011, with
 * calculate_sum_over_bergs_diagnositcs IB:5014-5071) + sum_up_spread_fields (IB:6077-6150) + ustar
 * ---------------------------------------------------------------------------------------------- */
static int g_wrap_x = 0;   /* periodic_reentry: mpp_update_domains(var_on_ocean) of IB:6103 on a rank that owns the whole zonal period */
static void sum_up_spread_field(const ko_grid *g, const double *acc, int base, int is_area, double *field) {
  const size_t ncell = (size_t)NI(g) * (size_t)(g->d.jed - g->d.jsd + 1);
  const int nic_w = g->d.iec - g->d.isc + 1;
  memset(field, 0, ncell * sizeof(double));
#define WI(i) ((g_wrap_x && (i) < g->d.isc) ? (i) + nic_w : ((g_wrap_x && (i) > g->d.iec) ? (i) - nic_w : (i)))
#define V(i, j, s) acc[(size_t)(base + (s) - 1) * ncell + GIDX(g, WI(i), j)]
  for (int j = g->d.jsc; j <= g->d.jec; ++j)
    for (int i = g->d.isc; i <= g->d.iec; ++i) {
      double dmda = V(i, j, 5) + (((V(i - 1, j - 1, 9) + V(i + 1, j + 1, 1)) + (V(i + 1, j - 1, 7) + V(i - 1, j + 1, 3)))
                                  + ((V(i - 1, j, 6) + V(i + 1, j, 4)) + (V(i, j - 1, 8) + V(i, j + 1, 2))));
      double a = GS(g, KID_G_AREA, i, j);
      if (a > 0) dmda = dmda / a * GS(g, KID_G_MSK, i, j);
      if (is_area) dmda = dmin(dmda, 1.0);
      field[GIDX(g, i, j)] = dmda;
    }
#undef V
}
/* calculate_mass_on_ocean (IB:4970-5011): the per-berg scatter half of create_gridded_icebergs_fields */
static int g_mass_only = 0;   /* calculate_mass_on_ocean(with_diagnostics=.false.), IB:5490 */
void ko_calculate_mass_on_ocean(const ko_grid *g, const kid_params *p, kid_berg_soa *b, double *acc) {
  const size_t ncell = (size_t)NI(g) * (size_t)(g->d.jed - g->d.jsd + 1);
  for (int s = 0; s < 36; ++s) memset(acc + (size_t)(KID_A_MASS_ON_OCEAN + s) * ncell, 0, ncell * sizeof(double));
  int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(b->n > 0 ? b->n : 1));
  ko_reference_order(b, perm);
  double bs[KID_NB_F64];
  const int dm = p->diag_mask;
  for (int64_t kk = 0; kk < b->n; ++kk) {
    const int64_t k = perm[kk];
    if (!berg_alive(b, k)) continue;
    load_berg(b, k, bs);
    const int i = b->i32[KID_BI_INE][k], j = b->i32[KID_BI_JNE][k];
    const double area = GS(g, KID_G_AREA, i, j);
    if (!(area > 0.)) continue;
    g_orient_use = (g_orient != NULL);
    if (g_orient) g_orient_now = g_orient[k];
    /* time_average_weight: the stage spreading of IB:7264/7395-7620 is zeroed again at IB:4984-4987 and never refilled -> nothing to spread */
    if ((p->add_weight_to_ocean && !p->time_average_weight) || p->find_melt_using_spread_mass)
      spread_mass(g, p, acc, bs, i, j, bs[KID_B_XI], bs[KID_B_YJ], bs[KID_B_MASS], bs[KID_B_MASS_OF_BITS], bs[KID_B_MASS_SCALING],
                  bs[KID_B_LENGTH] * bs[KID_B_WIDTH], bs[KID_B_THICKNESS], 1);
    g_orient_use = 0;
    if (g_mass_only) continue;
    const size_t c = GIDX(g, i, j);
    const double ms = bs[KID_B_MASS_SCALING];
#define ACC(F, v) acc[(size_t)(F) * ncell + c] = acc[(size_t)(F) * ncell + c] + (v)
    if (dm & KID_DIAG_VIRTUAL_AREA) {
      double Abits, Abits_fl, Abits_fl_bergy;
      if (p->bergy_bit_erosion_fraction > 0.) {
        double Lbits = dmin(dmin(dmin(bs[KID_B_LENGTH], bs[KID_B_WIDTH]), bs[KID_B_THICKNESS]), 40.);
        Abits = (bs[KID_B_MASS_OF_BITS] / p->rho_bergs) / Lbits;
      } else Abits = 0.0;
      if (p->fl_style == KID_FL_STYLE_FL_BITS) {
        double L_fl, W_fl, T_fl; ko_fl_bits_dimensions(p, bs[KID_B_THICKNESS], &L_fl, &W_fl, &T_fl);
        Abits_fl = (bs[KID_B_MASS_OF_FL_BITS] / p->rho_bergs) / T_fl;
        if (p->bergy_bit_erosion_fraction > 0.) {
          double Lbits = dmin(dmin(dmin(L_fl, W_fl), T_fl), 40.);
          Abits_fl_bergy = (bs[KID_B_MASS_OF_FL_BERGY_BITS] / p->rho_bergs) / Lbits;
        } else Abits_fl_bergy = 0.0;
      } else { Abits_fl = 0.0; Abits_fl_bergy = 0.0; }
      ACC(KID_A_VIRTUAL_AREA, (bs[KID_B_WIDTH] * bs[KID_B_LENGTH] + Abits + Abits_fl + Abits_fl_bergy) * ms);
    }
    if ((dm & KID_DIAG_MASS) || (dm & KID_DIAG_U_ICEBERG) || (dm & KID_DIAG_V_ICEBERG)) ACC(KID_A_MASS, bs[KID_B_MASS] / area * ms);
    if (dm & KID_DIAG_U_ICEBERG) ACC(KID_A_U_ICEBERG, ((bs[KID_B_MASS] / area * ms) * bs[KID_B_UVEL]));
    if (dm & KID_DIAG_V_ICEBERG) ACC(KID_A_V_ICEBERG, ((bs[KID_B_MASS] / area * ms) * bs[KID_B_VVEL]));
    if ((dm & KID_DIAG_BERGY_MASS) || p->add_weight_to_ocean) ACC(KID_A_BERGY_MASS, (bs[KID_B_MASS_OF_BITS] + bs[KID_B_MASS_OF_FL_BERGY_BITS]) / area * ms);
    if ((dm & KID_DIAG_FL_BITS_MASS) || p->add_weight_to_ocean) ACC(KID_A_FL_BITS_MASS, bs[KID_B_MASS_OF_FL_BITS] / area * ms);
    if ((dm & KID_DIAG_FL_BERGY_BITS_MASS) || p->add_weight_to_ocean) ACC(KID_A_FL_BERGY_BITS_MASS, bs[KID_B_MASS_OF_FL_BERGY_BITS] / area * ms);
#undef ACC
  }
  free(perm);
}
/* the per-cell half: sum_up_spread_fields (IB:6077-6150) + IB:3449-3488.  In a particle-sharded run this is what
 * follows the all-reduce of the accumulators. */
static const double *g_spread_mass_tmp = NULL;   /* spread_mass_tmp of IB:3411-3413 (Iceberg_melt_without_decay) */
static const double *g_spread_mass_old = NULL;   /* grd%spread_mass_old while ko_run_step runs with find_melt_using_spread_mass */
static double *g_shard_spread_mass = NULL;   /* see ko_set_spread_mass_buffer */
static void gather_fields_core(const ko_grid *g, const kid_params *p, double *acc, double *out);
void ko_gather_fields(const ko_grid *g, const kid_params *p, double *acc, double *out) {
  const size_t ncell = (size_t)NI(g) * (size_t)(g->d.jed - g->d.jsd + 1);
  const int shard = p->find_melt_using_spread_mass && g_shard_spread_mass && !g_spread_mass_old;
  if (shard) { g_spread_mass_old = g_shard_spread_mass; g_spread_mass_tmp = p->Iceberg_melt_without_decay ? g_shard_spread_mass + ncell : NULL; }
  gather_fields_core(g, p, acc, out);
  if (shard) { g_spread_mass_old = NULL; g_spread_mass_tmp = NULL; }
}
static void gather_fields_core(const ko_grid *g, const kid_params *p, double *acc, double *out) {
  const size_t ncell = (size_t)NI(g) * (size_t)(g->d.jed - g->d.jsd + 1);
  const int dm = p->diag_mask;
  double *o_mass = out + (size_t)KID_O_SPREAD_MASS * ncell, *o_area = out + (size_t)KID_O_SPREAD_AREA * ncell;
  double *o_u = out + (size_t)KID_O_SPREAD_UVEL * ncell, *o_v = out + (size_t)KID_O_SPREAD_VVEL * ncell;
  double *o_us = out + (size_t)KID_O_USTAR_ICEBERG * ncell;
  g_wrap_x = p->periodic_reentry && g->d.Lx > 0.;
  memset(o_u, 0, ncell * sizeof(double)); memset(o_v, 0, ncell * sizeof(double)); memset(o_area, 0, ncell * sizeof(double));
  if ((dm & KID_DIAG_SPREAD_UVEL) || p->pass_fields_to_ocean_model) sum_up_spread_field(g, acc, KID_A_UVEL_ON_OCEAN, 0, o_u);
  if ((dm & KID_DIAG_SPREAD_VVEL) || p->pass_fields_to_ocean_model) sum_up_spread_field(g, acc, KID_A_VVEL_ON_OCEAN, 0, o_v);
  if ((dm & KID_DIAG_SPREAD_AREA) || p->pass_fields_to_ocean_model) sum_up_spread_field(g, acc, KID_A_AREA_ON_OCEAN, 1, o_area);
  sum_up_spread_field(g, acc, KID_A_MASS_ON_OCEAN, 0, o_mass);
  /* u_iceberg/v_iceberg normalisation IB:3450-3462 */
  if ((dm & KID_DIAG_U_ICEBERG) || (dm & KID_DIAG_V_ICEBERG)) {
    for (int j = g->d.jsc; j <= g->d.jec; ++j) for (int i = g->d.isc; i <= g->d.iec; ++i) {
      size_t c = GIDX(g, i, j);
      double m = acc[(size_t)KID_A_MASS * ncell + c];
      if (m > 0.) {
        if (dm & KID_DIAG_U_ICEBERG) acc[(size_t)KID_A_U_ICEBERG * ncell + c] /= m;
        if (dm & KID_DIAG_V_ICEBERG) acc[(size_t)KID_A_V_ICEBERG * ncell + c] /= m;
      } else {
        if (dm & KID_DIAG_U_ICEBERG) acc[(size_t)KID_A_U_ICEBERG * ncell + c] = 0.;
        if (dm & KID_DIAG_V_ICEBERG) acc[(size_t)KID_A_V_ICEBERG * ncell + c] = 0.;
      }
    }
  }
  memset(o_us, 0, ncell * sizeof(double));
  if ((dm & KID_DIAG_USTAR_ICEBERG) || p->pass_fields_to_ocean_model) {
    for (int j = g->d.jsc; j <= g->d.jec; ++j) for (int i = g->d.isc; i <= g->d.iec; ++i) {
      size_t c = GIDX(g, i, j);
      double a = o_u[c] - GF(g, KID_F_UO, i, j), bb = o_v[c] - GF(g, KID_F_VO, i, j);
      double dvo = sqrt(a * a + bb * bb);
      double ustar = sqrt(p->cdrag_icebergs * (dvo * dvo + p->utide_icebergs * p->utide_icebergs));
      double ustar_h = dmax(p->ustar_icebergs_bg, ustar);
      if (o_area[c] == 0.0) ustar_h = 0.;
      o_us[c] = ustar_h;
    }
  }
  if (g_spread_mass_old) {  /* find_melt_using_spread_mass, IB:3436-3445: spread_mass_tmp = spread_mass on the computational domain, 0 in the halo */
    for (int j = g->d.jsd; j <= g->d.jed; ++j) for (int i = g->d.isd; i <= g->d.ied; ++i) {
      size_t c = GIDX(g, i, j);
      const int in_c = i >= g->d.isc && i <= g->d.iec && j >= g->d.jsc && j <= g->d.jec;
      const double after = g_spread_mass_tmp ? g_spread_mass_tmp[c] : (in_c ? o_mass[c] : 0.);
      acc[(size_t)KID_A_FLOATING_MELT * ncell + c] = (GS(g, KID_G_AREA, i, j) > 0.0) ? dmax((g_spread_mass_old[c] - after) / p->dt, 0.0) : 0.0;
    }
  }
  if (p->apply_thickness_cutoff_to_gridded_melt) {
    for (int j = g->d.jsd; j <= g->d.jed; ++j) for (int i = g->d.isd; i <= g->d.ied; ++i) {
      size_t c = GIDX(g, i, j);
      if ((p->melt_cutoff >= 0.) && (o_area[c] > 0.)) {
        double ave_thickness = o_mass[c] / (o_area[c] * p->rho_bergs);
        double ave_draft = ave_thickness * (p->rho_bergs / RHO_SEAWATER);
        if ((GS(g, KID_G_OCEAN_DEPTH, i, j) - ave_draft) < p->melt_cutoff) {
          acc[(size_t)KID_A_FLOATING_MELT * ncell + c] = 0.0; acc[(size_t)KID_A_CALVING_HFLX * ncell + c] = 0.0;
        }
      }
    }
  }
}

void ko_create_gridded_icebergs_fields(const ko_grid *g, const kid_params *p, kid_berg_soa *b, double *acc, double *out) {
  ko_calculate_mass_on_ocean(g, p, b, acc);
  ko_gather_fields(g, p, acc, out);
}
/* everything of one step that is per berg (the part a rank does on its own shard); accumulators zeroed first */
/* Sharded runs with find_melt_using_spread_mass: the caller's two planes for grd%spread_mass_old and spread_mass_tmp, filled
 * by ko_step_local from THIS shard's bergs, summed over the ranks by the caller, read by ko_gather_fields. */
void ko_set_spread_mass_buffer(double *two_planes) { g_shard_spread_mass = two_planes; }
void ko_step_local(const ko_grid *g, const kid_params *p, kid_berg_soa *b, int64_t capacity, double *acc, double *scalars) {
  const size_t ncell = (size_t)NI(g) * (size_t)(g->d.jed - g->d.jsd + 1);
  memset(acc, 0, (size_t)KID_NACC * ncell * sizeof(double));
  if (!p->mts && !p->old_interp_flds_order) ko_interp_gridded_fields_to_bergs(g, p, b);
  if (!p->static_icebergs) ko_evolve_icebergs(g, p, b, scalars);
  if (p->footloose) ko_footloose_calving(g, p, b, capacity, acc, scalars);
  if (!p->old_interp_flds_order) ko_interp_gridded_fields_to_bergs(g, p, b);
  const int fm = p->find_melt_using_spread_mass && g_shard_spread_mass;
  if (fm) {  /* IB:5490-5503, this shard's share */
    g_mass_only = 1; ko_calculate_mass_on_ocean(g, p, b, acc); g_mass_only = 0;
    memset(g_shard_spread_mass, 0, 2 * ncell * sizeof(double));
    g_wrap_x = p->periodic_reentry && g->d.Lx > 0.;
    sum_up_spread_field(g, acc, KID_A_MASS_ON_OCEAN, 0, g_shard_spread_mass);
    for (int s = 0; s < 36; ++s) memset(acc + (size_t)(KID_A_MASS_ON_OCEAN + s) * ncell, 0, ncell * sizeof(double));
  }
  ko_thermodynamics(g, p, b, acc, scalars);
  if (fm && p->Iceberg_melt_without_decay) {  /* IB:3411-3413 */
    sum_up_spread_field(g, acc, KID_A_MASS_ON_OCEAN, 0, g_shard_spread_mass + ncell);
    for (int s = 0; s < 36; ++s) memset(acc + (size_t)(KID_A_MASS_ON_OCEAN + s) * ncell, 0, ncell * sizeof(double));
  }
  ko_calculate_mass_on_ocean(g, p, b, acc);
}
/* one icebergs_run() worth of the hot path, IB:5423-5512 (non-MTS, non-interactive) */
void ko_run_step(const ko_grid *g, const kid_params *p, kid_berg_soa *b, int64_t capacity,
                 double *acc, double *out, double *scalars) {
  const size_t ncell = (size_t)NI(g) * (size_t)(g->d.jed - g->d.jsd + 1);
  memset(acc, 0, (size_t)KID_NACC * ncell * sizeof(double));
  if (!p->mts && !p->old_interp_flds_order) ko_interp_gridded_fields_to_bergs(g, p, b);
  if (!p->static_icebergs) ko_evolve_icebergs(g, p, b, scalars);
  if (p->footloose) ko_footloose_calving(g, p, b, capacity, acc, scalars);
  if (!p->old_interp_flds_order) ko_interp_gridded_fields_to_bergs(g, p, b);
  double *spread_mass_old = NULL;
  if (p->find_melt_using_spread_mass) {  /* IB:5490-5503 */
    g_mass_only = 1; ko_calculate_mass_on_ocean(g, p, b, acc); g_mass_only = 0;
    spread_mass_old = (double *)calloc(ncell, sizeof(double));
    g_wrap_x = p->periodic_reentry && g->d.Lx > 0.;
    sum_up_spread_field(g, acc, KID_A_MASS_ON_OCEAN, 0, spread_mass_old);
    for (int s = 0; s < 36; ++s) memset(acc + (size_t)(KID_A_MASS_ON_OCEAN + s) * ncell, 0, ncell * sizeof(double));
  }
  ko_thermodynamics(g, p, b, acc, scalars);
  double *spread_mass_tmp = NULL;
  if (spread_mass_old && p->Iceberg_melt_without_decay) {  /* IB:3411-3413: what thermodynamics spread is the mass after the melt */
    spread_mass_tmp = (double *)calloc(ncell, sizeof(double));
    sum_up_spread_field(g, acc, KID_A_MASS_ON_OCEAN, 0, spread_mass_tmp);
  }
  g_spread_mass_tmp = spread_mass_tmp;
  g_spread_mass_old = spread_mass_old;   /* IB:3436-3445 happens inside create_gridded_icebergs_fields, before the cutoff of IB:3477 */
  ko_create_gridded_icebergs_fields(g, p, b, acc, out);
  g_spread_mass_old = NULL; g_spread_mass_tmp = NULL;
  free(spread_mass_old); free(spread_mass_tmp);
  int64_t alive = 0;
  for (int64_t k = 0; k < b->n; ++k) alive += berg_alive(b, k);
  scalars[KID_S_NBERGS_ALIVE] = (double)alive;
}

/* footloose_calving: see oracle/kid_oracle_footloose.c */
int64_t ko_sizeof(int which) {
  switch (which) { case 0: return (int64_t)sizeof(kid_params); case 1: return (int64_t)sizeof(kid_grid_desc);
                   case 2: return (int64_t)sizeof(kid_berg_soa); case 3: return (int64_t)sizeof(ko_grid); default: return -1; }
}

/* ---- berg migration between sub-domains: the loops of send_bergs_to_other_pes around its mpp calls ---- */

/* check_and_find_cell, FW:5973-6008: the indices handed in, then the regular-grid guess, then a scan of the data domain */
int ko_check_and_find_cell(const ko_grid *g, double x, double y, int *oi, int *oj) {
  const kid_grid_desc *d = &g->d;
  if (!(*oi - 1 < d->isd || *oi > d->ied || *oj - 1 < d->jsd || *oj > d->jed) && ko_is_point_in_cell(g, x, y, *oi, *oj)) return 1;
  const double lon0 = GS(g, KID_G_LON, d->isd, d->jsd), lon1 = GS(g, KID_G_LON, d->isd + 1, d->jsd + 1);
  const double lat0 = GS(g, KID_G_LAT, d->isd, d->jsd), lat1 = GS(g, KID_G_LAT, d->isd + 1, d->jsd + 1);
  *oi = (int)floor((x - lon0) / (lon1 - lon0)) + d->isd + 1;                              /* FW:5993-5994 */
  *oj = (int)floor((y - lat0) / (lat1 - lat0)) + d->jsd + 1;
  if (!(*oi - 1 < d->isd || *oi > d->ied || *oj - 1 < d->jsd || *oj > d->jed) && ko_is_point_in_cell(g, x, y, *oi, *oj)) return 1;
  *oi = -999; *oj = -999;
  for (int j = d->jsd + 1; j <= d->jed; ++j)
    for (int i = d->isd + 1; i <= d->ied; ++i)
      if (ko_is_point_in_cell(g, x, y, i, j)) { *oi = i; *oj = j; return 1; }
  return 0;
}

/* the selection and packing loops of send_bergs_to_other_pes (FW:3022-3048 east/west, FW:3099-3124 north/south) with
 * pack_berg_into_buffer2 for bergs without bonds (FW:3268-3301: 34 reals, integers as reals).  dir 0 E, 1 W, 2 N, 3 S.
 * A berg the step deleted for leaving the computational domain (alive == 0, its cell outside) is what the reference
 * still holds at this point; it is packed and its cell index moved inside (the deletion, FW:3034).  Returns the count. */
long ko_send_bergs(const ko_grid *g, kid_berg_soa *b, int dir, double *buf) {
  const kid_grid_desc *d = &g->d;
  long n = 0;
  for (long k = 0; k < (long)b->n; ++k) {
    if (b->f64[KID_B_HALO_BERG] && b->f64[KID_B_HALO_BERG][k] >= 0.5) continue;           /* FW:3027 */
    const int i = b->i32[KID_BI_INE][k], j = b->i32[KID_BI_JNE][k];
    const int take = dir == 0 ? i > d->iec : dir == 1 ? i < d->isc : dir == 2 ? j > d->jec : j < d->jsc;
    if (!take) continue;
    double *o = buf + (size_t)n * 34;
    static const int order[34] = {KID_B_LON, KID_B_LAT, KID_B_UVEL, KID_B_VVEL, KID_B_UVEL_PREV, KID_B_VVEL_PREV, KID_B_XI, KID_B_YJ,
      KID_B_START_LON, KID_B_START_LAT, -1, KID_B_START_DAY, KID_B_START_MASS, KID_B_MASS, KID_B_THICKNESS, KID_B_WIDTH, KID_B_LENGTH,
      KID_B_FL_K, KID_B_MASS_SCALING, KID_B_MASS_OF_BITS, KID_B_MASS_OF_FL_BITS, KID_B_MASS_OF_FL_BERGY_BITS, KID_B_HEAT_DENSITY, -2, -3,
      KID_B_AXN, KID_B_AYN, KID_B_BXN, KID_B_BYN, KID_B_HALO_BERG, KID_B_STATIC_BERG, -4, -5, KID_B_OD};
    for (int q = 0; q < 34; ++q) {
      const int f = order[q];
      if (f >= 0) o[q] = b->f64[f] ? b->f64[f][k] : 0.;
      else if (f == -1) o[q] = (double)b->i32[KID_BI_START_YEAR][k];
      else if (f == -2) o[q] = (double)i;
      else if (f == -3) o[q] = (double)j;
      else if (f == -4) o[q] = (double)(int32_t)(b->id[k] >> 32);                         /* split_id, FW:3297 */
      else o[q] = (double)(int32_t)(b->id[k] & 0xffffffffll);
    }
    b->i32[KID_BI_ALIVE][k] = 0; b->i32[KID_BI_INE][k] = d->isc; b->i32[KID_BI_JNE][k] = d->jsc;
    ++n;
  }
  return n;
}

/* unpack_berg_from_buffer2 without bonds (FW:3503-3541, 3573-3577, 3626-3637) into rows b->n .. b->n + m - 1 of arrays
 * that have the room; returns the number of bergs no cell took (FATAL in the reference, FW:3660; dropped here) */
long ko_unpack_bergs(const ko_grid *g, const kid_params *p, kid_berg_soa *b, const double *buf, long m) {
  long lost = 0;
  for (long q = 0; q < m; ++q) {
    const double *r = buf + (size_t)q * 34;
    const long k = (long)b->n;
    for (int f = 0; f < KID_NB_F64; ++f) if (b->f64[f]) b->f64[f][k] = 0.;
    for (int f = 0; f < KID_NB_I32; ++f) if (b->i32[f]) b->i32[f][k] = 0;
    PUT(b, KID_B_LON, k, r[0]); PUT(b, KID_B_LAT, k, r[1]); PUT(b, KID_B_UVEL, k, r[2]); PUT(b, KID_B_VVEL, k, r[3]);
    PUT(b, KID_B_UVEL_PREV, k, r[4]); PUT(b, KID_B_VVEL_PREV, k, r[5]); PUT(b, KID_B_XI, k, r[6]); PUT(b, KID_B_YJ, k, r[7]);
    PUT(b, KID_B_START_LON, k, r[8]); PUT(b, KID_B_START_LAT, k, r[9]); b->i32[KID_BI_START_YEAR][k] = (int)lrint(r[10]);
    PUT(b, KID_B_START_DAY, k, r[11]); PUT(b, KID_B_START_MASS, k, r[12]); PUT(b, KID_B_MASS, k, r[13]); PUT(b, KID_B_THICKNESS, k, r[14]);
    PUT(b, KID_B_WIDTH, k, r[15]); PUT(b, KID_B_LENGTH, k, r[16]); PUT(b, KID_B_FL_K, k, r[17]); PUT(b, KID_B_MASS_SCALING, k, r[18]);
    PUT(b, KID_B_MASS_OF_BITS, k, r[19]); PUT(b, KID_B_MASS_OF_FL_BITS, k, r[20]); PUT(b, KID_B_MASS_OF_FL_BERGY_BITS, k, r[21]);
    PUT(b, KID_B_HEAT_DENSITY, k, r[22]);
    PUT(b, KID_B_AXN, k, r[25]); PUT(b, KID_B_AYN, k, r[26]); PUT(b, KID_B_BXN, k, r[27]); PUT(b, KID_B_BYN, k, r[28]);
    PUT(b, KID_B_HALO_BERG, k, r[29]); PUT(b, KID_B_STATIC_BERG, k, r[30]); PUT(b, KID_B_OD, k, r[33]);
    b->id[k] = (int64_t)(((uint64_t)(uint32_t)(int32_t)lrint(r[31]) << 32) | (uint64_t)(uint32_t)(int32_t)lrint(r[32]));   /* id_from_2_ints */
    PUT(b, KID_B_UVEL_OLD, k, r[2]); PUT(b, KID_B_VVEL_OLD, k, r[3]); PUT(b, KID_B_LON_OLD, k, r[0]); PUT(b, KID_B_LAT_OLD, k, r[1]);   /* FW:3573-3577 */
    int i = (int)lrint(r[23]), j = (int)lrint(r[24]);
    if (!ko_check_and_find_cell(g, r[0], r[1], &i, &j)) { ++lost; continue; }             /* FW:3628; find_cell_wide repeats the same scan */
    double xi, yj; int perr = 0;
    (void)ko_pos_within_cell(g, p, r[0], r[1], i, j, &xi, &yj, &perr);                    /* FW:3634 */
    b->i32[KID_BI_INE][k] = i; b->i32[KID_BI_JNE][k] = j; PUT(b, KID_B_XI, k, xi); PUT(b, KID_B_YJ, k, yj);
    b->i32[KID_BI_ALIVE][k] = 1;
    b->n += 1;
  }
  return lost;
}

/* ------------------------------------------------------------------------------------------------
 * bergs_chksum FW:6889-6987, berg_chksum FW:6990-7068, time_hash / pos_hash FW:4364-4376; mpp_chksum (FMS, not in the
 * reference tree): wrap-around 64-bit sum of the IEEE bit patterns, kept as a default integer (low 32 bits).
 * Followed to the letter: i and chksum5 start over in every cell (FW:6919); transfer(rtmp,i8) has a scalar mold, so
 * itmp(1:36) are 36 copies of the low word of rtmp(1) = lon (FW:7055).  b must be in reference traversal order
 * (ko_reference_order).  out = chksum, chksum2, chksum3, chksum4, chksum5, # of bergs on the computational domain.
 * ---------------------------------------------------------------------------------------------- */
static uint64_t dbits(double x) { uint64_t u; memcpy(&u, &x, sizeof(u)); return u; }
static int32_t one_berg_chksum(const kid_berg_soa *b, int64_t k) {
  uint32_t itmp[43];
  const uint32_t w = (uint32_t)dbits(b->f64[KID_B_LON][k]);
  for (int q = 0; q < 36; ++q) itmp[q] = w;
  itmp[36] = (uint32_t)(int32_t)(b->f64[KID_B_HALO_BERG] ? b->f64[KID_B_HALO_BERG][k] : 0.);
  itmp[37] = (uint32_t)(int32_t)(b->f64[KID_B_STATIC_BERG] ? b->f64[KID_B_STATIC_BERG][k] : 0.);
  itmp[38] = (uint32_t)(b->i32[KID_BI_START_YEAR] ? b->i32[KID_BI_START_YEAR][k] : 0);
  itmp[39] = (uint32_t)b->i32[KID_BI_INE][k]; itmp[40] = (uint32_t)b->i32[KID_BI_JNE][k];
  const int64_t id = b->id ? b->id[k] : 0;
  itmp[41] = (uint32_t)(int32_t)(id >> 32); itmp[42] = (uint32_t)(int32_t)(id & 0xFFFFFFFFll);
  uint32_t c1 = 0, c2 = 0, c3 = 0;
  for (uint32_t q = 1; q <= 43; ++q) { c1 += itmp[q - 1]; c2 += itmp[q - 1] * q; c3 += itmp[q - 1] * q * q; }
  return (int32_t)(c1 + c2 + c3);
}
void ko_bergs_chksum(const ko_grid *g, const kid_berg_soa *b, int64_t out[6]) {
  const kid_grid_desc *d = &g->d;
  const size_t ncell = (size_t)NI(g) * (size_t)(d->jed - d->jsd + 1);
  int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(b->n > 0 ? b->n : 1));
  ko_reference_order(b, perm);
  int64_t nb = 0;
  for (int64_t q = 0; q < b->n; ++q) {
    const int64_t k = perm[q];
    const int i = b->i32[KID_BI_INE][k], j = b->i32[KID_BI_JNE][k];
    if (berg_alive(b, k) && i >= d->isc && i <= d->iec && j >= d->jsc && j <= d->jec) nb += 1;
  }
  const int64_t rows = nb > 0 ? nb : 1;
  double *fld = (double *)calloc((size_t)rows * 19, sizeof(double)), *fld2 = (double *)calloc((size_t)rows * 19, sizeof(double));
  double *tmp = (double *)calloc(ncell, sizeof(double));
  int32_t *icnt = (int32_t *)calloc(ncell, sizeof(int32_t));
  static const int src[16] = {KID_B_LON, KID_B_LAT, KID_B_UVEL, KID_B_VVEL, KID_B_MASS, KID_B_THICKNESS, KID_B_WIDTH, KID_B_LENGTH,
                              KID_B_AXN, KID_B_AYN, KID_B_BXN, KID_B_BYN, KID_B_UVEL_OLD, KID_B_VVEL_OLD, KID_B_LON_OLD, KID_B_LAT_OLD};
  uint32_t ichk5 = 0;
  int64_t q = 0;
  for (int gj = d->jsc; gj <= d->jec; ++gj)
    for (int gi = d->isc; gi <= d->iec; ++gi) {
      int64_t i = 0; ichk5 = 0;
      /* perm is sorted by (jne, ine, inorder): skip what lies before this cell (dead rows, rows outside the domain) */
      while (q < b->n && (b->i32[KID_BI_JNE][perm[q]] < gj || (b->i32[KID_BI_JNE][perm[q]] == gj && b->i32[KID_BI_INE][perm[q]] < gi))) ++q;
      for (; q < b->n && b->i32[KID_BI_JNE][perm[q]] == gj && b->i32[KID_BI_INE][perm[q]] == gi; ++q) {
        const int64_t k = perm[q];
        if (!berg_alive(b, k)) continue;
        const int32_t iberg = one_berg_chksum(b, k);
        const double sy = (double)(b->i32[KID_BI_START_YEAR] ? b->i32[KID_BI_START_YEAR][k] : 0);
        const double time_hash = b->f64[KID_B_START_DAY][k] + 366. * sy;
        const double pos_hash = b->f64[KID_B_START_LON][k] + 360. * (b->f64[KID_B_START_LAT][k] + 90.);
        double *row = fld + (size_t)i * 19;
        for (int c = 0; c < 16; ++c) row[c] = b->f64[src[c]] ? b->f64[src[c]][k] : 0.;
        row[16] = time_hash; row[17] = pos_hash; row[18] = (double)iberg;
        const size_t cc = GIDX(g, gi, gj);
        icnt[cc] += 1;
        for (int c = 0; c < 19; ++c) fld2[(size_t)i * 19 + c] = row[c] * (double)icnt[cc];
        tmp[cc] = tmp[cc] + time_hash * pos_hash + log(b->f64[KID_B_MASS][k]);
        ichk5 += (uint32_t)iberg;
        ++i;
      }
    }
  uint64_t s1 = 0, s2 = 0, s3 = 0, s4 = 0;
  for (int64_t c = 0; c < rows * 19; ++c) { s1 += dbits(fld[c]); s2 += dbits(fld2[c]); }
  for (int gj = d->jsd; gj <= d->jed; ++gj)
    for (int gi = d->isd; gi <= d->ied; ++gi) {
      const uint64_t u = dbits(tmp[GIDX(g, gi, gj)]);
      s3 += u;
      if (gi >= d->isc && gi <= d->iec && gj >= d->jsc && gj <= d->jec) s4 += u;
    }
  out[0] = (int32_t)(uint32_t)s1; out[1] = (int32_t)(uint32_t)s2; out[2] = (int32_t)(uint32_t)s3; out[3] = (int32_t)(uint32_t)s4;
  out[4] = (int32_t)ichk5; out[5] = nb;
  free(perm); free(fld); free(fld2); free(tmp); free(icnt);
}

/* kid_oracle_footloose.c -- CPU restatement (ORACLE, test infrastructure) of footloose calving.
 *   footloose_calving   /root/reference/src/icebergs.F90:2503-2734 (get_footloose_displacement IB:2688-2732)
 *   calve_fl_icebergs   /root/reference/src/icebergs.F90:6405-6569
 *   find_cell           /root/reference/src/icebergs_framework.F90:6011-6040
 *   generate_id         /root/reference/src/icebergs_framework.F90:4165-4179, id_from_2_ints FW:7276-7282
 * displace_fl_bergs (the namelist default): the child is put at a random place on the parent's perimeter.  The reference
 * draws that number from FMS's Mersenne-Twister stream (IB:2548-2550, 2631, 2664), which is not in the reference tree and
 * whose sequence depends on the traversal and the PE layout; here it is the counter-based generator of include/kid_rng.h,
 * keyed by (seed, parent id, footloose step, draw) -- the same function in the HIP library.  Everything downstream of the
 * number (side and offset, metres -> degrees, tangent plane, find_cell, the four corner fall-backs with their metres-as-
 * degrees arithmetic, the grounded-cell fall-back, pos_within_cell) follows the reference line by line.
 * Bonded footloose calving is a FATAL in the reference.
 * One deliberate difference: the "new berg from FL bits" branch (IB:2663-2667) uses l_b of the berg at hand; the
 * reference reuses the local l_b left by whichever berg last went through the calving block.
 * Children are appended to the SoA in traversal order (the reference inserts them into the child's cell list).
 * PARITY UNPINNED: the reference holds no vector for this path that can be recomputed here (its footloose regression
 * numbers need netCDF restarts and FMS's random stream); the restatement is checked by reading and by properties.
 */
#include "kid_oracle.h"
#include <stddef.h>
#include <stdlib.h>
#include <math.h>

#define RHO_SEAWATER 1025.0
#define GRAVITY 9.8
#define NI(g) ((g)->d.ied - (g)->d.isd + 1)
#define GIDX(g, i, j) ((size_t)((i) - (g)->d.isd) + (size_t)((j) - (g)->d.jsd) * (size_t)NI(g))

#include "../include/kid_rng.h"
#define GS(g, f, i, j) ((g)->stat[(f)][GIDX(g, i, j)])

/* the footloose step: counts the calls of footloose_calving (the third word of the generator's counter) */
static uint32_t g_fl_step = 0;
void ko_set_fl_step(int64_t step) { g_fl_step = (uint32_t)step; }
int64_t ko_get_fl_step(void) { return (int64_t)g_fl_step; }

/* the generator itself, for the known-answer tests (tests/test_oracle_pins.py) */
void ko_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { kid_philox4x32_10(ctr, key, out); }
double ko_fl_uniform(int32_t seed, int64_t berg_id, int64_t step, int32_t draw) { return kid_fl_uniform((uint32_t)seed, berg_id, (uint32_t)step, (uint32_t)draw); }

/* find_cell FW:6011-6040: the structured-grid guess, then a scan of the computational domain */
static int find_cell(const ko_grid *g, double x, double y, int *oi, int *oj) {
  const kid_grid_desc *d = &g->d;
  const double lon0 = GS(g, KID_G_LON, d->isd, d->jsd), lon1 = GS(g, KID_G_LON, d->isd + 1, d->jsd + 1);
  const double lat0 = GS(g, KID_G_LAT, d->isd, d->jsd), lat1 = GS(g, KID_G_LAT, d->isd + 1, d->jsd + 1);
  *oi = (int)floor((x - lon0) / (lon1 - lon0)) + d->isd + 1;
  *oj = (int)floor((y - lat0) / (lat1 - lat0)) + d->jsd + 1;
  if (*oi > d->isc - 1 && *oi < d->iec + 1 && *oj > d->jsc - 1 && *oj < d->jec + 1)
    if (ko_is_point_in_cell(g, x, y, *oi, *oj)) return 1;
  *oi = -999; *oj = -999;
  for (int j = d->jsc; j <= d->jec; ++j)
    for (int i = d->isc; i <= d->iec; ++i)
      if (ko_is_point_in_cell(g, x, y, i, j)) { *oi = i; *oj = j; return 1; }
  return 0;
}

/* get_footloose_displacement IB:2688-2732: a place on the parent's perimeter from rn, as a displacement in grid units */
static void footloose_displacement(const ko_grid *g, const kid_params *p, double rn, double lon, double lat, double length, double width,
                                   double *fl_disp_x, double *fl_disp_y) {
  double fx, fy, interp_loc;
  if (rn < 0.25) { interp_loc = 4. * rn; fx = length * (interp_loc - 0.5); fy = 0.5 * width; }                 /* north side */
  else if (rn < 0.5) { interp_loc = 4. * (rn - 0.25); fx = 0.5 * length; fy = width * (interp_loc - 0.5); }   /* east side */
  else if (rn < 0.75) { interp_loc = 4. * (rn - 0.5); fx = length * (interp_loc - 0.5); fy = -0.5 * width; }  /* south side */
  else { interp_loc = 4. * (rn - 0.75); fx = -0.5 * length; fy = 0.5 * width * (interp_loc - 0.5); }          /* west side (the 0.5 is the reference's, IB:2714) */
  if (g->d.grid_is_latlon) {
    const int on_tang = (lat > 89.) && g->d.grid_is_latlon;
    double lon1 = lon, lat1 = lat, x1 = 0., y1 = 0., dxdl1, dydl;
    if (on_tang) ko_rotpos_to_tang(p, lon1, lat1, &x1, &y1);
    ko_meters_to_grid(g, p, lat1, &dxdl1, &dydl);
    if (on_tang) {
      double xdot2, ydot2;
      ko_rotvec_to_tang(p, lon1, fx, fy, &xdot2, &ydot2);
      x1 = x1 + xdot2; y1 = y1 + ydot2;
      ko_rotpos_from_tang(p, x1, y1, &lon1, &lat1);
    } else { lon1 = lon1 + fx * dxdl1; lat1 = lat1 + fy * dydl; }
    fx = lon1 - lon; fy = lat1 - lat;
  }
  *fl_disp_x = fx; *fl_disp_y = fy;
}

static double getf(const kid_berg_soa *b, int f, int64_t k) { return b->f64[f] ? b->f64[f][k] : 0.0; }
static void putf(kid_berg_soa *b, int f, int64_t k, double v) { if (b->f64[f]) b->f64[f][k] = v; }

/* calve_fl_icebergs IB:6405-6569; `draw` picks the random number of this event; returns 0 if the SoA is full */
static int calve_child(const ko_grid *g, const kid_params *p, kid_berg_soa *b, int64_t capacity, int64_t pk,
                       double k, double l_b, int from_bits, uint32_t draw) {
  if (b->n >= capacity) return 0;
  const int64_t c = b->n;
  for (int f = 0; f < KID_NB_F64; ++f) putf(b, f, c, 0.0);
  const double plon = getf(b, KID_B_LON, pk), plat = getf(b, KID_B_LAT, pk);
  double fl_disp_x = 0.0, fl_disp_y = 0.0;
  int displace = p->displace_fl_bergs != 0;
  if (displace) {
    /* IB:2631 / 2664: a fresh number per event, or the one number of the run (fl_init_child_xy_by_pe) */
    const double rn = p->fl_init_child_xy_by_pe ? kid_fl_uniform((uint32_t)p->fl_rng_seed, 0, 0u, 0u)
                                                : kid_fl_uniform((uint32_t)p->fl_rng_seed, b->id ? b->id[pk] : 0, g_fl_step, draw);
    const double plen = getf(b, KID_B_LENGTH, pk), pwid = getf(b, KID_B_WIDTH, pk);
    footloose_displacement(g, p, rn, plon, plat, plen, pwid, &fl_disp_x, &fl_disp_y);
    double clon = plon + fl_disp_x, clat = plat + fl_disp_y;   /* IB:6433-6435 */
    int ci, cj;
    int lres = find_cell(g, clon, clat, &ci, &cj);
    if (!lres) {  /* not on this PE's computational domain: try the corners (IB:6438-6467; metres added to degrees as written) */
      clon = plon - 0.5 * plen; clat = plat - 0.5 * pwid; lres = find_cell(g, clon, clat, &ci, &cj);
      if (!lres) { clon = plon - 0.5 * plen; clat = plat + 0.5 * pwid; lres = find_cell(g, clon, clat, &ci, &cj); }
      if (!lres) { clon = plon + 0.5 * plen; clat = plat + 0.5 * pwid; lres = find_cell(g, clon, clat, &ci, &cj); }
      if (!lres) { clon = plon + 0.5 * plen; clat = plat - 0.5 * pwid; lres = find_cell(g, clon, clat, &ci, &cj); }
      if (!lres) { fl_disp_x = 0.0; fl_disp_y = 0.0; displace = 0; }
      else { fl_disp_x = plon - clon; fl_disp_y = plat - clat; }   /* (sign as in the reference, IB:6466) */
    }
    if (displace) {
      if (GS(g, KID_G_AREA, ci, cj) == 0.) { fl_disp_x = 0.0; fl_disp_y = 0.0; displace = 0; }   /* grounded cell IB:6471-6472 */
      else {
        double xi, yj; int err = 0;
        (void)ko_pos_within_cell(g, p, clon, clat, ci, cj, &xi, &yj, &err);
        putf(b, KID_B_LON, c, clon); putf(b, KID_B_LAT, c, clat); putf(b, KID_B_XI, c, xi); putf(b, KID_B_YJ, c, yj);
        b->i32[KID_BI_INE][c] = ci; b->i32[KID_BI_JNE][c] = cj;
      }
    }
  }
  if (!displace) {  /* position = the parent's (IB:6479-6486) */
    putf(b, KID_B_LON, c, plon); putf(b, KID_B_LAT, c, plat);
    putf(b, KID_B_XI, c, getf(b, KID_B_XI, pk)); putf(b, KID_B_YJ, c, getf(b, KID_B_YJ, pk));
    b->i32[KID_BI_INE][c] = b->i32[KID_BI_INE][pk]; b->i32[KID_BI_JNE][c] = b->i32[KID_BI_JNE][pk];
  }
  const double pms = getf(b, KID_B_MASS_SCALING, pk);
  if (from_bits) { /* IB:6488-6497 */
    double Lfl, Wfl, Tfl;
    ko_fl_bits_dimensions(p, getf(b, KID_B_THICKNESS, pk), &Lfl, &Wfl, &Tfl);
    const double cmass = Tfl * Lfl * Wfl * p->rho_bergs;
    const double cms = k * p->new_berg_from_fl_bits_mass_thres / cmass;
    putf(b, KID_B_LENGTH, c, Lfl); putf(b, KID_B_WIDTH, c, Wfl); putf(b, KID_B_THICKNESS, c, Tfl);
    putf(b, KID_B_MASS, c, cmass); putf(b, KID_B_MASS_SCALING, c, cms);
    const double percent_fl = (cmass * cms) / (getf(b, KID_B_MASS_OF_FL_BITS, pk) * pms);
    putf(b, KID_B_MASS_OF_BITS, c, (percent_fl * getf(b, KID_B_MASS_OF_FL_BERGY_BITS, pk) * pms) / cms);
    putf(b, KID_B_MASS_OF_FL_BERGY_BITS, pk, (1 - percent_fl) * getf(b, KID_B_MASS_OF_FL_BERGY_BITS, pk));
    putf(b, KID_B_MASS_OF_FL_BITS, pk, getf(b, KID_B_MASS_OF_FL_BITS, pk) - k * p->new_berg_from_fl_bits_mass_thres / pms);
  } else { /* IB:6499-6504 */
    const double len = l_b * 3., wid = l_b, thick = getf(b, KID_B_THICKNESS, pk);
    putf(b, KID_B_LENGTH, c, len); putf(b, KID_B_WIDTH, c, wid); putf(b, KID_B_THICKNESS, c, thick);
    putf(b, KID_B_MASS, c, wid * len * thick * p->rho_bergs);
    putf(b, KID_B_MASS_SCALING, c, pms * k);
    putf(b, KID_B_MASS_OF_BITS, c, 0.0);
  }
  putf(b, KID_B_START_LON, c, getf(b, KID_B_LON, c)); putf(b, KID_B_START_LAT, c, getf(b, KID_B_LAT, c));
  putf(b, KID_B_LON_OLD, c, getf(b, KID_B_LON_OLD, pk) + fl_disp_x); putf(b, KID_B_LAT_OLD, c, getf(b, KID_B_LAT_OLD, pk) + fl_disp_y);
  putf(b, KID_B_START_DAY, c, p->current_yearday);
  putf(b, KID_B_MASS_OF_FL_BITS, c, 0.0); putf(b, KID_B_MASS_OF_FL_BERGY_BITS, c, 0.0);
  putf(b, KID_B_FL_K, c, -1.0);
  if (b->i32[KID_BI_START_YEAR]) b->i32[KID_BI_START_YEAR][c] = p->current_year;
  { /* generate_id FW:4165-4179 at the parent's cell */
    const int i = b->i32[KID_BI_INE][pk], j = b->i32[KID_BI_JNE][pk];
    int32_t cnt = 1;
    if (g->iceberg_counter) { g->iceberg_counter[GIDX(g, i, j)] += 1; cnt = g->iceberg_counter[GIDX(g, i, j)]; }
    const int iNg = g->d.gni > 0 ? g->d.gni : g->d.iec - g->d.isc + 1, ij0 = g->d.gni > 0 ? g->d.gi0 + g->d.gni * g->d.gj0 : 0;
    const int32_t ij = i + (iNg * (j - 1)) + ij0;
    if (b->id) b->id[c] = (int64_t)cnt * ((int64_t)1 << 32) + (int64_t)ij;
  }
  putf(b, KID_B_HALO_BERG, c, 0.0);
  static const int same[] = {KID_B_START_MASS, KID_B_UVEL, KID_B_VVEL, KID_B_AXN, KID_B_AYN, KID_B_BXN, KID_B_BYN,
                             KID_B_UVEL_PREV, KID_B_VVEL_PREV, KID_B_UVEL_OLD, KID_B_VVEL_OLD, KID_B_HEAT_DENSITY,
                             KID_B_STATIC_BERG, KID_B_UO, KID_B_VO, KID_B_UI, KID_B_VI, KID_B_UA, KID_B_VA, KID_B_SSH_X,
                             KID_B_SSH_Y, KID_B_SST, KID_B_SSS, KID_B_CN, KID_B_HI, KID_B_OD};
  for (unsigned q = 0; q < sizeof(same) / sizeof(same[0]); ++q) putf(b, same[q], c, getf(b, same[q], pk));
  if (b->i32[KID_BI_N_BONDS]) b->i32[KID_BI_N_BONDS][c] = 0;
  if (b->i32[KID_BI_ALIVE]) b->i32[KID_BI_ALIVE][c] = 1;
  b->n += 1;
  return 1;
}

void ko_footloose_calving(const ko_grid *g, const kid_params *p, kid_berg_soa *b, int64_t capacity, double *acc, double *scalars) {
  const size_t ncell = (size_t)NI(g) * (size_t)(g->d.jed - g->d.jsd + 1);
  /* constants IB:2538-2547 */
  const double e1 = exp(0.25 * p->pi), drho = RHO_SEAWATER - p->rho_bergs, sigmay = p->fl_strength * 1000;
  const double lfootparam = e1 * RHO_SEAWATER * sigmay / (6 * p->rho_bergs * GRAVITY * drho);
  const double poisson = 0.3, youngs = p->fl_youngs;
  const double l_c = p->pi / (2. * sqrt(2.)), lw_c = 1. / (GRAVITY * RHO_SEAWATER), B_c = youngs / (12. * (1. - pow(poisson, 2.)));
  const int64_t n0 = b->n;
  /* the loop of IB:2552-2677 in its own order: cells j outer / i inner, a cell's list in `inorder` -- the order in
   * which generate_id hands out a cell's counter values when several of its bergs calve in one step */
  int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n0 > 0 ? n0 : 1));
  ko_reference_order(b, perm);
  for (int64_t kk = 0; kk < n0; ++kk) {
    const int64_t q = perm[kk];
    if (b->i32[KID_BI_ALIVE] && !b->i32[KID_BI_ALIVE][q]) continue;
    const int i = b->i32[KID_BI_INE][q], j = b->i32[KID_BI_JNE][q];
    if (i < g->d.isc || i > g->d.iec || j < g->d.jsc || j > g->d.jec) continue; /* computational domain only IB:2554 */
    const size_t c = GIDX(g, i, j);
    const double area = g->stat[KID_G_AREA][c], ms = getf(b, KID_B_MASS_SCALING, q);
    double l_b;
    {
      const double Tq = getf(b, KID_B_THICKNESS, q);
      l_b = l_c * pow(lw_c * B_c * pow(Tq, 3.), 0.25);
    }
    if (!(getf(b, KID_B_STATIC_BERG, q) == 1 || getf(b, KID_B_FL_K, q) < 0)) {
      double T = getf(b, KID_B_THICKNESS, q), W = getf(b, KID_B_WIDTH, q), L = getf(b, KID_B_LENGTH, q);
      const double l_w = pow(lw_c * B_c * pow(T, 3.), 0.25);
      l_b = l_c * l_w;
      const double l_b3 = 3 * l_b;
      double cc = ceil((L - l_b3) / l_b3); const double Lmin = L - cc * l_b3;
      cc = ceil((W - l_b3) / l_b3); const double Wmin = W - cc * l_b3;
      const double max_k = fmax(floor((L * W - Lmin * Wmin) / (l_b3 * l_b)), 0);
      double k;
      if (max_k == 0) k = 0;
      else {
        const double foot_l = lfootparam * T / l_w;
        const double foot_area = foot_l * l_b3;
        k = floor(getf(b, KID_B_FL_K, q) / foot_area);
        if (k > max_k) k = max_k;
        putf(b, KID_B_FL_K, q, getf(b, KID_B_FL_K, q) - k * foot_area);
      }
      if (k > 0) {
        double ds, Ln, Wn;
        if (cc > 0) {
          ds = 0.5 * ((L + W) - sqrt(pow(L + W, 2.) - 4. * (l_b3 * l_b * k)));
          Ln = L - ds; Wn = W - ds;
          if (Wn < Wmin) { Ln = Ln * (1 - (Wmin - Wn) / Wmin); Wn = Wmin; }
        } else {
          ds = k * 3. * pow(l_b, 2.) / W;
          Ln = L - ds; Wn = W;
        }
        const double dA = L * W - Ln * Wn;
        if (p->fl_style == KID_FL_STYLE_NEW_BERGS) {
          if (!calve_child(g, p, b, capacity, q, k, l_b, 0, 0u)) scalars[KID_S_ERROR_COUNT] += 1.;
          scalars[KID_S_NBERGS_CALVED_FL] += 1.;
        } else {
          const double dM_fl_bits = p->rho_bergs * T * dA;
          putf(b, KID_B_MASS_OF_FL_BITS, q, getf(b, KID_B_MASS_OF_FL_BITS, q) + dM_fl_bits);
          if (area != 0.) acc[(size_t)KID_A_FL_BITS_SRC * ncell + c] += dM_fl_bits / (p->dt * area) * ms;
        }
        if (Ln <= 0 || Wn <= 0) { scalars[KID_S_ERROR_COUNT] += 1.; /* FATAL IB:2649 */ }
        else {
          if (p->allow_bergs_to_roll) ko_rolling(p, &T, &Wn, &Ln);
          putf(b, KID_B_THICKNESS, q, T); putf(b, KID_B_WIDTH, q, Wn); putf(b, KID_B_LENGTH, q, Ln);
          putf(b, KID_B_MASS, q, Ln * Wn * T * p->rho_bergs);
        }
      }
    }
    if (getf(b, KID_B_MASS_OF_FL_BITS, q) * ms > p->new_berg_from_fl_bits_mass_thres) { /* IB:2663-2673 */
      const double k = floor(getf(b, KID_B_MASS_OF_FL_BITS, q) * ms / p->new_berg_from_fl_bits_mass_thres);
      if (!calve_child(g, p, b, capacity, q, k, l_b, 1, 1u)) scalars[KID_S_ERROR_COUNT] += 1.;
      scalars[KID_S_NBERGS_CALVED_FL] += 1.;
      if (area != 0.) acc[(size_t)KID_A_FL_BITS_SRC * ncell + c] -= k * p->new_berg_from_fl_bits_mass_thres / (p->dt * area);
    }
  }
  free(perm);
  g_fl_step += 1u;
}

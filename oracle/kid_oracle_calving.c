/* kid_oracle_calving.c -- CPU restatement (ORACLE, test infrastructure) of the calving source:
 *   the calving block of icebergs_run   /root/reference/src/icebergs.F90:5203-5231, 5397
 *   get_running_mean_calving            IB:5999-6038
 *   accumulate_calving                  IB:6153-6222
 *   calve_icebergs                      IB:6225-6402   (generate_id FW:4165-4179, ij_component_of_id FW:4224-4239)
 * mpp_sum over one rank is the identity.  tidal_drift>0 (random interpolation offsets, IB:6355-6359) is not restated.
 * PARITY UNPINNED: no recorded vector exists for this block.
 */
#include "kid_oracle.h"
#include <math.h>
#include <stdlib.h>

#define NI(g) ((g)->d.ied - (g)->d.isd + 1)
#define NJ(g) ((g)->d.jed - (g)->d.jsd + 1)
#define GIDX(g, i, j) ((size_t)((i) - (g)->d.isd) + (size_t)((j) - (g)->d.jsd) * (size_t)NI(g))

static void putf(kid_berg_soa *b, int f, int64_t k, double v) { if (b->f64[f]) b->f64[f][k] = v; }

int ko_calving(const ko_grid *g, const kid_params *p, const kid_calving_params *cp, const double *calving_in,
               const double *calving_hflx_in, ko_calving_state *s, kid_berg_soa *b, int64_t capacity, double *scalars) {
  const int isc = g->d.isc, iec = g->d.iec, jsc = g->d.jsc, jec = g->d.jec;
  const int nic = iec - isc + 1;
  const size_t ncell = (size_t)NI(g) * (size_t)NJ(g);
  const double *area = g->stat[KID_G_AREA], *msk = g->stat[KID_G_MSK], *lat = g->stat[KID_G_LAT], *lon = g->stat[KID_G_LON];
  double *calving = s->calving, *hflx = s->calving_hflx;
  for (int q = 0; q < KID_NCALV_SCALARS; ++q) scalars[q] = 0.;
  /* ---- icebergs_run IB:5203-5231 ---- */
  double tmpsum = 0.;
  for (int j = jsc; j <= jec; ++j) for (int i = isc; i <= iec; ++i) tmpsum += calving_in[(size_t)(i - isc) + (size_t)(j - jsc) * nic] * area[GIDX(g, i, j)];
  scalars[KID_CS_NET_CALVING_RECEIVED] = tmpsum * p->dt;                                           /* IB:5203-5204 */
  for (int j = jsc; j <= jec; ++j) for (int i = isc; i <= iec; ++i) {                             /* IB:5207-5212 */
    const size_t c = GIDX(g, i, j), a = (size_t)(i - isc) + (size_t)(j - jsc) * nic;
    hflx[c] = calving_hflx_in[a] * msk[c];
    calving[c] = calving_in[a] * msk[c];
  }
  if (cp->tau_calving > 0.) {                                                                     /* IB:5215-5219, 5999-6038 */
    if (!s->rmean_calving_initialized) { for (size_t c = 0; c < ncell; ++c) s->rmean_calving[c] = calving[c]; s->rmean_calving_initialized = 1; }
    if (!s->rmean_calving_hflx_initialized) { for (size_t c = 0; c < ncell; ++c) s->rmean_calving_hflx[c] = hflx[c]; s->rmean_calving_hflx_initialized = 1; }
    const double tau = cp->tau_calving / (365. * 24 * 60 * 60);   /* as written at IB:6020 */
    double alpha = tau / (tau + p->dt), beta;
    if (alpha != 0.) {
      if (alpha > 0.5) { beta = p->dt / (tau + p->dt); alpha = 1. - beta; } else beta = 1. - alpha;
      for (size_t c = 0; c < ncell; ++c) {
        s->rmean_calving[c] = beta * calving[c] + alpha * s->rmean_calving[c];
        s->rmean_calving_hflx[c] = beta * hflx[c] + alpha * s->rmean_calving_hflx[c];
        calving[c] = s->rmean_calving[c];
        hflx[c] = s->rmean_calving_hflx[c];
      }
    }
  }
  for (size_t c = 0; c < ncell; ++c) calving[c] = calving[c] * msk[c] * area[c];                  /* IB:5221, kg/s */
  tmpsum = 0.;
  for (int j = jsc; j <= jec; ++j) for (int i = isc; i <= iec; ++i) tmpsum += calving[GIDX(g, i, j)];
  scalars[KID_CS_NET_INCOMING_CALVING] = tmpsum * p->dt;                                           /* IB:5222-5223 */
  for (size_t c = 0; c < ncell; ++c) hflx[c] = hflx[c] * msk[c];                                  /* IB:5227 */
  tmpsum = 0.;
  for (int j = jsc; j <= jec; ++j) for (int i = isc; i <= iec; ++i) tmpsum += hflx[GIDX(g, i, j)] * area[GIDX(g, i, j)];
  scalars[KID_CS_NET_INCOMING_CALVING_HEAT] = tmpsum * p->dt;                                      /* IB:5230-5231 */
  /* ---- accumulate_calving IB:6153-6222 ---- */
  if (s->first_call && !cp->restarted) {                                                          /* IB:6171-6190 */
    s->first_call = 0;
    double st = 0.;
    for (int k = 0; k < KID_NCLASSES; ++k) for (int j = jsc; j <= jec; ++j) for (int i = isc; i <= iec; ++i) st += s->stored_ice[k * ncell + GIDX(g, i, j)];
    scalars[KID_CS_STORED_START] = st;
    for (int j = jsc; j <= jec; ++j) for (int i = isc; i <= iec; ++i) {
      const size_t c = GIDX(g, i, j);
      if (calving[c] != 0.) {
        double sum_ice = 0.;
        for (int k = 0; k < KID_NCLASSES; ++k) sum_ice += s->stored_ice[k * ncell + c];
        s->stored_heat[c] = sum_ice * hflx[c] * area[c] / calving[c];
      }
    }
    st = 0.;
    for (int j = jsc; j <= jec; ++j) for (int i = isc; i <= iec; ++i) st += s->stored_heat[GIDX(g, i, j)];
    scalars[KID_CS_STORED_HEAT_START] = st;
  }
  double remaining_dist_s = 1., remaining_dist_n = 1.;
  for (int k = 0; k < KID_NCLASSES; ++k) {                                                        /* IB:6192-6201 */
    for (size_t c = 0; c < ncell; ++c)
      s->stored_ice[k * ncell + c] = s->stored_ice[k * ncell + c] + p->dt * calving[c] * (lat[c] < 0. ? cp->distribution_s[k] : cp->distribution_n[k]);
    remaining_dist_s = remaining_dist_s - cp->distribution_s[k];
    remaining_dist_n = remaining_dist_n - cp->distribution_n[k];
  }
  double used = 0., heat_used = 0., unused = 0.;
  for (int j = jsc; j <= jec; ++j) for (int i = isc; i <= iec; ++i) {                             /* IB:6211 */
    const size_t c = GIDX(g, i, j);
    used += calving[c] * (1. - (lat[c] < 0. ? remaining_dist_s : remaining_dist_n));
  }
  scalars[KID_CS_NET_CALVING_USED] = used * p->dt;                                                 /* IB:6212 */
  double *tmp = (double *)malloc(ncell * sizeof(double));
  for (size_t c = 0; c < ncell; ++c) {                                                            /* IB:6214-6220 */
    const double rd = lat[c] < 0. ? remaining_dist_s : remaining_dist_n;
    calving[c] = calving[c] * rd;
    tmp[c] = p->dt * hflx[c] * area[c] * (1. - rd);
    s->stored_heat[c] = s->stored_heat[c] + tmp[c];
    hflx[c] = hflx[c] * rd;
  }
  for (int j = jsc; j <= jec; ++j) for (int i = isc; i <= iec; ++i) heat_used += tmp[GIDX(g, i, j)];
  free(tmp);
  scalars[KID_CS_NET_INCOMING_CALVING_HEAT_USED] = heat_used;                                      /* IB:6218 */
  for (int j = jsc; j <= jec; ++j) for (int i = isc; i <= iec; ++i) unused += calving[GIDX(g, i, j)];
  scalars[KID_CS_UNUSED_CALVING] = unused;                                                         /* IB:5397 */
  /* ---- calve_icebergs IB:6225-6402 ---- */
  for (size_t c = 0; c < KID_NCLASSES * ncell; ++c) s->real_calving[c] = 0.;                      /* IB:6253 */
  double calving_to_bergs = 0., heat_to_bergs = 0.;
  const int iNg = g->d.gni > 0 ? g->d.gni : iec - isc + 1, ij0 = g->d.gni > 0 ? g->d.gi0 + g->d.gni * g->d.gj0 : 0;   /* kid_grid_desc: a tile hashes the global cell */
  for (int k = 0; k < KID_NCLASSES; ++k) for (int j = jsc; j <= jec; ++j) for (int i = isc; i <= iec; ++i) {
    const size_t c = GIDX(g, i, j);
    double ddt = 0.;
    const int south = lat[c] < 0.;                                                                /* IB:6262-6270 */
    const double initial_mass = south ? p->initial_mass_s[k] : p->initial_mass_n[k];
    const double mass_scaling = south ? cp->mass_scaling_s[k] : cp->mass_scaling_n[k];
    const double initial_thickness = south ? cp->initial_thickness_s[k] : cp->initial_thickness_n[k];
    const double initial_width = south ? cp->initial_width_s[k] : cp->initial_width_n[k];
    const double initial_length = south ? cp->initial_length_s[k] : cp->initial_length_n[k];
    double *stored = s->stored_ice + k * ncell + c;
    while (*stored >= initial_mass * mass_scaling) {                                              /* IB:6273 */
      const double blon = 0.25 * ((lon[c] + lon[GIDX(g, i - 1, j - 1)]) + (lon[GIDX(g, i - 1, j)] + lon[GIDX(g, i, j - 1)]));
      const double blat = 0.25 * ((lat[c] + lat[GIDX(g, i - 1, j - 1)]) + (lat[GIDX(g, i - 1, j)] + lat[GIDX(g, i, j - 1)]));
      double xi, yj; int err = 0;
      const int lret = ko_pos_within_cell(g, p, blon, blat, i, j, &xi, &yj, &err);
      if (!lret) scalars[KID_CS_ERROR_COUNT] += 1.;                                                /* FATAL IB:6281 */
      if (b->n >= capacity) { scalars[KID_CS_ERROR_COUNT] += 1.; return -1; }
      const int64_t q = b->n;
      for (int f = 0; f < KID_NB_F64; ++f) putf(b, f, q, 0.0);   /* uvel..byn, *_prev, *_old, fl_k, bits, halo/static, mts and dem fields: all zero, IB:6294-6356 */
      putf(b, KID_B_LON, q, blon); putf(b, KID_B_LAT, q, blat);
      b->i32[KID_BI_INE][q] = i; b->i32[KID_BI_JNE][q] = j;
      putf(b, KID_B_XI, q, xi); putf(b, KID_B_YJ, q, yj);
      if (p->interactive_icebergs_on || p->footloose) { putf(b, KID_B_LON_OLD, q, blon); putf(b, KID_B_LAT_OLD, q, blat); }  /* IB:6297-6305 */
      putf(b, KID_B_MASS, q, initial_mass); putf(b, KID_B_THICKNESS, q, initial_thickness);
      putf(b, KID_B_WIDTH, q, initial_width); putf(b, KID_B_LENGTH, q, initial_length);
      putf(b, KID_B_START_LON, q, blon); putf(b, KID_B_START_LAT, q, blat);
      if (b->i32[KID_BI_START_YEAR]) b->i32[KID_BI_START_YEAR][q] = p->current_year;
      { /* generate_id FW:4165-4179 */
        int32_t cnt = 1;
        if (g->iceberg_counter) { g->iceberg_counter[c] += 1; cnt = g->iceberg_counter[c]; }
        const int32_t ij = i + (iNg * (j - 1)) + ij0;
        if (b->id) b->id[q] = (int64_t)cnt * ((int64_t)1 << 32) + (int64_t)ij;
      }
      putf(b, KID_B_START_DAY, q, p->current_yearday + ddt / 86400.);
      putf(b, KID_B_START_MASS, q, initial_mass); putf(b, KID_B_MASS_SCALING, q, mass_scaling);
      const double heat_density = s->stored_heat[c] / *stored;                                     /* IB:6329, J/kg */
      putf(b, KID_B_HEAT_DENSITY, q, heat_density);
      if (b->i32[KID_BI_N_BONDS]) b->i32[KID_BI_N_BONDS][q] = 0;
      if (b->i32[KID_BI_CONGLOM_ID]) b->i32[KID_BI_CONGLOM_ID][q] = 0;
      if (b->i32[KID_BI_ALIVE]) b->i32[KID_BI_ALIVE][q] = 1;
      if (!p->old_interp_flds_order) {                                                            /* IB:6353-6364 */
        double env[13];
        ko_interp_flds(g, p, blon, blat, i, j, xi, yj, env);
        for (int e = 0; e < 13; ++e) putf(b, KID_B_UO + e, q, env[e]);
      }
      b->n += 1;                                                                                  /* add_new_berg_to_list IB:6366 */
      const double calved_to_berg = initial_mass * mass_scaling;                                  /* IB:6367-6383 */
      const double heat_to_berg = calved_to_berg * heat_density;
      s->stored_heat[c] = s->stored_heat[c] - heat_to_berg;
      heat_to_bergs = heat_to_bergs + heat_to_berg;
      *stored = *stored - calved_to_berg;
      calving_to_bergs = calving_to_bergs + calved_to_berg;
      s->real_calving[k * ncell + c] = s->real_calving[k * ncell + c] + calved_to_berg / p->dt;
      ddt = ddt - p->dt * 2. / 17.;
      scalars[KID_CS_NBERGS_CALVED] += 1.;
      scalars[(south ? KID_CS_NBERGS_CALVED_BY_CLASS_S : KID_CS_NBERGS_CALVED_BY_CLASS_N) + k] += 1.;
    }
  }
  scalars[KID_CS_NET_CALVING_TO_BERGS] = calving_to_bergs;                                         /* IB:6399 */
  scalars[KID_CS_NET_HEAT_TO_BERGS] = heat_to_bergs;                                               /* IB:6400 */
  return 0;
}

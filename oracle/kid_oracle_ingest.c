/* kid_oracle_ingest.c -- CPU restatement (ORACLE, test infrastructure) of the forcing ingest block of icebergs_run,
 * /root/reference/src/icebergs.F90:5236-5383, and invert_tau_for_du IB:8272-8296.
 *
 * mpp_update_domains (FMS, not in the reference tree) is replaced by its single-rank meaning: nothing on a closed
 * domain; with `cyclic_x` every halo column takes, row by row, what the column one zonal period away holds.
 * add_iceberg_thickness_to_SSH (IB:5330-5337) is not restated.  PARITY UNPINNED: no recorded vector exists for this block.
 */
#include "kid_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

#define NI(g) ((g)->d.ied - (g)->d.isd + 1)
#define NJ(g) ((g)->d.jed - (g)->d.jsd + 1)
#define IX(g, i, j) ((size_t)((i) - (g)->d.isd) + (size_t)((j) - (g)->d.jsd) * (size_t)NI(g))
#define MSK(g, i, j) ((g)->stat[KID_G_MSK][IX(g, i, j)])

static void wrap_x(const ko_grid *g, double *f) { /* the single-rank mpp_update_domains on a zonally cyclic domain */
  const int nic = g->d.iec - g->d.isc + 1;
  for (int j = g->d.jsd; j <= g->d.jed; ++j) {
    for (int i = g->d.isd; i < g->d.isc; ++i) f[IX(g, i, j)] = f[IX(g, i + nic, j)];
    for (int i = g->d.iec + 1; i <= g->d.ied; ++i) f[IX(g, i, j)] = f[IX(g, i - nic, j)];
  }
}
static void update(const ko_grid *g, const kid_forcing_in *in, double *f) { if (in->cyclic_x) wrap_x(g, f); }
static double mask4(const ko_grid *g, int i, int j) { /* IB:5254 */
  return fmin(fmin(MSK(g, i, j), MSK(g, i + 1, j)), fmin(MSK(g, i, j + 1), MSK(g, i + 1, j + 1)));
}
/* element (a, b), 1-based Fortran indices, of an (n1, n2) column-major array */
#define A2(p, n1, a, b) ((p)[(size_t)((a) - 1) + (size_t)((b) - 1) * (size_t)(n1)])

/* out[KID_NFORCING]: planes over the data domain; they are read-modify-write (cells the block does not touch keep
 * their previous content, as grd%* do) */
int ko_ingest_forcing(const ko_grid *g, const kid_forcing_in *in, double *const out[KID_NFORCING]) {
  const int isc = g->d.isc, iec = g->d.iec, jsc = g->d.jsc, jec = g->d.jec;
  const int nic = iec - isc + 1, njc = jec - jsc + 1;
  double *uo = out[KID_F_UO], *vo = out[KID_F_VO], *ui = out[KID_F_UI], *vi = out[KID_F_VI], *ua = out[KID_F_UA], *va = out[KID_F_VA];
  double *ssh = out[KID_F_SSH], *sst = out[KID_F_SST], *sss = out[KID_F_SSS], *cn = out[KID_F_CN], *hi = out[KID_F_HI];
  if (in->vel_stagger == KID_BGRID_NE) { /* IB:5236-5243 */
    if (in->u_ni != nic + 2 || in->u_nj != njc + 2 || in->v_ni != nic + 2 || in->v_nj != njc + 2) return -1;
    for (int j = jsc - 1; j <= jec + 1; ++j) for (int i = isc - 1; i <= iec + 1; ++i) {
      const int a = i - (isc - 1) + 1, b = j - (jsc - 1) + 1;
      uo[IX(g, i, j)] = A2(in->uo, in->u_ni, a, b); vo[IX(g, i, j)] = A2(in->vo, in->v_ni, a, b);
      ui[IX(g, i, j)] = A2(in->ui, in->u_ni, a, b); vi[IX(g, i, j)] = A2(in->vi, in->v_ni, a, b);
    }
    update(g, in, uo); update(g, in, vo); update(g, in, ui); update(g, in, vi);
  } else if (in->vel_stagger == KID_CGRID_NE) { /* IB:5244-5259 */
    const int Iu_off = (in->u_ni - (iec - isc)) / 2 - isc + 1, ju_off = (in->u_nj - (jec - jsc)) / 2 - jsc + 1;
    const int iv_off = (in->v_ni - (iec - isc)) / 2 - isc + 1, Jv_off = (in->v_nj - (jec - jsc)) / 2 - jsc + 1;
    for (int i = isc - 1; i <= iec; ++i) for (int j = jsc - 1; j <= jec; ++j) {
      const int Iu = i + Iu_off, ju = j + ju_off, iv = i + iv_off, Jv = j + Jv_off;
      if (Iu < 1 || Iu > in->u_ni || ju < 1 || ju + 1 > in->u_nj || iv < 1 || iv + 1 > in->v_ni || Jv < 1 || Jv > in->v_nj) return -1;
      const double mask = mask4(g, i, j);
      uo[IX(g, i, j)] = mask * 0.5 * (A2(in->uo, in->u_ni, Iu, ju) + A2(in->uo, in->u_ni, Iu, ju + 1));
      ui[IX(g, i, j)] = mask * 0.5 * (A2(in->ui, in->u_ni, Iu, ju) + A2(in->ui, in->u_ni, Iu, ju + 1));
      vo[IX(g, i, j)] = mask * 0.5 * (A2(in->vo, in->v_ni, iv, Jv) + A2(in->vo, in->v_ni, iv + 1, Jv));
      vi[IX(g, i, j)] = mask * 0.5 * (A2(in->vi, in->v_ni, iv, Jv) + A2(in->vi, in->v_ni, iv + 1, Jv));
    }
  } else return -1;
  if (in->stress_stagger == KID_BGRID_NE) { /* IB:5264-5267 */
    if (in->taux_ni != nic || in->taux_nj != njc || in->tauy_ni != nic || in->tauy_nj != njc) return -1;
    for (int j = jsc; j <= jec; ++j) for (int i = isc; i <= iec; ++i) {
      ua[IX(g, i, j)] = A2(in->tauxa, nic, i - isc + 1, j - jsc + 1); va[IX(g, i, j)] = A2(in->tauya, nic, i - isc + 1, j - jsc + 1);
    }
  } else if (in->stress_stagger == KID_CGRID_NE || in->stress_stagger == KID_AGRID) { /* IB:5268-5293, 5294-5313 */
    const size_t ncell = (size_t)NI(g) * (size_t)NJ(g);
    int Iu_off = -isc + 1, ju_off = -jsc + 1, iv_off = -isc + 1, Jv_off = -jsc + 1;
    if (in->stress_stagger == KID_CGRID_NE) {
      Iu_off = (in->taux_ni - (iec - isc)) / 2 - isc + 1; ju_off = (in->taux_nj - (jec - jsc)) / 2 - jsc + 1;
      iv_off = (in->tauy_ni - (iec - isc)) / 2 - isc + 1; Jv_off = (in->tauy_nj - (jec - jsc)) / 2 - jsc + 1;
      if (isc + Iu_off < 1 || iec + Iu_off > in->taux_ni || jsc + ju_off < 1 || jec + ju_off > in->taux_nj) return -1;
      if (isc + iv_off < 1 || iec + iv_off > in->tauy_ni || jsc + Jv_off < 1 || jec + Jv_off > in->tauy_nj) return -1;
    } else if (in->taux_ni != nic || in->taux_nj != njc || in->tauy_ni != nic || in->tauy_nj != njc) return -1;
    double *ut = (double *)calloc(ncell, sizeof(double)), *vt = (double *)calloc(ncell, sizeof(double)); /* halos stay 0, IB:5276 */
    for (int i = isc; i <= iec; ++i) for (int j = jsc; j <= jec; ++j) {
      ut[IX(g, i, j)] = A2(in->tauxa, in->taux_ni, i + Iu_off, j + ju_off); vt[IX(g, i, j)] = A2(in->tauya, in->tauy_ni, i + iv_off, j + Jv_off);
    }
    update(g, in, ut); update(g, in, vt);
    for (int i = isc - 1; i <= iec; ++i) for (int j = jsc - 1; j <= jec; ++j) {
      const double mask = mask4(g, i, j);
      if (in->stress_stagger == KID_CGRID_NE) {
        ua[IX(g, i, j)] = mask * 0.5 * (ut[IX(g, i, j)] + ut[IX(g, i, j + 1)]);
        va[IX(g, i, j)] = mask * 0.5 * (vt[IX(g, i, j)] + vt[IX(g, i + 1, j)]);
      } else {
        ua[IX(g, i, j)] = mask * 0.25 * ((ut[IX(g, i, j)] + ut[IX(g, i + 1, j + 1)]) + (ut[IX(g, i + 1, j)] + ut[IX(g, i, j + 1)]));
        va[IX(g, i, j)] = mask * 0.25 * ((vt[IX(g, i, j)] + vt[IX(g, i + 1, j + 1)]) + (vt[IX(g, i + 1, j)] + vt[IX(g, i, j + 1)]));
      }
    }
    free(ut); free(vt);
  } else return -1;
  update(g, in, uo); update(g, in, vo); update(g, in, ui); update(g, in, vi); /* IB:5318-5319 */
  if (!in->tau_is_velocity) { /* invert_tau_for_du IB:8272-8296, whole arrays */
    const double cd = 0.0015;
    for (int j = g->d.jsd; j <= g->d.jed; ++j) for (int i = g->d.isd; i <= g->d.ied; ++i) {
      const double u = ua[IX(g, i, j)], v = va[IX(g, i, j)];
      const double tau2 = u * u + v * v, cddvmod = sqrt(cd * sqrt(tau2));
      if (cddvmod != 0.) { ua[IX(g, i, j)] = u / cddvmod; va[IX(g, i, j)] = v / cddvmod; } else { ua[IX(g, i, j)] = 0.; va[IX(g, i, j)] = 0.; }
    }
  }
  update(g, in, ua); update(g, in, va); /* IB:5326 */
  for (int j = jsc - 1; j <= jec + 1; ++j) for (int i = isc - 1; i <= iec + 1; ++i) { /* IB:5329, 5339, 5348-5351 */
    const int a = i - (isc - 1) + 1, b = j - (jsc - 1) + 1;
    ssh[IX(g, i, j)] = A2(in->ssh, nic + 2, a, b); cn[IX(g, i, j)] = A2(in->cn, nic + 2, a, b); hi[IX(g, i, j)] = A2(in->hi, nic + 2, a, b);
  }
  update(g, in, ssh); update(g, in, cn); update(g, in, hi);
  double max_SST = -HUGE_VAL; /* IB:5340-5346 */
  for (int j = jsc; j <= jec; ++j) for (int i = isc; i <= iec; ++i) max_SST = fmax(max_SST, A2(in->sst, nic, i - isc + 1, j - jsc + 1) * MSK(g, i, j));
  for (int j = jsc; j <= jec; ++j) for (int i = isc; i <= iec; ++i) {
    const double t = A2(in->sst, nic, i - isc + 1, j - jsc + 1);
    sst[IX(g, i, j)] = (max_SST > 120.0) ? t - 273.15 : t;
    sss[IX(g, i, j)] = in->sss ? A2(in->sss, nic, i - isc + 1, j - jsc + 1) : -1.0; /* IB:5354-5361 */
  }
  update(g, in, sst);
  for (int i = g->d.isd; i <= g->d.ied; ++i) for (int j = g->d.jsd; j <= g->d.jed; ++j) { /* IB:5364-5383 */
    const size_t c = IX(g, i, j);
    if (MSK(g, i, j) < 0.5) { ua[c] = 0.; va[c] = 0.; uo[c] = 0.; vo[c] = 0.; ui[c] = 0.; vi[c] = 0.; sst[c] = 0.; sss[c] = 0.; cn[c] = 0.; hi[c] = 0.; }
    double *const fl[10] = {ua, va, uo, vo, ui, vi, sst, sss, cn, hi};
    for (int q = 0; q < 10; ++q) if (fl[q][c] != fl[q][c]) fl[q][c] = 0.;
  }
  return 0;
}

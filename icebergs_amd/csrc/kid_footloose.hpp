// kid_footloose.hpp -- footloose calving on the SoA (gfx950), one lane per berg.
//   footloose_calving  IB:2503-2734 (get_footloose_displacement IB:2688-2732)   calve_fl_icebergs  IB:6405-6569
//   find_cell  FW:6011-6040      generate_id  FW:4165-4179
// A calving event is rare (a parent sheds a child every few hundred steps), so this is a short streaming kernel
// between the evolve and the thermodynamics launches.  New bergs are appended behind the population through an
// atomic cursor.  Their ids come from the per-cell counter of the parent's cell (generate_id FW:4165-4179), which the reference
// hands out in the order its loop meets the events: cells j outer / i inner, a cell's bergs in list order (`inorder`), a
// berg's calving event before its new-berg-from-bits event.  Here a child first gets a provisional (negative) id naming its
// parent's row and the event, and fl_assign_ids_* (below, launched once the pass has ended) gives the children of every cell
// their counter values in exactly that order -- the same ids whatever the schedule of the lanes.  displace_fl_bergs: the child's place on the parent's perimeter comes from the counter-
// based generator of include/kid_rng.h, keyed by (seed, parent id, footloose step, draw) -- the reference draws from
// FMS's sequential stream in traversal order, which has no counterpart on a GPU; what follows the number (side and
// offset, metres -> degrees, find_cell and its corner / grounded-cell fall-backs, pos_within_cell) is the reference's.
// The "new berg from FL bits" branch hands calve_fl_icebergs the local l_b left by the last ELIGIBLE berg of the loop (IB:2573,
// 2667) -- a stale value for a footloose child (fl_k < 0) that holds bits -- but with berg_from_bits present that dummy argument
// is never read (IB:6488-6497 take the dimensions from fl_bits_dimensions; l_b is read at IB:6499-6501 only): nothing to reproduce.
#pragma once
#include "kid_device.hpp"
#include "kid_thermo.hpp"
#define KID_RNG_FN __host__ __device__ static inline
#include "../../include/kid_rng.h"

namespace kid {

struct FlChildCtx {
  int *cursor;          // number of children appended in this launch
  int32_t *counter;     // grd%iceberg_counter_grd
  long long n, capacity;
  int iNg;              // zonal size of the global grid (ij_component_of_id, FW:4227-4240)
  unsigned step;        // the footloose step: third counter word of the random number generator (kid_rng.h)
};

// find_cell FW:6011-6040: the structured-grid guess, then a scan of the computational domain (one lane; reached when the
// guess fails, i.e. on grids that are not uniform in index space)
__device__ __noinline__ bool fl_find_cell(const DevGrid &g, double x, double y, int &oi, int &oj) {
  const GeoRec a = g.geo[g.idx(g.isd, g.jsd)], c = g.geo[g.idx(g.isd + 1, g.jsd + 1)];
  oi = (int)floor((x - a.lon) / (c.lon - a.lon)) + g.isd + 1;
  oj = (int)floor((y - a.lat) / (c.lat - a.lat)) + g.jsd + 1;
  if (oi > g.isc - 1 && oi < g.iec + 1 && oj > g.jsc - 1 && oj < g.jec + 1)
    if (is_point_in_cell<false>(g, GlbCell{g, g.idx(oi, oj)}.corners(), x, y)) return true;
  oi = -999; oj = -999;
  for (int j = g.jsc; j <= g.jec; ++j)
    for (int i = g.isc; i <= g.iec; ++i)
      if (is_point_in_cell<false>(g, GlbCell{g, g.idx(i, j)}.corners(), x, y)) { oi = i; oj = j; return true; }
  return false;
}

// get_footloose_displacement IB:2688-2732
__device__ __noinline__ void fl_displacement(const DevGrid &g, const kid_params &p, double rn, double lon, double lat, double length, double width,
                                             double &fl_disp_x, double &fl_disp_y) {
  double fx, fy, interp_loc;
  if (rn < 0.25) { interp_loc = 4. * rn; fx = length * (interp_loc - 0.5); fy = 0.5 * width; }                 // north side
  else if (rn < 0.5) { interp_loc = 4. * (rn - 0.25); fx = 0.5 * length; fy = width * (interp_loc - 0.5); }   // east side
  else if (rn < 0.75) { interp_loc = 4. * (rn - 0.5); fx = length * (interp_loc - 0.5); fy = -0.5 * width; }  // south side
  else { interp_loc = 4. * (rn - 0.75); fx = -0.5 * length; fy = 0.5 * width * (interp_loc - 0.5); }          // west side (the 0.5 is the reference's, IB:2714)
  if (g.latlon) {
    const bool on_tang = (lat > 89.);
    double lon1 = lon, lat1 = lat, x1 = 0., y1 = 0.;
    if (on_tang) rotpos_to_tang(p, lon1, lat1, x1, y1);
    const double dxdl1 = (180. / p.pi) / (p.Rearth * cos(lat1 * (p.pi / 180.))), dydl = (180. / p.pi) / p.Rearth;   // IB:462-477
    if (on_tang) {
      double xdot2, ydot2;
      rotvec_to_tang(p, lon1, fx, fy, xdot2, ydot2);
      x1 = x1 + xdot2; y1 = y1 + ydot2;
      rotpos_from_tang(p, x1, y1, lon1, lat1);
    } else { lon1 = lon1 + fx * dxdl1; lat1 = lat1 + fy * dydl; }
    fx = lon1 - lon; fy = lat1 - lat;
  }
  fl_disp_x = fx; fl_disp_y = fy;
}

// calve_fl_icebergs IB:6405-6569 in two halves.
// (1) fl_calve_event, at the moment of the event, inline in the rare branch of whoever runs footloose_core: reserves the child's
//     row and writes what depends on the parent's state at this very moment -- the child's size, masses and provisional id, the
//     parent's share of the footloose bits for a berg made from bits -- and leaves the parent's length and width (which the
//     thermodynamics of the same launch will change) in the child's start_lon / start_lat for the second half.
// (2) fl_place_children_kernel, one lane per child once the pass has ended and the host knows how many there are: position
//     (the random number of the event, get_footloose_displacement, find_cell and its fall-backs, pos_within_cell), the members
//     copied from the parent -- none of which the thermodynamics touches -- and the constants.
// As ONE out-of-line function in the middle of the fused step this cost the hot build of the footloose profile 100 registers:
// everything live across a call has to sit in callee-saved registers, which are half of the file (252 registers + 240 bytes of
// scratch per lane with the call, 155 without it).  Returns false if the SoA is full.
constexpr int KID_FL_STASH_LEN = KID_B_START_LON, KID_FL_STASH_WID = KID_B_START_LAT;
template <class BP>
__device__ __forceinline__ bool fl_calve_event(const kid_params &p, const BP &b, const FlChildCtx &cx, long long pk, double k, double l_b, bool from_bits) {
  const int slot = atomicAdd(cx.cursor, 1);
  const long long c = cx.n + slot;
  if (c >= cx.capacity) return false;
  b.f[KID_FL_STASH_LEN][c] = b.f[KID_B_LENGTH][pk]; b.f[KID_FL_STASH_WID][c] = b.f[KID_B_WIDTH][pk];
  const double pms = b.f[KID_B_MASS_SCALING][pk];
  if (from_bits) {  // IB:6488-6497
    double Lfl, Wfl, Tfl;
    fl_bits_dimensions_inl<0>(p, b.f[KID_B_THICKNESS][pk], Lfl, Wfl, Tfl);
    const double cmass = Tfl * Lfl * Wfl * p.rho_bergs;
    const double cms = k * p.new_berg_from_fl_bits_mass_thres / cmass;
    b.f[KID_B_LENGTH][c] = Lfl; b.f[KID_B_WIDTH][c] = Wfl; b.f[KID_B_THICKNESS][c] = Tfl;
    b.f[KID_B_MASS][c] = cmass; b.f[KID_B_MASS_SCALING][c] = cms;
    const double percent_fl = (cmass * cms) / (b.f[KID_B_MASS_OF_FL_BITS][pk] * pms);
    b.f[KID_B_MASS_OF_BITS][c] = (percent_fl * b.f[KID_B_MASS_OF_FL_BERGY_BITS][pk] * pms) / cms;
    b.f[KID_B_MASS_OF_FL_BERGY_BITS][pk] = (1 - percent_fl) * b.f[KID_B_MASS_OF_FL_BERGY_BITS][pk];
    b.f[KID_B_MASS_OF_FL_BITS][pk] = b.f[KID_B_MASS_OF_FL_BITS][pk] - k * p.new_berg_from_fl_bits_mass_thres / pms;
  } else {          // IB:6499-6504
    const double len = l_b * 3., wid = l_b, thick = b.f[KID_B_THICKNESS][pk];
    b.f[KID_B_LENGTH][c] = len; b.f[KID_B_WIDTH][c] = wid; b.f[KID_B_THICKNESS][c] = thick;
    b.f[KID_B_MASS][c] = wid * len * thick * p.rho_bergs;
    b.f[KID_B_MASS_SCALING][c] = pms * k;
    b.f[KID_B_MASS_OF_BITS][c] = 0.0;
  }
  // provisional id: -(1 + 2 * parent row + event); fl_assign_ids_* replaces it by generate_id's value (FW:4165-4179)
  b.id[c] = -(((int64_t)pk << 1 | (from_bits ? 1 : 0)) + 1);
  return true;
}
template <class BP>
__device__ __forceinline__ void fl_place_child(const DevGrid &g, const kid_params &p, const BP &b, unsigned step, long long c) {
  const int64_t prov = -b.id[c] - 1;
  const long long pk = (long long)(prov >> 1);
  const unsigned draw = (unsigned)(prov & 1);   // 0 for the calving block (IB:2631), 1 for the new-berg-from-bits block (IB:2664)
  const double plen = b.f[KID_FL_STASH_LEN][c], pwid = b.f[KID_FL_STASH_WID][c];
#pragma unroll 1
  for (int f = 0; f < KID_NB_F64; ++f)
    if (f != KID_B_LENGTH && f != KID_B_WIDTH && f != KID_B_THICKNESS && f != KID_B_MASS && f != KID_B_MASS_SCALING && f != KID_B_MASS_OF_BITS) b.f[f][c] = 0.0;
  const int pi = b.i[KID_BI_INE][pk], pj = b.i[KID_BI_JNE][pk];
  const double plon = b.f[KID_B_LON][pk], plat = b.f[KID_B_LAT][pk];
  double fl_disp_x = 0.0, fl_disp_y = 0.0;
  bool displace = p.displace_fl_bergs != 0;
  if (displace) {
    // IB:2631 / 2664: a fresh number per event, or the one number of the run (fl_init_child_xy_by_pe)
    const double rn = p.fl_init_child_xy_by_pe ? kid_fl_uniform((uint32_t)p.fl_rng_seed, 0, 0u, 0u)
                                               : kid_fl_uniform((uint32_t)p.fl_rng_seed, b.id[pk], step, draw);
    fl_displacement(g, p, rn, plon, plat, plen, pwid, fl_disp_x, fl_disp_y);
    double clon = plon + fl_disp_x, clat = plat + fl_disp_y;   // IB:6433-6435
    int ci, cj;
    bool lres = fl_find_cell(g, clon, clat, ci, cj);
    if (!lres) {  // not in the computational domain: try the corners (IB:6438-6467; metres added to degrees as written)
      clon = plon - 0.5 * plen; clat = plat - 0.5 * pwid; lres = fl_find_cell(g, clon, clat, ci, cj);
      if (!lres) { clon = plon - 0.5 * plen; clat = plat + 0.5 * pwid; lres = fl_find_cell(g, clon, clat, ci, cj); }
      if (!lres) { clon = plon + 0.5 * plen; clat = plat + 0.5 * pwid; lres = fl_find_cell(g, clon, clat, ci, cj); }
      if (!lres) { clon = plon + 0.5 * plen; clat = plat - 0.5 * pwid; lres = fl_find_cell(g, clon, clat, ci, cj); }
      if (!lres) { fl_disp_x = 0.0; fl_disp_y = 0.0; displace = false; }
      else { fl_disp_x = plon - clon; fl_disp_y = plat - clat; }   // (sign as in the reference, IB:6466)
    }
    if (displace) {
      if (g.geo[g.idx(ci, cj)].area == 0.) { fl_disp_x = 0.0; fl_disp_y = 0.0; displace = false; }   // grounded cell IB:6471-6472
      else {
        double xi, yj; int err = 0; bool bail = false;
        (void)pos_within_cell<false>(g, p, GlbCell{g, g.idx(ci, cj)}, clon, clat, ci, cj, xi, yj, err, bail);
        b.f[KID_B_LON][c] = clon; b.f[KID_B_LAT][c] = clat; b.f[KID_B_XI][c] = xi; b.f[KID_B_YJ][c] = yj;
        b.i[KID_BI_INE][c] = ci; b.i[KID_BI_JNE][c] = cj;
      }
    }
  }
  if (!displace) {  // position = the parent's (IB:6479-6486)
    b.f[KID_B_LON][c] = plon; b.f[KID_B_LAT][c] = plat;
    b.f[KID_B_XI][c] = b.f[KID_B_XI][pk]; b.f[KID_B_YJ][c] = b.f[KID_B_YJ][pk];
    b.i[KID_BI_INE][c] = pi; b.i[KID_BI_JNE][c] = pj;
  }
  b.f[KID_B_START_LON][c] = b.f[KID_B_LON][c]; b.f[KID_B_START_LAT][c] = b.f[KID_B_LAT][c];
  b.f[KID_B_LON_OLD][c] = b.f[KID_B_LON_OLD][pk] + fl_disp_x; b.f[KID_B_LAT_OLD][c] = b.f[KID_B_LAT_OLD][pk] + fl_disp_y;
  b.f[KID_B_START_DAY][c] = p.current_yearday;
  b.f[KID_B_MASS_OF_FL_BITS][c] = 0.0; b.f[KID_B_MASS_OF_FL_BERGY_BITS][c] = 0.0;
  b.f[KID_B_FL_K][c] = -1.0;
  b.i[KID_BI_START_YEAR][c] = p.current_year;
  b.f[KID_B_HALO_BERG][c] = 0.0;
  const int same[] = {KID_B_START_MASS, KID_B_UVEL, KID_B_VVEL, KID_B_AXN, KID_B_AYN, KID_B_BXN, KID_B_BYN,
                      KID_B_UVEL_PREV, KID_B_VVEL_PREV, KID_B_UVEL_OLD, KID_B_VVEL_OLD, KID_B_HEAT_DENSITY,
                      KID_B_STATIC_BERG, KID_B_UO, KID_B_VO, KID_B_UI, KID_B_VI, KID_B_UA, KID_B_VA, KID_B_SSH_X,
                      KID_B_SSH_Y, KID_B_SST, KID_B_SSS, KID_B_CN, KID_B_HI, KID_B_OD};
#pragma unroll 1
  for (unsigned q = 0; q < sizeof(same) / sizeof(same[0]); ++q) b.f[same[q]][c] = b.f[same[q]][pk];
  b.i[KID_BI_N_BONDS][c] = 0;
  b.i[KID_BI_ALIVE][c] = 1;
}
template <class BP>
__global__ void __launch_bounds__(64) fl_place_children_kernel(const DevGrid g, const kid_params *__restrict__ pp, const BP *__restrict__ bt, long long n_old, int m, unsigned step) {
  const int q = (int)(blockIdx.x * 64u + threadIdx.x);
  if (q < m) fl_place_child(g, *pp, *bt, step, n_old + q);
}

// footloose_calving for one berg (IB:2503-2734).  The berg's own values come in and go out through the arguments (the fused
// step holds them in registers: no round trip through memory between its evolve and its thermodynamics); whatever
// changes is ALSO written to the berg's row where the reference changes it, because fl_calve_event reads the parent's row.
// bits_rows_touched: fl_calve_event(from_bits) has rewritten mass_of_fl_bits and mass_of_fl_bergy_bits of the row.
template <class BP>
__device__ __forceinline__ void footloose_core(const DevGrid &g, const kid_params &p, const BP &b, const FlChildCtx &cx, long long q, int i, int j, double area,
                                               double ms, double static_berg, double &M, double &T, double &W, double &L, double &flk, double &bits,
                                               bool &bits_rows_touched, double *acc, size_t ncell, double *scal) {
  if (i < g.isc || i > g.iec || j < g.jsc || j > g.jec) return;  // computational domain only, IB:2554
  const int c = g.idx(i, j);
  // constants IB:2538-2547
  const double e1 = g.fl_e1, drho = RHO_SEAWATER - p.rho_bergs, sigmay = p.fl_strength * 1000;  // exp(pi/4) from the host
  const double lfootparam = e1 * RHO_SEAWATER * sigmay / (6 * p.rho_bergs * GRAVITY * drho);
  const double l_c = p.pi / (2. * sqrt(2.)), lw_c = FL_LW_C, B_c = p.fl_youngs / (12. * (1. - 0.3 * 0.3));
  const double l_w = kid_root4(lw_c * B_c * kid_cube(T));
  const double l_b = l_c * l_w;
  double nerr = 0., ncalved = 0.;
  if (!(static_berg == 1 || flk < 0)) {
    const double l_b3 = 3 * l_b;
    double cc = ceil((L - l_b3) / l_b3); const double Lmin = L - cc * l_b3;
    cc = ceil((W - l_b3) / l_b3); const double Wmin = W - cc * l_b3;
    const double max_k = dmax(floor((L * W - Lmin * Wmin) / (l_b3 * l_b)), 0);
    double k = 0;
    if (max_k != 0) {
      const double foot_l = lfootparam * T / l_w;
      const double foot_area = foot_l * l_b3;
      k = floor(flk / foot_area);
      if (k > max_k) k = max_k;
      if (k != 0) { flk = flk - k * foot_area; b.f[KID_B_FL_K][q] = flk; }   // (k = 0 leaves fl_k as it is, bit for bit)
    }
    if (k > 0) {
      double ds, Ln, Wn;
      if (cc > 0) {
        ds = 0.5 * ((L + W) - sqrt(((L + W) * (L + W)) - 4. * (l_b3 * l_b * k)));
        Ln = L - ds; Wn = W - ds;
        if (Wn < Wmin) { Ln = Ln * (1 - (Wmin - Wn) / Wmin); Wn = Wmin; }
      } else {
        ds = k * 3. * (l_b * l_b) / W;
        Ln = L - ds; Wn = W;
      }
      const double dA = L * W - Ln * Wn;
      if (p.fl_style == KID_FL_STYLE_NEW_BERGS) {
        if (!fl_calve_event(p, b, cx, q, k, l_b, false)) nerr += 1.;
        ncalved += 1.;
      } else {
        const double dM_fl_bits = p.rho_bergs * T * dA;
        bits = bits + dM_fl_bits;
        b.f[KID_B_MASS_OF_FL_BITS][q] = bits;
        if (area != 0.) unsafeAtomicAdd(acc + (size_t)KID_A_FL_BITS_SRC * ncell + c, dM_fl_bits / (p.dt * area) * ms);
      }
      if (Ln <= 0 || Wn <= 0) nerr += 1.;  // FATAL IB:2649
      else {
        if (p.allow_bergs_to_roll) rolling(p, T, Wn, Ln);
        W = Wn; L = Ln; M = Ln * Wn * T * p.rho_bergs;
        b.f[KID_B_THICKNESS][q] = T; b.f[KID_B_WIDTH][q] = Wn; b.f[KID_B_LENGTH][q] = Ln;
        b.f[KID_B_MASS][q] = M;
      }
    }
  }
  if (bits * ms > p.new_berg_from_fl_bits_mass_thres) {  // IB:2663-2673
    const double k = floor(bits * ms / p.new_berg_from_fl_bits_mass_thres);
    if (!fl_calve_event(p, b, cx, q, k, l_b, true)) nerr += 1.;
    bits_rows_touched = true;
    ncalved += 1.;
    if (area != 0.) unsafeAtomicAdd(acc + (size_t)KID_A_FL_BITS_SRC * ncell + c, -(k * p.new_berg_from_fl_bits_mass_thres / (p.dt * area)));
  }
  if (ncalved != 0.) unsafeAtomicAdd(scal + KID_S_NBERGS_CALVED_FL, ncalved);
  if (nerr != 0.) unsafeAtomicAdd(scal + KID_S_ERROR_COUNT, nerr);
}
// the berg's row as the only state (footloose_kernel, the general build)
template <class BP>
__device__ __forceinline__ void footloose_one(const DevGrid &g, const kid_params &p, const BP &b, const FlChildCtx &cx, long long q,
                                              double *acc, size_t ncell, double *scal) {
  const int i = b.i[KID_BI_INE][q], j = b.i[KID_BI_JNE][q];
  if (i < g.isc || i > g.iec || j < g.jsc || j > g.jec) return;
  double M = b.f[KID_B_MASS][q], T = b.f[KID_B_THICKNESS][q], W = b.f[KID_B_WIDTH][q], L = b.f[KID_B_LENGTH][q];
  double flk = b.f[KID_B_FL_K][q], bits = b.f[KID_B_MASS_OF_FL_BITS][q];
  bool touched = false;
  footloose_core(g, p, b, cx, q, i, j, g.geo[g.idx(i, j)].area, b.f[KID_B_MASS_SCALING][q], b.f[KID_B_STATIC_BERG][q], M, T, W, L, flk, bits, touched, acc, ncell, scal);
}


// ---- ids of the children of one footloose pass, in the reference's order ---------------------------------------------------
// rows [n_old, n_old + m) hold the children with provisional ids.  Events per pass are few (~1e-4 of the bergs), events per
// cell one, rarely two: every child pushes itself on its parent cell's list (head[cell], next[]), then counts the children of
// that list that the reference's loop meets before it -- parents compared by the `inorder` keys (FW:4318-4359), equal keys by
// the parent's id, one parent's two events by the event -- and takes counter + 1 + that count.
struct FlIdCtx { int32_t *head, *next; int64_t *newid; int32_t *counter; long long n_old; int m; int iNg, ij0; };   // iNg, ij0: kid_grid_desc gni / gi0 / gj0
__device__ __forceinline__ void fl_event_of(const int64_t prov, long long &parent, int &ev) {
  const int64_t v = -prov - 1;
  parent = (long long)(v >> 1); ev = (int)(v & 1);
}
template <class BP>
__device__ __forceinline__ bool fl_event_before(const BP &b, long long pa, int ea, long long pb, int eb) {   // a strictly before b
  if (pa == pb) return ea < eb;
  const int ya = b.i[KID_BI_START_YEAR][pa], yb = b.i[KID_BI_START_YEAR][pb];
  if (ya != yb) return ya < yb;
  const int keys[4] = {KID_B_START_DAY, KID_B_START_MASS, KID_B_START_LON, KID_B_START_LAT};
  for (int q = 0; q < 4; ++q) {
    const double va = b.f[keys[q]][pa], vb = b.f[keys[q]][pb];
    if (va < vb) return true;
    if (va > vb) return false;
  }
  return b.id[pa] < b.id[pb];
}
template <class BP>
__global__ void __launch_bounds__(256) fl_assign_ids_push(const DevGrid g, const BP *bt, const FlIdCtx x) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= x.m) return;
  const BP &b = *bt;
  long long pr; int ev;
  fl_event_of(b.id[x.n_old + e], pr, ev);
  const int cell = g.idx(b.i[KID_BI_INE][pr], b.i[KID_BI_JNE][pr]);
  x.next[e] = atomicExch(x.head + cell, e);
}
template <class BP>
__global__ void __launch_bounds__(256) fl_assign_ids_rank(const DevGrid g, const BP *bt, const FlIdCtx x) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= x.m) return;
  const BP &b = *bt;
  long long pr; int ev;
  fl_event_of(b.id[x.n_old + e], pr, ev);
  const int pi = b.i[KID_BI_INE][pr], pj = b.i[KID_BI_JNE][pr];
  const int cell = g.idx(pi, pj);
  int before = 0;
  for (int o = x.head[cell]; o >= 0; o = x.next[o]) {
    if (o == e) continue;
    long long po; int eo;
    fl_event_of(b.id[x.n_old + o], po, eo);
    if (fl_event_before(b, po, eo, pr, ev)) ++before;
  }
  const int32_t cnt = x.counter[cell] + 1 + before;          // generate_id: the counter is incremented, then used
  x.newid[e] = (int64_t)cnt * ((int64_t)1 << 32) + (int64_t)(pi + (x.iNg * (pj - 1)) + x.ij0);
}
template <class BP>
__global__ void __launch_bounds__(256) fl_assign_ids_store(const DevGrid g, const BP *bt, const FlIdCtx x) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= x.m) return;
  const BP &b = *bt;
  long long pr; int ev;
  fl_event_of(b.id[x.n_old + e], pr, ev);   // (every lane decodes its own provisional id only)
  const int cell = g.idx(b.i[KID_BI_INE][pr], b.i[KID_BI_JNE][pr]);
  atomicAdd(x.counter + cell, 1);           // (integer: the order of the adds does not matter)
  x.head[cell] = -1;                        // (the same value from every child of the cell: the lists are empty again)
  b.id[x.n_old + e] = x.newid[e];
}

}  // namespace kid

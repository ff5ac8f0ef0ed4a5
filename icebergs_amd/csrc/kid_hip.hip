// kid_hip.hip -- kernels and C ABI of libkid_hip.so (gfx950 / MI355X).  See include/kid.h.
//
// Data layout in HBM
//   bergs   : structure of arrays, one contiguous fp64/int32 array of `capacity` elements per field of the
//             reference's `type iceberg` (FW:290-359), in reference traversal order (cell-major).
//   grid    : the 11 static + 11 forcing planes as the host passes them, plus three packed record arrays
//             (VelRec/TrcRec/GeoRec, kid_device.hpp) rebuilt by a prepass kernel whenever the forcing changes.
//   accum   : KID_NACC planes of ni*nj fp64 + KID_NSCALAR scalars in ONE block (one RCCL all-reduce per step),
//             followed by KID_NOUT derived planes.
// Kernels
//   pack_static / pack_forcing : per cell, builds the records and hoists the SSH-slope stencils (IB:4903-4926).
//   berg_kernel<RK,OLD,PH>     : one lane per berg; PH selects interp / evolve / thermodynamics / spreading, so
//                                the reference's phase-by-phase call sites and the fused per-step launch share
//                                one body.  HBM-streaming on the SoA, grid served from L2 / Infinity Cache.
//   gather_kernel              : per cell, the 9-point gather of sum_up_spread_fields (IB:6126-6138) + ustar.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string.h>
#include <string>
#include <vector>
#include <unordered_map>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "../../include/kid.h"
#include "kid_device.hpp"
#include "kid_thermo.hpp"
#include "kid_footloose.hpp"
#include "kid_berg_kernel.hpp"

using namespace kid;

// the plain hot build of the fused RK4 step: its own translation unit in the product build (kid_hot_plain.hip, another machine
// scheduler), included here in a one-file build
#ifdef KID_HOT_PLAIN_SEPARATE
namespace kid { int launch_hot_plain(int K, unsigned nblocks, void *stream, const DevGrid *gtab, const kid_params *pp, const void *bp, long long n, double *acc,
                                     size_t ncell, const void *flags, const void *redo); }
#else
#include "kid_hot_plain.inc"
#endif

namespace {

// -------------------------------------------------------------------------------------------------------
// grid prepass kernels
// -------------------------------------------------------------------------------------------------------
struct GridPlanes { const double *st[KID_NGRID_STATIC]; const double *fo[KID_NFORCING]; };

__global__ void __launch_bounds__(256) pack_static_kernel(GridPlanes gp, GeoRec *geo, double *hotok, double *latref, double pi_180, int ni, int nj, int latlon, double Lx) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ni * nj) return;
  GeoRec r;
  r.lon = gp.st[KID_G_LON][c]; r.lat = gp.st[KID_G_LAT][c]; r.area = gp.st[KID_G_AREA][c]; r.msk = gp.st[KID_G_MSK][c];
  geo[c] = r;
  if (latref) { double s_, c_; sincos(r.lat * pi_180, &s_, &c_); latref[2 * c] = s_; latref[2 * c + 1] = c_; }   // DevGrid::latref
  // DevGrid::hotok: may the hot build step a berg of this cell?  Its in-cell test is "(xi, yj) inside the unit square",
  // which is the reference's point-in-cell test (FW:6076-6160) exactly when the four corners form a strictly convex
  // quadrilateral (the bilinear map of calc_xiyj is then one-to-one onto it); polar cells (FW:6359) and cells whose
  // packet would reach outside the data domain are left to the general build.
  const int il = c % ni, jl = c / ni;
  double ok = 0.;
  if (il >= 1 && jl >= 1 && il + 1 < ni && jl + 1 < nj) {
    const double *lon = gp.st[KID_G_LON], *lat = gp.st[KID_G_LAT];
    const double x0 = lon[c - ni - 1], y0 = lat[c - ni - 1];
    const double x1 = mod_around(lon[c - ni], x0, Lx), y1 = lat[c - ni];
    const double x2 = mod_around(lon[c], x0, Lx), y2 = lat[c];
    const double x3 = mod_around(lon[c - 1], x0, Lx), y3 = lat[c - 1];
    const double k0 = (x1 - x0) * (y2 - y1) - (y1 - y0) * (x2 - x1), k1 = (x2 - x1) * (y3 - y2) - (y2 - y1) * (x3 - x2);
    const double k2 = (x3 - x2) * (y0 - y3) - (y3 - y2) * (x0 - x3), k3 = (x0 - x3) * (y1 - y0) - (y0 - y3) * (x1 - x0);
    const bool convex = (k0 > 0. && k1 > 0. && k2 > 0. && k3 > 0.) || (k0 < 0. && k1 < 0. && k2 < 0. && k3 < 0.);
    const bool polar = latlon && dmax(dmax(y0, y1), dmax(y2, y3)) >= 89.999;
    // 2: moreover its sides lie along the axes as calc_xiyj sees the corners (beta = delta = gamma = kappa = 0 there: its
    // linear branch, with xi = dx / alpha), the cell of every regular lat-lon grid -- the hot build takes that branch directly
    const double *lo = gp.st[KID_G_LON];
    const bool rect = lo[c - 1] == lo[c - ni - 1] && lo[c] == lo[c - ni] && lat[c - ni] == lat[c - ni - 1] && lat[c] == lat[c - 1];
    if (convex && !polar) ok = rect ? 2. : 1.;
  }
  hotok[c] = ok;
}

// The cell packets of the hot build (kid_device.hpp, PK_*).  A wave takes one row of cells at a time, lane q copying
// elements q and q + 64 of a cell from where packet_source() says they live -- resolved once per lane, as in the hot
// build's own staging before the packets were gathered -- for KID_PKT_CELLS_PER_WAVE consecutive cells, eight loads in
// flight.  (One lane per (cell, element) with the index arithmetic done per lane was ~150 instructions per element: 0.65 ms
// for the 2000 x 1000 grid of config 3, compute-bound; this is the bandwidth of 576 B written per cell.)  Cells on the rim
// of the data domain have no complete neighbourhood and are never hot (hotok = 0): only that flag is written for them.
// Runs after pack_forcing_kernel (it reads the neighbours' records).
enum { KID_PKT_CELLS_PER_WAVE = 32 };
__global__ void __launch_bounds__(256) pack_packets_kernel(const DevGrid g, double *__restrict__ pkt) {
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
  const int j = (int)blockIdx.y;
  const int i0 = ((int)blockIdx.x * 4 + wave) * KID_PKT_CELLS_PER_WAVE;
  if (i0 >= g.ni) return;
  const bool second = lane < PK_SIZE - 64;
  const PacketSrc s0 = packet_source(g, lane), s1 = packet_source(g, second ? 64 + lane : 0);
  const bool row_inside = j >= 1 && j < g.nj - 1;
  const int i1 = min(i0 + KID_PKT_CELLS_PER_WAVE, g.ni);
  for (int ib = i0; ib < i1; ib += 8) {
    double a[8], a2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = ib + u;
      const bool interior = row_inside && i >= 1 && i < g.ni - 1 && i < i1;
      const long long c = interior ? (long long)j * g.ni + i : (long long)g.ni + 1;   // (a cell whose whole neighbourhood exists)
      a[u] = *reinterpret_cast<const double *>(s0.base + c * s0.stride);
      a2[u] = *reinterpret_cast<const double *>(s1.base + c * s1.stride);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = ib + u;
      if (i >= i1) continue;
      const bool interior = row_inside && i >= 1 && i < g.ni - 1;
      // the packet's flag word (PkCell::flags): +-(1 + 2 [sides along the axes] + 4 [unrotated: cos = 1, sin = 0 at the four
      // corners]), positive where the hot build may step a berg of the cell (DevGrid::hotok, and no NaN among the
      // sea-surface-slope stencil values: IB:4869-4870 stays with the general build); 0 on the rim of the data domain
      const bool is_cos = lane < PK_CORNER && (lane & 7) == 0, is_sin = lane < PK_CORNER && (lane & 7) == 1;
      const bool rotated = __ballot((is_cos && a[u] != 1.) || (is_sin && a[u] != 0.)) != 0ull;
      const bool nan_stencil = __ballot(lane >= PK_DDX && lane < PK_AREA && a[u] != a[u]) != 0ull;
      double *o = pkt + ((long long)j * g.ni + i) * PK_GSTRIDE;
      o[lane] = interior ? a[u] : 0.;
      if (second) {
        double v = interior ? a2[u] : 0.;
        if (64 + lane == PK_HOTOK) v = interior ? ((v != 0. && !nan_stencil) ? 1. : -1.) * (1. + (v == 2. ? 2. : 0.) + (rotated ? 0. : 4.)) : 0.;
        o[64 + lane] = v;
      }
    }
  }
}

// Optional extras fused into the per-cell prepass (kid_step_prepare): keep a copy of the ssh plane, zero the cell's
// entries of the first `zero_planes` accumulator planes, zero two redo counters -- one launch instead of five.
struct PrepExtras { double *ssh_copy; double *acc; int zero_planes; int *cnt0, *cnt1; };
__global__ void __launch_bounds__(256) pack_forcing_kernel(GridPlanes gp, VelRec *vel, TrcRec *trc, int ni, int nj, PrepExtras ex) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ni * nj) return;
  if (ex.ssh_copy) ex.ssh_copy[c] = gp.fo[KID_F_SSH][c];
  if (ex.acc) {
    const size_t ncell = (size_t)ni * (size_t)nj;
    for (int q = 0; q < ex.zero_planes; ++q) ex.acc[(size_t)q * ncell + (size_t)c] = 0.;
  }
  if (c == 0) { if (ex.cnt0) *ex.cnt0 = 0; if (ex.cnt1) *ex.cnt1 = 0; }
  const int il = c % ni, jl = c / ni;
  VelRec v;
  v.cosr = gp.st[KID_G_COS][c]; v.sinr = gp.st[KID_G_SIN][c];
  v.uo = gp.fo[KID_F_UO][c]; v.vo = gp.fo[KID_F_VO][c]; v.ui = gp.fo[KID_F_UI][c]; v.vi = gp.fo[KID_F_VI][c];
  v.ua = gp.fo[KID_F_UA][c]; v.va = gp.fo[KID_F_VA][c];
  vel[c] = v;
  const double *dx = gp.st[KID_G_DX], *dy = gp.st[KID_G_DY], *msk = gp.st[KID_G_MSK], *ssh = gp.fo[KID_F_SSH];
  TrcRec t;
  t.sst = gp.fo[KID_F_SST][c]; t.sss = gp.fo[KID_F_SSS][c]; t.cn = gp.fo[KID_F_CN][c]; t.hi = gp.fo[KID_F_HI][c];
  t.od = gp.st[KID_G_OCEAN_DEPTH][c] + ssh[c];  // IB:4897
  t.msk = msk[c];
  t.ddx = 0.; t.ddy = 0.;
  if (il + 1 < ni && jl >= 1) {  // ddx_ssh(i,j) IB:4903-4913
    const double dxp = 0.5 * (dx[c + 1] + dx[c + 1 - ni]);
    const double dx0 = 0.5 * (dx[c] + dx[c - ni]);
    t.ddx = 2. * (ssh[c + 1] - ssh[c]) / (dx0 + dxp) * msk[c + 1] * msk[c];
  }
  if (jl + 1 < nj && il >= 1) {  // ddy_ssh(i,j) IB:4916-4926
    const double dyp = 0.5 * (dy[c + ni] + dy[c + ni - 1]);
    const double dy0 = 0.5 * (dy[c] + dy[c - 1]);
    t.ddy = 2. * (ssh[c + ni] - ssh[c]) / (dy0 + dyp) * msk[c + ni] * msk[c];
  }
  trc[c] = t;
}

// footloose_calving (IB:2503-2734): streaming pass, one lane per berg that existed when the pass started
__global__ void __launch_bounds__(256) footloose_kernel(const DevGrid g, const kid_params *__restrict__ pp, const BergPtrs *__restrict__ bt,
                                                        const FlChildCtx cx, double *__restrict__ acc, const size_t ncell) {
  const long long q = (long long)blockIdx.x * 256ll + threadIdx.x;
  if (q >= cx.n) return;
  const BergPtrs &b = *bt;
  if (b.i[KID_BI_ALIVE][q] == 0) return;
  footloose_one(g, *pp, b, cx, q, acc, ncell, acc - KID_NSCALAR);
}

// the device-side tables are rewritten by stream-ordered one-lane kernels (the new contents travel as kernel arguments):
// no host synchronisation, unlike a copy from pageable memory
__global__ void set_berg_table_kernel(const BergPtrs src, BergPtrs *dst) { if (threadIdx.x == 0 && blockIdx.x == 0) *dst = src; }
__global__ void set_params_kernel(const kid_params src, kid_params *dst) { if (threadIdx.x == 0 && blockIdx.x == 0) *dst = src; }
__global__ void set_time_kernel(int32_t year, double yearday, kid_params *dst) { if (threadIdx.x == 0 && blockIdx.x == 0) { dst->current_year = year; dst->current_yearday = yearday; } }
__global__ void set_grid_kernel(const DevGrid src, DevGrid *dst) { if (threadIdx.x == 0 && blockIdx.x == 0) *dst = src; }

#include "kid_mts.inc"

// -------------------------------------------------------------------------------------------------------
// IB:6077-6150 sum_up_spread_fields + IB:3449-3488, per cell of the computational domain
// -------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gather_kernel(const DevGrid g, const kid_params p, double *__restrict__ acc,
                                                     double *__restrict__ out, const size_t ncell, double *__restrict__ totals,
                                                     const double *__restrict__ spread_mass_old, const double *__restrict__ spread_mass_tmp) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < KID_NSCALAR) {  // fold this step's increments into the running totals kept on `bergs` (IB:3130, 3295)
    totals[t] += (acc - KID_NSCALAR)[t];   // (the step's scalar increments sit in front of plane 0)
    (acc - KID_NSCALAR)[t] = 0.;
  }
  const int nic = g.iec - g.isc + 1, njc = g.jec - g.jsc + 1;
  if (t >= nic * njc) return;
  const int i = g.isc + t % nic, j = g.jsc + t / nic;
  const int c = g.idx(i, j), ni = g.ni;
  const double a = g.geo[c].area, m = g.geo[c].msk;
  const int dm = p.diag_mask;
  const bool wrap = p.periodic_reentry != 0 && g.Lx > 0.;
  auto nine = [&](int base) {
    const double *v = acc + (size_t)base * ncell;
    // periodic_reentry: the neighbour column across the zonal seam (mpp_update_domains of var_on_ocean, IB:6103)
    const int wm = (wrap && i == g.isc) ? nic : 0, wp = (wrap && i == g.iec) ? -nic : 0;
#define KID_V(di, dj, s) v[(size_t)((s) - 1) * ncell + (size_t)(c + (di) + ((di) < 0 ? wm : ((di) > 0 ? wp : 0)) + (dj) * ni)]
    double dmda = KID_V(0, 0, 5) + (((KID_V(-1, -1, 9) + KID_V(1, 1, 1)) + (KID_V(1, -1, 7) + KID_V(-1, 1, 3)))
                                   + ((KID_V(-1, 0, 6) + KID_V(1, 0, 4)) + (KID_V(0, -1, 8) + KID_V(0, 1, 2))));
#undef KID_V
    if (a > 0) dmda = dmda / a * m;
    return dmda;
  };
  double su = 0., sv = 0., sa = 0.;
  if ((dm & KID_DIAG_SPREAD_UVEL) || p.pass_fields_to_ocean_model) su = nine(KID_A_UVEL_ON_OCEAN);
  if ((dm & KID_DIAG_SPREAD_VVEL) || p.pass_fields_to_ocean_model) sv = nine(KID_A_VVEL_ON_OCEAN);
  if ((dm & KID_DIAG_SPREAD_AREA) || p.pass_fields_to_ocean_model) sa = dmin(nine(KID_A_AREA_ON_OCEAN), 1.0);
  const double sm = nine(KID_A_MASS_ON_OCEAN);
  out[(size_t)KID_O_SPREAD_MASS * ncell + c] = sm;
  out[(size_t)KID_O_SPREAD_AREA * ncell + c] = sa;
  out[(size_t)KID_O_SPREAD_UVEL * ncell + c] = su;
  out[(size_t)KID_O_SPREAD_VVEL * ncell + c] = sv;
  if ((dm & KID_DIAG_U_ICEBERG) || (dm & KID_DIAG_V_ICEBERG)) {  // IB:3450-3462
    const double mass = acc[(size_t)KID_A_MASS * ncell + c];
    if (dm & KID_DIAG_U_ICEBERG) acc[(size_t)KID_A_U_ICEBERG * ncell + c] = (mass > 0.) ? acc[(size_t)KID_A_U_ICEBERG * ncell + c] / mass : 0.;
    if (dm & KID_DIAG_V_ICEBERG) acc[(size_t)KID_A_V_ICEBERG * ncell + c] = (mass > 0.) ? acc[(size_t)KID_A_V_ICEBERG * ncell + c] / mass : 0.;
  }
  double ustar_h = 0.;
  if ((dm & KID_DIAG_USTAR_ICEBERG) || p.pass_fields_to_ocean_model) {  // IB:3466-3474
    const double du = su - g.vel[c].uo, dv = sv - g.vel[c].vo;
    const double dvo = sqrt(du * du + dv * dv);
    const double ustar = sqrt(p.cdrag_icebergs * (dvo * dvo + p.utide_icebergs * p.utide_icebergs));
    ustar_h = dmax(p.ustar_icebergs_bg, ustar);
    if (sa == 0.0) ustar_h = 0.;
  }
  out[(size_t)KID_O_USTAR_ICEBERG * ncell + c] = ustar_h;
  if (spread_mass_old)  // find_melt_using_spread_mass: the melt flux is what the gridded mass lost over the step, IB:3436-3445
    acc[(size_t)KID_A_FLOATING_MELT * ncell + c] = (a > 0.0) ? dmax((spread_mass_old[c] - (spread_mass_tmp ? spread_mass_tmp[c] : sm)) / p.dt, 0.0) : 0.0;
  if (p.apply_thickness_cutoff_to_gridded_melt && (p.melt_cutoff >= 0.) && (sa > 0.)) {  // IB:3477-3488 (comp. domain)
    const double ave_thickness = sm / (sa * p.rho_bergs);
    const double ave_draft = ave_thickness * (p.rho_bergs / RHO_SEAWATER);
    if ((g.ocean_depth[c] - ave_draft) < p.melt_cutoff) {
      acc[(size_t)KID_A_FLOATING_MELT * ncell + c] = 0.0; acc[(size_t)KID_A_CALVING_HFLX * ncell + c] = 0.0;
    }
  }
}

// sum_up_spread_fields(.., 'mass') alone (IB:6126-6138) into a plane of its own: grd%spread_mass_old, IB:5495-5497
__global__ void __launch_bounds__(256) mass_gather_kernel(const DevGrid g, const double *__restrict__ acc, double *__restrict__ plane, const size_t ncell, const bool wrap) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int nic = g.iec - g.isc + 1, njc = g.jec - g.jsc + 1;
  if (t >= nic * njc) return;
  const int i = g.isc + t % nic, j = g.jsc + t / nic;
  const int c = g.idx(i, j), ni = g.ni;
  const double a = g.geo[c].area, m = g.geo[c].msk;
  const double *v = acc + (size_t)KID_A_MASS_ON_OCEAN * ncell;
  const int wm = (wrap && i == g.isc) ? nic : 0, wp = (wrap && i == g.iec) ? -nic : 0;
#define KID_V(di, dj, s) v[(size_t)((s) - 1) * ncell + (size_t)(c + (di) + ((di) < 0 ? wm : ((di) > 0 ? wp : 0)) + (dj) * ni)]
  double dmda = KID_V(0, 0, 5) + (((KID_V(-1, -1, 9) + KID_V(1, 1, 1)) + (KID_V(1, -1, 7) + KID_V(-1, 1, 3)))
                                 + ((KID_V(-1, 0, 6) + KID_V(1, 0, 4)) + (KID_V(0, -1, 8) + KID_V(0, 1, 2))));
#undef KID_V
  if (a > 0) dmda = dmda / a * m;
  plane[c] = dmda;
}

__global__ void __launch_bounds__(256) count_alive_kernel(const int32_t *alive, long long n, unsigned long long *out) {
  __shared__ unsigned wsum[4];
  unsigned local = 0;
  for (long long k = (long long)blockIdx.x * 256ll + threadIdx.x; k < n; k += (long long)gridDim.x * 256ll) local += alive[k] ? 1u : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (unsigned long long)(wsum[0] + wsum[1] + wsum[2] + wsum[3]));
}
__global__ void __launch_bounds__(256) compact_f64_kernel(const double *src, double *dst, const int32_t *alive, const unsigned *pos, long long n) {
  const long long k = (long long)blockIdx.x * 256ll + threadIdx.x;
  if (k < n && alive[k]) dst[pos[k]] = src[k];
}
__global__ void __launch_bounds__(256) compact_i32_kernel(const int32_t *src, int32_t *dst, const int32_t *alive, const unsigned *pos, long long n) {
  const long long k = (long long)blockIdx.x * 256ll + threadIdx.x;
  if (k < n && alive[k]) dst[pos[k]] = src[k];
}
__global__ void __launch_bounds__(256) compact_i64_kernel(const int64_t *src, int64_t *dst, const int32_t *alive, const unsigned *pos, long long n) {
  const long long k = (long long)blockIdx.x * 256ll + threadIdx.x;
  if (k < n && alive[k]) dst[pos[k]] = src[k];
}
// Rows that must survive a dead-tail drop, a compaction or a re-binning.  In a domain-decomposed run (kid_migrate.inc) a berg
// that left the tile is a DEAD row whose cell lies outside the computational domain until kid_pack_emigrants has collected
// it (evolve -> move_berg_between_cells -> send_bergs_to_other_pes is the reference's own order, IB:5433-5447); it counts as
// occupied here, keeps its cell as sort key and is never stepped (alive = 0).  Once packed, its cell is moved inside.
struct KeepSel { int isc, iec, jsc, jec; const double *halo; };
__global__ void __launch_bounds__(256) keep_flags_kernel(const int32_t *alive, const int32_t *ine, const int32_t *jne, const KeepSel s, int32_t *keep, long long n) {
  const long long k = (long long)blockIdx.x * 256ll + threadIdx.x;
  if (k >= n) return;
  const int i = ine[k], j = jne[k];
  const bool waiting = (i < s.isc || i > s.iec || j < s.jsc || j > s.jec) && !(s.halo && s.halo[k] >= 0.5);   // the selection of FW:3027-3035, 3101-3109
  keep[k] = (alive[k] != 0 || waiting) ? 1 : 0;
}
__global__ void __launch_bounds__(256) alive_to_u32_kernel(const int32_t *alive, unsigned *flag, long long n) {
  const long long k = (long long)blockIdx.x * 256ll + threadIdx.x;
  if (k < n) flag[k] = alive[k] ? 1u : 0u;
}
// cell key of every berg (dead bergs sort to the end) + identity permutation, for kid_move_berg_between_cells
__global__ void __launch_bounds__(256) cell_key_kernel(const int32_t *ine, const int32_t *jne, const int32_t *alive, int isd, int jsd, int ni,
                                                       unsigned dead_key, unsigned *key, unsigned *idx, long long n) {
  const long long k = (long long)blockIdx.x * 256ll + threadIdx.x;
  if (k < n) {
    key[k] = alive[k] ? (unsigned)((ine[k] - isd) + (jne[k] - jsd) * ni) : dead_key;
    idx[k] = (unsigned)k;
  }
}
// Counting sort by cell for kid_move_berg_between_cells: (1) key + rank of every berg within its cell (one atomic per
// berg on the cell's counter), (2) exclusive scan of the counters, (3) destination row = cell start + rank.  Three small
// launches instead of the ~20 of a comparison sort of 1e6 pairs; the order of the bergs inside a cell is the order the
// atomics arrive in (results never depend on the row order, see kid.h).
__global__ void __launch_bounds__(256) cell_rank_kernel(const int32_t *ine, const int32_t *jne, const int32_t *alive, int isd, int jsd, int ni,
                                                        unsigned dead_key, unsigned *key, unsigned *rank, unsigned *hist, long long n) {
  const long long k = (long long)blockIdx.x * 256ll + threadIdx.x;
  const bool in = k < n;
  const unsigned c = in ? (alive[k] ? (unsigned)((ine[k] - isd) + (jne[k] - jsd) * ni) : dead_key) : 0xffffffffu;
  // the SoA is almost sorted: the lanes of a wave form a few runs of equal cell.  One atomic per run (its head lane
  // adds the run length), not one per berg: ~14 lanes would otherwise queue on the same address.
  const int lane = (int)__lane_id();
  const unsigned prev = __shfl_up(c, 1);
  const bool head = (lane == 0) || (prev != c);
  const unsigned long long heads = __ballot(head);
  const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));   // heads at or below this lane
  const int head_lane = 63 - __clzll(below);
  const unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));                  // heads after this lane
  const int next_head = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;
  unsigned base = 0u;
  if (head && in) base = atomicAdd(hist + c, (unsigned)(next_head - lane));
  base = __shfl(base, head_lane);
  if (in) { key[k] = c; rank[k] = base + (unsigned)(lane - head_lane); }
}
__global__ void __launch_bounds__(256) cell_place_kernel(const unsigned *key, const unsigned *rank, const unsigned *start, unsigned *perm, unsigned *inv, long long n) {
  const long long k = (long long)blockIdx.x * 256ll + threadIdx.x;
  if (k < n) { const unsigned pos = start[key[k]] + rank[k]; perm[pos] = (unsigned)k; inv[k] = pos; }
}
// a list of row numbers through the re-binning (slow-lane schedule: the bergs handed over by the hot build just before it)
__global__ void __launch_bounds__(256) translate_list_kernel(int *list, const int *count, const unsigned *inv) {
  const int total = *count;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) list[t] = (int)inv[list[t]];
}
// every field of the SoA through the permutation in ONE launch: a thread owns a destination row, reads the source row
// number once and walks the fields (the loads of consecutive fields are independent)
enum { KID_PERM_MAX = KID_NB_F64 + KID_NB_I32 + 2 };   // every field, the ids, the lane array of the slow-lane schedule
struct PermTable { const void *src[KID_PERM_MAX]; void *dst[KID_PERM_MAX]; int n8, n4; };  // entries [0, n8) are 8-byte fields, [n8, n8 + n4) 4-byte
__global__ void __launch_bounds__(256) permute_all_kernel(const PermTable t, const unsigned *perm, long long n) {
  const long long k = (long long)blockIdx.x * 256ll + threadIdx.x;
  if (k >= n) return;
  const unsigned p = perm[k];
  int f = 0;
  for (; f + 4 <= t.n8; f += 4) {
    const double a = ((const double *)t.src[f])[p], b = ((const double *)t.src[f + 1])[p], c = ((const double *)t.src[f + 2])[p], d = ((const double *)t.src[f + 3])[p];
    ((double *)t.dst[f])[k] = a; ((double *)t.dst[f + 1])[k] = b; ((double *)t.dst[f + 2])[k] = c; ((double *)t.dst[f + 3])[k] = d;
  }
  for (; f < t.n8; ++f) ((double *)t.dst[f])[k] = ((const double *)t.src[f])[p];
  for (; f < t.n8 + t.n4; ++f) ((int32_t *)t.dst[f])[k] = ((const int32_t *)t.src[f])[p];
}
template <typename TT>
__global__ void __launch_bounds__(256) permute_kernel(const TT *src, TT *dst, const unsigned *perm, long long n) {
  const long long k = (long long)blockIdx.x * 256ll + threadIdx.x;
  if (k < n) dst[k] = src[perm[k]];
}
__global__ void __launch_bounds__(256) fill_i32_kernel(int32_t *p, int32_t v, long long n) {
  const long long k = (long long)blockIdx.x * 256ll + threadIdx.x;
  if (k < n) p[k] = v;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
struct kid_handle {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  kid_grid_desc gd{};
  kid_params params{};
  int ni = 0, nj = 0;
  size_t ncell = 0;
  int64_t capacity = 0, n = 0;
  // device memory
  double *d_static[KID_NGRID_STATIC] = {}, *d_forcing[KID_NFORCING] = {};
  VelRec *d_vel = nullptr; TrcRec *d_trc = nullptr; GeoRec *d_geo = nullptr; double *d_hotok = nullptr; double *d_latref = nullptr;
  // the accumulator block: KID_NSCALAR step scalars, then KID_NACC planes of ncell (scalars first, so that they and the planes a
  // step really fills -- a prefix -- are ONE contiguous range for the all-reduce); d_acc points at plane 0
  double *d_acc_own = nullptr, *d_acc = nullptr;
  double *d_out = nullptr;                        // KID_NOUT*ncell
  double *d_totals = nullptr;                     // KID_NSCALAR running totals (the block's scalars are per-step)
  BergPtrs bp{};
  BergPtrs *d_bp = nullptr;      // device copy of the field-pointer table
  kid_params *d_params = nullptr; // device copy of the parameter block
  DevGrid *d_grid = nullptr;      // device copy of the grid descriptor (berg_kernel reads it as a table)
  bool tables_dirty = true;
  double *d_spare_f64 = nullptr; unsigned *d_pos = nullptr, *d_flag = nullptr; void *d_scan_tmp = nullptr; size_t scan_tmp_bytes = 0;
  unsigned long long *d_count = nullptr;
  int *d_redo_list = nullptr, *d_redo_count = nullptr;  // bergs the FAST build hands to the general build
  // pipelined launches (kid_set_side_stream): the population goes through the hot build in two halves on the main
  // stream while the general build of each half runs on the side stream, under the hot build of the other half
  hipStream_t side_stream = nullptr; bool pipelined = false;
  int *d_redo_list2 = nullptr, *d_redo_count2 = nullptr;
  int *d_redo_cnt[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [part][parity]: a step's counters are zeroed by its prepass
  int redo_parity = 0; bool redo_prezeroed = false, acc_prezeroed = false;
  bool uploaded_nonzero[KID_NB_F64] = {};  // fields that held anything but zeros at the last upload
  bool tail_valid = false;                 // the dead rows are exactly the tail (true between a re-binning and the next launch)
  bool mig_mode = false;                   // the host exchanges bergs with other handles (kid_migrate.inc): dead rows waiting to be packed are kept
  int32_t *d_keep = nullptr;               // layout flags of such a run (keep_flags_kernel)
  long long mig_last[2] = {0, 0};          // records of the last pack call per slot: sizes the pinned staging buffer
  bool env_ever_stored = false;            // some launch since the last upload wrote berg%uo..od
  bool env_off_requested = false;          // kid_set_store_environment(h, 0)
  double *d_orient = nullptr;              // bond-derived hexagon orientation per berg (mts / interacting bergs)
  hipEvent_t evF[2] = {nullptr, nullptr}, evG[2] = {nullptr, nullptr}; bool evG_live[2] = {false, false};
  // "slow lane" schedule (kid_set_side_stream mode 2, launch_berg_lanes)
  int side_mode = 0; int *d_lane = nullptr, *d_lane_alt = nullptr; hipEvent_t evR = nullptr; int lane_step = 1; bool lanes_active = false, carry_valid = false, evC_live = false;
  hipEvent_t evC = nullptr, evP = nullptr;
  VelRec *d_vel2 = nullptr; TrcRec *d_trc2 = nullptr; DevGrid *d_grid2 = nullptr; int forc_parity = 0;  // forcing records of the odd steps
  double *d_pkt[2] = {nullptr, nullptr};   // gathered cell packets of the hot build, one set per parity of the forcing records
  // A/B switches for measurements and tests, read from the environment ONCE, when the handle is created (nothing on the
  // stepping path calls getenv): KID_MTS_NO_GRAPH, KID_STABLE_RESORT, KID_NO_PLAIN_BUILD, KID_FL_UNFUSED, KID_NEW_ORDER_UNFUSED,
  // KID_MTS_ALWAYS_LABEL, KID_MTS_NO_FUSED, KID_MTS_FUSED_BLOCKS_CAP (workgroups the fused sub-step kernel may use: tests of its
  // fall-back), KID_MTS_POLL_LIMIT (spins before a lane of that kernel gives up: tests of the time-out)
  struct DebugOpts { bool stable_resort = false, no_plain_build = false, fl_unfused = false, new_order_unfused = false, mts_always_label = false, mts_no_fused = false;
                     int mts_fused_blocks_cap = 0, mts_poll_limit = 0; } dbg;
  int32_t *d_iceberg_counter = nullptr;  // grd%iceberg_counter_grd (FW:1017)
  bool fl_place_warm = false;
  unsigned fl_step = 0;                  // footloose passes so far: third counter word of the child-placement generator (kid_rng.h)
  int *d_fl_cursor = nullptr;
  int32_t *d_fl_head = nullptr, *d_fl_next = nullptr; int64_t *d_fl_newid = nullptr; long long fl_ev_capacity = 0;   // fl_assign_ids_*
  unsigned *d_key[2] = {nullptr, nullptr}, *d_idx[2] = {nullptr, nullptr};  // radix-sort ping-pong buffers
  void *d_sort_tmp = nullptr; size_t sort_tmp_bytes = 0;
  double *d_perm_spare = nullptr;
  BergPtrs bp_alt{};            // second set of field arrays: the re-binning writes all fields there in one launch, then swaps
  unsigned *d_cell_hist = nullptr; void *d_cscan_tmp = nullptr; size_t cscan_tmp_bytes = 0; bool stable_resort = false;
  int resort_interval = 16, steps_since_sort = 0;
  // multiple time stepping / DEM
  MtsDev mts{}; MtsDev *d_mts = nullptr;
  int mb = 0; bool mts_ready = false, mts_dirty = true, have_bonds = false, visited = false;
  unsigned long long *d_key64[2] = {nullptr, nullptr}; int *d_rows[2] = {nullptr, nullptr};
  int *d_static_rows = nullptr; long long static_rows_n = -1, static_rows_cap = 0;   // rows ordered by the five static `inorder` keys (mts_build_order)
  unsigned long long *d_last_key = nullptr; int *d_order_flag = nullptr; long long order_n = -1;   // cell key of every row at the last sort: an unchanged population keeps its order
  unsigned long long *d_bond_sig = nullptr; bool labels_stale = true;   // per-berg signature of what the conglomerate labelling reads (set_conglom_ids skips itself while nothing changes)
  MtsDev mts_shadow{}; bool mts_shadow_valid = false;   // what d_mts holds (the table is re-uploaded only when it differs)
  int conglom_batch = 8;                                // label-propagation sweeps launched before the first convergence check (set_conglom_ids)
  void *d_mts_tmp = nullptr; size_t mts_tmp_bytes = 0;
  hipGraphExec_t sub_graph_exec = nullptr;  // the captured sub-step loop of evolve_icebergs_mts
  long long sub_graph_n = -1; int sub_graph_steps = 0; bool sub_graph_pair = false; double sub_graph_dt = 0.; hipStream_t sub_graph_stream = nullptr;
  bool use_graph = true;
  int fused_blocks[2] = {0, 0};             // co-resident workgroups of mts_substeps_kernel<4> / <8> (0: not asked yet)
  bool fused_ran = false;                   // a fused sub-step launch since the last look at its time-out counter
  Flags flags{0, 0, 1, 0, 0};
  // trajectories (kid_traj.inc): one buffer per sampled field, grown on demand
  bool traj_on = false; kid_traj_params traj_params{}; double *d_traj_f[64] = {}; double *d_traj_day = nullptr; int64_t *d_traj_id = nullptr;
  int32_t *d_traj_year = nullptr, *d_traj_nbonds = nullptr; unsigned long long *d_traj_cursor = nullptr; long long traj_capacity = 0, traj_count_bound = 0; int traj_nf = 0;
  double *d_mig_buf = nullptr; long long mig_capacity = 0; unsigned long long *mig_word = nullptr;   // pinned staging of kid_pack_emigrants / kid_unpack_immigrants
  double *d_btraj_f[16] = {}; int64_t *d_btraj_id[2] = {}; int32_t *d_btraj_i[2] = {}; long long btraj_capacity = 0, btraj_count_bound = 0;   // bond samples
  double *d_spread_mass_old = nullptr;   // grd%spread_mass_old (find_melt_using_spread_mass, IB:5495-5497) + spread_mass_tmp; the handle's own or the caller's (kid_bind_spread_mass_old)
  double *d_spread_mass_old_own = nullptr;
  bool have_static = false, have_forcing = false, have_planes = false;  // have_planes: d_forcing holds all eleven planes
  double *d_calv_state = nullptr, *d_calv_scal = nullptr, *d_calv_part = nullptr; unsigned char *d_calv_flag = nullptr; int2 *d_calv_list = nullptr; double *calv_host = nullptr; kid_calving_params calv_params{};  // kid_calving (kid_calving.inc)
  bool calving_on = false, calving_first_call = true, rmean_init = false, rmean_hflx_init = false;
  double *d_ingest_stage = nullptr; size_t ingest_stage_count = 0; unsigned long long *d_ingest_key = nullptr;  // kid_ingest_forcing
  bool profile = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double berg_ms = 0., all_ms = 0.; int64_t berg_launches = 0;
  std::string err;
};

#define KID_HIP(h, call)                                                                      \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);                          \
      return KID_EHIP;                                                                        \
    }                                                                                         \
  } while (0)

static DevGrid dev_grid(const kid_handle *h) {
  DevGrid g;
  g.isd = h->gd.isd; g.ied = h->gd.ied; g.jsd = h->gd.jsd; g.jed = h->gd.jed;
  g.isc = h->gd.isc; g.iec = h->gd.iec; g.jsc = h->gd.jsc; g.jec = h->gd.jec;
  g.ni = h->ni; g.nj = h->nj; g.latlon = h->gd.grid_is_latlon; g.regular = h->gd.grid_is_regular; g.Lx = h->gd.Lx;
  g.vel = h->forc_parity ? h->d_vel2 : h->d_vel; g.trc = h->forc_parity ? h->d_trc2 : h->d_trc; g.geo = h->d_geo; g.hotok = h->d_hotok; g.latref = h->d_latref;
  g.pkt = h->d_pkt[h->forc_parity ? 1 : 0];
  g.dx = h->d_static[KID_G_DX]; g.dy = h->d_static[KID_G_DY]; g.ocean_depth = h->d_static[KID_G_OCEAN_DEPTH];
  g.ssh = h->d_forcing[KID_F_SSH];
  g.sin_lat_ref = sin((h->params.pi / 180.) * h->params.lat_ref);
  g.pi_180 = h->params.pi / 180.; g.r180_pi = 180. / h->params.pi; g.dydl = (180. / h->params.pi) / h->params.Rearth;
  g.rho_ratio = h->params.rho_bergs / RHO_SEAWATER;
  g.fl_e1 = exp(0.25 * h->params.pi);
  return g;
}
// area/Uvel/Vvel_on_ocean (27 planes) are intermediates of spread_area / spread_uvel / spread_vvel / ustar_iceberg
// (IB:3419-3474), which the reference evaluates only for `id_*>0` or pass_fields_to_ocean_model: when nobody reads them
// they are neither scattered nor zeroed nor exchanged (the reference fills them regardless and drops them)
static bool footprint_needed(const kid_params &p) {
  const int diag = KID_DIAG_SPREAD_UVEL | KID_DIAG_SPREAD_VVEL | KID_DIAG_SPREAD_AREA | KID_DIAG_USTAR_ICEBERG;
  return p.pass_fields_to_ocean_model || (p.diag_mask & diag);
}
static int nacc_active(const kid_handle *h) {
  const int diag_planes = KID_DIAG_MELT_BY_CLASS | KID_DIAG_FL_PARENT_MELT | KID_DIAG_FL_CHILD_MELT | KID_DIAG_MELT_BUOY |
                          KID_DIAG_MELT_EROS | KID_DIAG_MELT_CONV | KID_DIAG_MELT_BUOY_FL | KID_DIAG_MELT_EROS_FL |
                          KID_DIAG_MELT_CONV_FL | KID_DIAG_VIRTUAL_AREA | KID_DIAG_MASS | KID_DIAG_U_ICEBERG | KID_DIAG_V_ICEBERG;
  if (h->params.diag_mask & diag_planes) return KID_NACC;
  return footprint_needed(h->params) ? KID_NACC_CORE : KID_A_MASS_ON_OCEAN + 9;
}
static int check_params(kid_handle *h, const kid_params *p) {
  if (p->dem && !p->mts) { h->err = "dem needs mts=T (the DEM forces live on the MTS sub-steps)"; return KID_EINVAL; }
  if (p->mts) {
    if (p->Runge_not_Verlet) { h->err = "Runge_not_Verlet must be false to use MTS or DEM (FW:1485-1490)"; return KID_EINVAL; }
    if (p->old_interp_flds_order) { h->err = "old_interp_flds_order is false whenever mts/dem is on (FW:1483)"; return KID_EINVAL; }
    if (p->mts_sub_steps < 1) { h->err = "mts_sub_steps must be >= 1 (pass the value ice_bergs_framework_init derives, FW:1296-1301)"; return KID_EINVAL; }
    if (p->dem && !p->explicit_inner_mts) { h->err = "dem forces explicit_inner_mts=T (FW:1433)"; return KID_EINVAL; }
    if (p->dem && !p->iceberg_bonds_on) { h->err = "dem needs iceberg_bonds_on"; return KID_EINVAL; }
    if (p->max_bonds < 1 || p->max_bonds > KID_MAX_BONDS) { h->err = "max_bonds out of range"; return KID_EINVAL; }
    if (p->use_broken_bonds_for_substep_contact && !(p->break_bonds_on_sub_steps && p->dem && p->iceberg_bonds_on)) {
      h->err = "use_broken_bonds_for_substep_contact requires break_bonds_on_sub_steps, dem and iceberg_bonds_on (FW:1438-1447)"; return KID_EINVAL; }
    if (p->footloose) { h->err = "footloose together with mts is not implemented"; return KID_EUNSUPPORTED; }
  } else if (p->interactive_icebergs_on || p->iceberg_bonds_on) {
    if (p->Runge_not_Verlet) { h->err = "interacting / bonded bergs under the single-time-step scheme are implemented for Verlet only (Runge_not_Verlet=F)"; return KID_EUNSUPPORTED; }
    if (p->footloose) { h->err = "footloose together with interacting bergs is not implemented"; return KID_EUNSUPPORTED; }
    if (p->iceberg_bonds_on && !p->interactive_icebergs_on) { h->err = "iceberg_bonds_on without interactive_icebergs_on is not implemented"; return KID_EUNSUPPORTED; }
    if (p->max_bonds < 1 || p->max_bonds > KID_MAX_BONDS) { h->err = "max_bonds out of range"; return KID_EINVAL; }
  }
  if (p->tidal_drift > 0.) { h->err = "tidal_drift needs FMS's random stream: not supported"; return KID_EUNSUPPORTED; }
  // time_average_weight=T is accepted: the spreading it moves into the integrator stages (IB:7264, 7395, 7433, 7490, 7620)
  // fills mass/area/Uvel/Vvel_on_ocean, but calculate_mass_on_ocean zeroes those planes again (IB:4984-4987) without
  // refilling them (IB:4997), and nothing reads them in between: the reference's spread_mass is identically zero in
  // this mode.  So the stage spreading is not launched and the spreading phase skips the berg (berg_kernel, PH_SPREAD).
  if (p->find_melt_using_spread_mass && (p->mts || p->interactive_icebergs_on || p->footloose)) {
    h->err = "find_melt_using_spread_mass is implemented for the plain evolve loop only (no bonds, interactions or footloose)";
    return KID_EUNSUPPORTED;
  }
  if (p->Runge_not_Verlet && p->footloose) { h->err = "Runge_not_Verlet must be false to use footloose (FW:1485-1490)"; return KID_EINVAL; }
  if (p->footloose && !p->use_operator_splitting) { h->err = "use_operator_splitting must be true to use footloose (FW:1476)"; return KID_EINVAL; }
  return KID_OK;
}

extern "C" {

// The build's arithmetic and measurement switches are part of its identity: a library built with -DKID_EXPERIMENTS (KID_EXP_*
// measurement macros, possibly wrong answers) or -DKID_EXACT_MATH (IEEE division/sqrt/pow everywhere) says so.
#ifndef KID_BUILD_ID
#define KID_BUILD_ID "unknown"
#endif
// "src <id>": the first 12 hex digits of the SHA-1 of the library's sources (csrc/Makefile): profiles name the build they measured
const char *kid_version(void) {
  return "kid_hip 0.3 (gfx950) src " KID_BUILD_ID
#ifdef KID_EXACT_MATH
         " exact-math"
#endif
#ifdef KID_EXPERIMENTS
         " EXPERIMENTS"
#endif
      ;
}
int64_t kid_sizeof(int which) {
  switch (which) { case 0: return (int64_t)sizeof(kid_params); case 1: return (int64_t)sizeof(kid_grid_desc);
                   case 2: return (int64_t)sizeof(kid_berg_soa); case 3: return (int64_t)sizeof(kid_bond_soa);
                   case 4: return (int64_t)sizeof(kid_forcing_in);
                   case 5: return (int64_t)sizeof(kid_calving_params); case 6: return (int64_t)sizeof(kid_calving_in); default: return -1; }
}
#ifdef KID_EXP_TIMING
// measurement build only: the segment clock of the per-berg kernels (kid_device.hpp, KID_TICK); out[0] = waves, out[1+n] = cycles of segment n
int kid_exp_timing(unsigned long long *out, int reset) {
  if (out) {
    std::vector<unsigned long long> all((size_t)KID_TPROF_WAVES * 16);
    if (hipMemcpyFromSymbol(all.data(), HIP_SYMBOL(kid_tprof), all.size() * sizeof(unsigned long long)) != hipSuccess) return KID_EHIP;
    for (int q = 0; q < 16; ++q) out[q] = 0ull;
    for (size_t r = 0; r < (size_t)KID_TPROF_WAVES; ++r) for (int q = 0; q < 16; ++q) out[q] += all[r * 16 + q];
  }
  if (reset) {
    void *sym = nullptr;
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(kid_tprof)) != hipSuccess || hipMemset(sym, 0, (size_t)KID_TPROF_WAVES * 16 * sizeof(unsigned long long)) != hipSuccess) return KID_EHIP;
  }
  return KID_OK;
}
#endif
const char *kid_last_error(const kid_handle *h) { return h ? h->err.c_str() : "null handle"; }

int kid_create(const kid_grid_desc *grid, const kid_params *params, int64_t capacity, int device, kid_handle **out) {
  if (!grid || !params || !out || capacity <= 0) return KID_EINVAL;
  if (capacity > (int64_t(1) << 29)) return KID_EINVAL;   // the per-berg kernel addresses a row by a 32-bit byte offset (2^29 rows of 8 bytes; 240 GB of berg state)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return KID_ENODEV;
  if (device < 0 || device >= ndev) return KID_ENODEV;
  kid_handle *h = new kid_handle();
  *out = h;  // returned even on failure so that kid_last_error() can be read; caller still destroys it
  h->device = device;
  KID_HIP(h, hipSetDevice(device));
  h->gd = *grid;
  h->ni = grid->ied - grid->isd + 1; h->nj = grid->jed - grid->jsd + 1;
  if (h->ni < 5 || h->nj < 5 || grid->isc - grid->isd < 2 || grid->ied - grid->iec < 2 || grid->jsc - grid->jsd < 2 || grid->jed - grid->jec < 2) {
    h->err = "grid needs a halo of at least 2 cells"; return KID_EINVAL;
  }
  h->ncell = (size_t)h->ni * (size_t)h->nj;
  int rc = check_params(h, params);
  if (rc) return rc;
  h->params = *params;
  h->flags.footprint = footprint_needed(h->params) ? 1 : 0;
  h->capacity = capacity;
  if (getenv("KID_MTS_NO_GRAPH")) h->use_graph = false;  // A/B switch for measurements
  h->dbg.stable_resort = getenv("KID_STABLE_RESORT") != nullptr; h->dbg.no_plain_build = getenv("KID_NO_PLAIN_BUILD") != nullptr;
  h->dbg.fl_unfused = getenv("KID_FL_UNFUSED") != nullptr; h->dbg.new_order_unfused = getenv("KID_NEW_ORDER_UNFUSED") != nullptr;
  h->dbg.mts_always_label = getenv("KID_MTS_ALWAYS_LABEL") != nullptr; h->dbg.mts_no_fused = getenv("KID_MTS_NO_FUSED") != nullptr;
  if (const char *e = getenv("KID_MTS_FUSED_BLOCKS_CAP")) h->dbg.mts_fused_blocks_cap = atoi(e);
  if (const char *e = getenv("KID_MTS_POLL_LIMIT")) h->dbg.mts_poll_limit = atoi(e);
  KID_HIP(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  for (int k = 0; k < KID_NGRID_STATIC; ++k) { KID_HIP(h, hipMalloc(&h->d_static[k], h->ncell * sizeof(double))); KID_HIP(h, hipMemset(h->d_static[k], 0, h->ncell * sizeof(double))); }
  for (int k = 0; k < KID_NFORCING; ++k) { KID_HIP(h, hipMalloc(&h->d_forcing[k], h->ncell * sizeof(double))); KID_HIP(h, hipMemset(h->d_forcing[k], 0, h->ncell * sizeof(double))); }
  KID_HIP(h, hipMalloc(&h->d_vel, h->ncell * sizeof(VelRec)));
  KID_HIP(h, hipMalloc(&h->d_trc, h->ncell * sizeof(TrcRec)));
  KID_HIP(h, hipMalloc(&h->d_geo, h->ncell * sizeof(GeoRec)));
  KID_HIP(h, hipMalloc(&h->d_hotok, h->ncell * sizeof(double)));
  KID_HIP(h, hipMemset(h->d_hotok, 0, h->ncell * sizeof(double)));
  if (grid->grid_is_latlon) KID_HIP(h, hipMalloc(&h->d_latref, 2 * h->ncell * sizeof(double)));   // DevGrid::latref
  const size_t accn = (size_t)KID_NACC * h->ncell + KID_NSCALAR;
  KID_HIP(h, hipMalloc(&h->d_acc_own, accn * sizeof(double)));
  KID_HIP(h, hipMemset(h->d_acc_own, 0, accn * sizeof(double)));
  h->d_acc = h->d_acc_own + KID_NSCALAR;
  KID_HIP(h, hipMalloc(&h->d_out, (size_t)KID_NOUT * h->ncell * sizeof(double)));
  KID_HIP(h, hipMemset(h->d_out, 0, (size_t)KID_NOUT * h->ncell * sizeof(double)));
  KID_HIP(h, hipMalloc(&h->d_totals, KID_NSCALAR * sizeof(double)));
  KID_HIP(h, hipMemset(h->d_totals, 0, KID_NSCALAR * sizeof(double)));
  for (int f = 0; f < KID_NB_F64; ++f) { KID_HIP(h, hipMalloc(&h->bp.f[f], (size_t)capacity * sizeof(double))); KID_HIP(h, hipMemset(h->bp.f[f], 0, (size_t)capacity * sizeof(double))); }
  for (int f = 0; f < KID_NB_I32; ++f) { KID_HIP(h, hipMalloc(&h->bp.i[f], (size_t)capacity * sizeof(int32_t))); KID_HIP(h, hipMemset(h->bp.i[f], 0, (size_t)capacity * sizeof(int32_t))); }
  KID_HIP(h, hipMalloc(&h->bp.id, (size_t)capacity * sizeof(int64_t)));
  KID_HIP(h, hipMemset(h->bp.id, 0, (size_t)capacity * sizeof(int64_t)));
  KID_HIP(h, hipMalloc(&h->d_bp, sizeof(BergPtrs)));
  KID_HIP(h, hipMalloc(&h->d_params, sizeof(kid_params)));
  KID_HIP(h, hipMalloc(&h->d_grid, sizeof(DevGrid)));
  KID_HIP(h, hipMalloc(&h->d_grid2, sizeof(DevGrid)));
  KID_HIP(h, hipMalloc(&h->d_vel2, h->ncell * sizeof(VelRec)));
  KID_HIP(h, hipMalloc(&h->d_trc2, h->ncell * sizeof(TrcRec)));
  for (int q = 0; q < 2; ++q) { KID_HIP(h, hipMalloc(&h->d_pkt[q], (size_t)h->ncell * PK_GSTRIDE * sizeof(double))); KID_HIP(h, hipMemset(h->d_pkt[q], 0, (size_t)h->ncell * PK_GSTRIDE * sizeof(double))); }
  KID_HIP(h, hipMalloc(&h->d_lane, (size_t)capacity * sizeof(int)));
  KID_HIP(h, hipMemset(h->d_lane, 0, (size_t)capacity * sizeof(int)));
  KID_HIP(h, hipEventCreateWithFlags(&h->evC, hipEventDisableTiming)); KID_HIP(h, hipEventCreateWithFlags(&h->evP, hipEventDisableTiming));
  KID_HIP(h, hipMalloc(&h->d_count, 4 * sizeof(unsigned long long)));
  KID_HIP(h, hipMalloc(&h->d_iceberg_counter, h->ncell * sizeof(int32_t)));
  KID_HIP(h, hipMemset(h->d_iceberg_counter, 0, h->ncell * sizeof(int32_t)));
  KID_HIP(h, hipMalloc(&h->d_fl_cursor, sizeof(int)));
  KID_HIP(h, hipMalloc(&h->d_redo_list, (size_t)capacity * sizeof(int)));
  KID_HIP(h, hipMalloc(&h->d_redo_count, sizeof(int)));
  KID_HIP(h, hipMalloc(&h->d_redo_list2, (size_t)capacity * sizeof(int)));
  KID_HIP(h, hipMalloc(&h->d_redo_count2, 4 * sizeof(int)));
  KID_HIP(h, hipMemset(h->d_redo_count2, 0, 4 * sizeof(int)));
  for (int part = 0; part < 2; ++part) for (int par = 0; par < 2; ++par) h->d_redo_cnt[part][par] = h->d_redo_count2 + (2 * part + par);
  for (int q = 0; q < 2; ++q) { KID_HIP(h, hipEventCreateWithFlags(&h->evF[q], hipEventDisableTiming)); KID_HIP(h, hipEventCreateWithFlags(&h->evG[q], hipEventDisableTiming)); }
  KID_HIP(h, hipEventCreate(&h->ev0)); KID_HIP(h, hipEventCreate(&h->ev1));
  KID_HIP(h, hipEventCreate(&h->ev2)); KID_HIP(h, hipEventCreate(&h->ev3));
  // The hipMemset calls above run on the null stream and are asynchronous to the host, while all later work goes to a
  // non-blocking stream that does not order itself behind the null stream: without this wait a zero-fill could land
  // after the first upload (seen once as a zeroed `ine` array of a 1-berg population).
  KID_HIP(h, hipDeviceSynchronize());
  return KID_OK;
}

static void mts_free(kid_handle *h);
int kid_destroy(kid_handle *h) {
  if (!h) return KID_EINVAL;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (auto &k : h->d_static) if (k) (void)hipFree(k);
  for (auto &k : h->d_forcing) if (k) (void)hipFree(k);
  if (h->d_ingest_stage) (void)hipFree(h->d_ingest_stage);
  if (h->d_calv_state) (void)hipFree(h->d_calv_state);
  if (h->d_calv_scal) (void)hipFree(h->d_calv_scal);
  if (h->d_calv_part) (void)hipFree(h->d_calv_part);
  if (h->d_calv_flag) (void)hipFree(h->d_calv_flag);
  if (h->d_calv_list) (void)hipFree(h->d_calv_list);
  if (h->calv_host) (void)hipHostFree(h->calv_host);
  if (h->d_ingest_key) (void)hipFree(h->d_ingest_key);
  if (h->d_vel) (void)hipFree(h->d_vel);
  if (h->d_trc) (void)hipFree(h->d_trc);
  if (h->d_geo) (void)hipFree(h->d_geo);
  if (h->d_hotok) (void)hipFree(h->d_hotok);
  if (h->d_latref) (void)hipFree(h->d_latref);
  if (h->d_keep) (void)hipFree(h->d_keep);
  if (h->d_acc_own) (void)hipFree(h->d_acc_own);
  if (h->d_out) (void)hipFree(h->d_out);
  if (h->d_totals) (void)hipFree(h->d_totals);
  for (auto &k : h->bp.f) if (k) (void)hipFree(k);
  for (auto &k : h->bp.i) if (k) (void)hipFree(k);
  if (h->bp.id) (void)hipFree(h->bp.id);
  if (h->d_spare_f64) (void)hipFree(h->d_spare_f64);
  if (h->d_pos) (void)hipFree(h->d_pos);
  if (h->d_flag) (void)hipFree(h->d_flag);
  if (h->d_scan_tmp) (void)hipFree(h->d_scan_tmp);
  if (h->d_count) (void)hipFree(h->d_count);
  if (h->d_bp) (void)hipFree(h->d_bp);
  if (h->d_params) (void)hipFree(h->d_params);
  if (h->d_grid) (void)hipFree(h->d_grid);
  if (h->d_grid2) (void)hipFree(h->d_grid2);
  if (h->d_vel2) (void)hipFree(h->d_vel2);
  if (h->d_trc2) (void)hipFree(h->d_trc2);
  for (int q = 0; q < 2; ++q) if (h->d_pkt[q]) (void)hipFree(h->d_pkt[q]);
  if (h->d_lane) (void)hipFree(h->d_lane);
  if (h->d_lane_alt) (void)hipFree(h->d_lane_alt);
  if (h->d_spread_mass_old_own) (void)hipFree(h->d_spread_mass_old_own);
  for (auto &q : h->d_traj_f) if (q) (void)hipFree(q);
  if (h->d_traj_day) (void)hipFree(h->d_traj_day);
  if (h->d_traj_id) (void)hipFree(h->d_traj_id);
  if (h->d_traj_year) (void)hipFree(h->d_traj_year);
  if (h->d_traj_nbonds) (void)hipFree(h->d_traj_nbonds);
  if (h->d_traj_cursor) (void)hipFree(h->d_traj_cursor);
  if (h->d_mig_buf) (void)hipHostFree(h->d_mig_buf);
  if (h->mig_word) (void)hipHostFree(h->mig_word);
  for (auto &q : h->d_btraj_f) if (q) (void)hipFree(q);
  for (int f = 0; f < 2; ++f) { if (h->d_btraj_id[f]) (void)hipFree(h->d_btraj_id[f]); if (h->d_btraj_i[f]) (void)hipFree(h->d_btraj_i[f]); }
  if (h->evR) (void)hipEventDestroy(h->evR);
  if (h->evC) (void)hipEventDestroy(h->evC);
  if (h->evP) (void)hipEventDestroy(h->evP);
  for (int q = 0; q < 2; ++q) { if (h->d_key[q]) (void)hipFree(h->d_key[q]); if (h->d_idx[q]) (void)hipFree(h->d_idx[q]); }
  if (h->d_sort_tmp) (void)hipFree(h->d_sort_tmp);
  if (h->d_perm_spare) (void)hipFree(h->d_perm_spare);
  for (auto &q : h->bp_alt.f) if (q) (void)hipFree(q);
  for (auto &q : h->bp_alt.i) if (q) (void)hipFree(q);
  if (h->bp_alt.id) (void)hipFree(h->bp_alt.id);
  if (h->d_cell_hist) (void)hipFree(h->d_cell_hist);
  if (h->d_cscan_tmp) (void)hipFree(h->d_cscan_tmp);
  if (h->d_iceberg_counter) (void)hipFree(h->d_iceberg_counter);
  if (h->d_fl_cursor) (void)hipFree(h->d_fl_cursor);
  if (h->d_fl_head) (void)hipFree(h->d_fl_head);
  if (h->d_fl_next) (void)hipFree(h->d_fl_next);
  if (h->d_fl_newid) (void)hipFree(h->d_fl_newid);
  mts_free(h);
  if (h->d_orient) (void)hipFree(h->d_orient);
  if (h->d_redo_list) (void)hipFree(h->d_redo_list);
  if (h->d_redo_count) (void)hipFree(h->d_redo_count);
  if (h->d_redo_list2) (void)hipFree(h->d_redo_list2);
  if (h->d_redo_count2) (void)hipFree(h->d_redo_count2);
  for (int q = 0; q < 2; ++q) { if (h->evF[q]) (void)hipEventDestroy(h->evF[q]); if (h->evG[q]) (void)hipEventDestroy(h->evG[q]); }
  for (auto &pe : h->pending) { (void)hipEventDestroy(pe.first); (void)hipEventDestroy(pe.second); }
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev2) (void)hipEventDestroy(h->ev2);
  if (h->ev3) (void)hipEventDestroy(h->ev3);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return KID_OK;
}

int kid_set_params(kid_handle *h, const kid_params *params) {
  if (!h || !params) return KID_EINVAL;
  int rc = check_params(h, params);
  if (rc) return rc;
  if (!h->tables_dirty && h->d_params) {
    // A model clock that advances (bergs%current_year / current_yearday, every step of a real run) is not a reason to
    // rebuild the device tables -- which waits for the side stream, i.e. puts the slow lane's general build back between
    // two hot builds.  The two words are written in place by a stream-ordered one-lane kernel; the launches that may
    // still be in flight on the side stream (fused step without footloose) never read them.
    kid_params a = h->params, b = *params;
    a.current_year = b.current_year = 0; a.current_yearday = b.current_yearday = 0.;
    if (std::memcmp(&a, &b, sizeof(kid_params)) == 0) {
      h->params = *params;
      KID_HIP(h, hipSetDevice(h->device));
      hipLaunchKernelGGL(set_time_kernel, dim3(1), dim3(64), 0, h->stream, params->current_year, params->current_yearday, h->d_params);
      KID_HIP(h, hipGetLastError());
      return KID_OK;
    }
  }
  h->params = *params;
  h->tables_dirty = true;
  h->labels_stale = true;   // (the conglomerate labelling reads dem / max_bonds / use_broken_bonds_for_substep_contact)
  h->flags.footprint = footprint_needed(h->params) ? 1 : 0;
  if (!h->params.old_interp_flds_order && !h->env_off_requested) h->flags.store_env = 1;  // the stored environment is an input again
  return KID_OK;
}
static int refresh_tables(kid_handle *h);
// order the main stream behind general-build launches that are still in flight on the side stream
static int join_side(kid_handle *h) {
  for (int q = 0; q < 2; ++q)
    if (h->evG_live[q]) { KID_HIP(h, hipStreamWaitEvent(h->stream, h->evG[q], 0)); }
  if (h->evC_live) { KID_HIP(h, hipStreamWaitEvent(h->stream, h->evC, 0)); }
  return KID_OK;
}
// Leave the slow-lane schedule: both lanes are complete through the last step once the side stream is joined, so every
// berg can go back to the hot build and nothing is carried over.  Needed before anything that moves rows (the lists
// hold row numbers) or that launches outside launch_berg_lanes.
static int lanes_drain(kid_handle *h) {
  const int rc = join_side(h);
  if (rc || !h->lanes_active) return rc;
  KID_HIP(h, hipMemsetAsync(h->d_lane, 0, (size_t)h->capacity * sizeof(int), h->stream));
  // the other buffer of the in-step re-binning too: its rows beyond the population would come back with stamps of a
  // step count that starts over below
  if (h->d_lane_alt) KID_HIP(h, hipMemsetAsync(h->d_lane_alt, 0, (size_t)h->capacity * sizeof(int), h->stream));
  h->lanes_active = false; h->carry_valid = false; h->lane_step = 1;
  return KID_OK;
}
int kid_set_side_stream(kid_handle *h, void *s, int enable) {
  if (!h) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  int rc = join_side(h);
  if (rc) return rc;
  KID_HIP(h, hipStreamSynchronize(h->stream));
  { const int rc_d = lanes_drain(h); if (rc_d) return rc_d; }
  KID_HIP(h, hipStreamSynchronize(h->stream));
  h->evG_live[0] = h->evG_live[1] = false; h->evC_live = false;
  h->side_stream = (hipStream_t)s;
  h->pipelined = enable == 1;
  h->side_mode = (s && enable == 2) ? 2 : 0;
  return KID_OK;
}
int kid_set_stream(kid_handle *h, void *s) {
  if (!h) return KID_EINVAL;
  // NULL is the device's default (null) stream -- e.g. torch's default current stream -- not "no stream": work the
  // caller orders with its own events or collectives must really be on the stream it names
  h->stream = (hipStream_t)s;
  return KID_OK;
}
// the cell hash of a berg id (ij_component_of_id FW:4227-4240) on this grid: i + iNg * (j - 1) + ij0 with local i, j
static inline int id_iNg(const kid_grid_desc &d) { return d.gni > 0 ? d.gni : d.iec - d.isc + 1; }
static inline int id_ij0(const kid_grid_desc &d) { return d.gni > 0 ? d.gi0 + d.gni * d.gj0 : 0; }
static int mts_check_timeout(kid_handle *h);   // kid_mts_host.inc
int kid_sync(kid_handle *h) {
  if (!h) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  { const int rc_j = join_side(h); if (rc_j) return rc_j; }
  KID_HIP(h, hipStreamSynchronize(h->stream));
  return mts_check_timeout(h);
}

// gather the cell packets of the current parity from the record arrays (after either of them has changed)
static int pack_packets(kid_handle *h) {
  const unsigned bx = (unsigned)((h->ni + 4 * KID_PKT_CELLS_PER_WAVE - 1) / (4 * KID_PKT_CELLS_PER_WAVE));
  hipLaunchKernelGGL(pack_packets_kernel, dim3(bx, (unsigned)h->nj), dim3(256), 0, h->stream, dev_grid(h), h->d_pkt[h->forc_parity ? 1 : 0]);
  KID_HIP(h, hipGetLastError());
  return KID_OK;
}
static int pack_static(kid_handle *h) {
  GridPlanes gp;
  for (int k = 0; k < KID_NGRID_STATIC; ++k) gp.st[k] = h->d_static[k];
  for (int k = 0; k < KID_NFORCING; ++k) gp.fo[k] = h->d_forcing[k];
  const int nb = (int)((h->ncell + 255) / 256);
  hipLaunchKernelGGL(pack_static_kernel, dim3(nb), dim3(256), 0, h->stream, gp, h->d_geo, h->d_hotok, h->d_latref, h->params.pi / 180., h->ni, h->nj, (int)h->gd.grid_is_latlon, h->gd.Lx);
  KID_HIP(h, hipGetLastError());
  return h->have_forcing ? pack_packets(h) : KID_OK;   // (the other parity's packets are rebuilt by the pack_forcing that precedes their use)
}
// src[k]: where forcing plane k currently lives on the device (the handle's own copy, or the caller's buffer)
static int pack_forcing(kid_handle *h, const double *const src[KID_NFORCING], const PrepExtras ex = PrepExtras{nullptr, nullptr, 0, nullptr, nullptr}) {
  GridPlanes gp;
  for (int k = 0; k < KID_NGRID_STATIC; ++k) gp.st[k] = h->d_static[k];
  for (int k = 0; k < KID_NFORCING; ++k) gp.fo[k] = src[k] ? src[k] : h->d_forcing[k];
  const int nb = (int)((h->ncell + 255) / 256);
  // a side stream is in use: general-build launches of the previous step may still read its records -> alternate two sets
  if (h->side_mode == 2 || h->pipelined) h->forc_parity ^= 1;
  hipLaunchKernelGGL(pack_forcing_kernel, dim3(nb), dim3(256), 0, h->stream, gp, h->forc_parity ? h->d_vel2 : h->d_vel, h->forc_parity ? h->d_trc2 : h->d_trc, h->ni, h->nj, ex);
  return pack_packets(h);
}

int kid_set_static_grid(kid_handle *h, const double *const fields[KID_NGRID_STATIC]) {
  if (!h || !fields) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  for (int k = 0; k < KID_NGRID_STATIC; ++k) {
    if (!fields[k]) { if (k == KID_G_LONC || k == KID_G_LATC) continue; h->err = "static grid field missing"; return KID_EINVAL; }
    KID_HIP(h, hipMemcpyAsync(h->d_static[k], fields[k], h->ncell * sizeof(double), hipMemcpyHostToDevice, h->stream));
  }
  h->have_static = true;
  int rc = pack_static(h);
  if (rc) return rc;
  const double *none[KID_NFORCING] = {};
  rc = pack_forcing(h, none);
  if (rc) return rc;
  KID_HIP(h, hipStreamSynchronize(h->stream));
  return KID_OK;
}

int kid_set_forcing(kid_handle *h, const double *const fields[KID_NFORCING]) {
  if (!h || !fields) return KID_EINVAL;
  if (!h->have_static) { h->err = "kid_set_static_grid must be called first"; return KID_EINVAL; }
  KID_HIP(h, hipSetDevice(h->device));
  for (int k = 0; k < KID_NFORCING; ++k) {
    if (!fields[k]) continue;  // NULL keeps the previous plane
    KID_HIP(h, hipMemcpyAsync(h->d_forcing[k], fields[k], h->ncell * sizeof(double), hipMemcpyHostToDevice, h->stream));
  }
  h->have_forcing = true;
  { bool all = true; for (int k = 0; k < KID_NFORCING; ++k) all = all && (fields[k] || h->have_planes); h->have_planes = all; }
  const double *none[KID_NFORCING] = {};
  int rc = pack_forcing(h, none);
  if (rc) return rc;
  KID_HIP(h, hipStreamSynchronize(h->stream));  // the host arrays may be reused by the caller
  return KID_OK;
}

int kid_set_forcing_device(kid_handle *h, const double *const fields[KID_NFORCING]) {
  if (!h || !fields) return KID_EINVAL;
  if (!h->have_static) { h->err = "kid_set_static_grid must be called first"; return KID_EINVAL; }
  KID_HIP(h, hipSetDevice(h->device));
  // The per-cell records are built straight from the caller's planes; only ssh is also kept as a plane (the
  // grounding_fraction path of the mass spreading reads it, IB:3942).
  if (fields[KID_F_SSH])
    KID_HIP(h, hipMemcpyAsync(h->d_forcing[KID_F_SSH], fields[KID_F_SSH], h->ncell * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  h->have_forcing = true;
  return pack_forcing(h, fields);
}

// kid_set_forcing_device + kid_zero_accumulators (+ the reset of this step's redo counters) as ONE per-cell launch
int kid_step_prepare(kid_handle *h, const double *const fields[KID_NFORCING]) {
  if (!h) return KID_EINVAL;
  if (!h->have_static) { h->err = "kid_set_static_grid must be called first"; return KID_EINVAL; }
  if (!fields && !h->have_forcing) { h->err = "kid_set_forcing must be called before stepping"; return KID_EINVAL; }
  KID_HIP(h, hipSetDevice(h->device));
  const double *none[KID_NFORCING] = {};
  const double *const *src = fields ? fields : none;
  h->redo_parity ^= 1;
  if (h->side_mode == 2) {  // slow-lane schedule: the counter this prepass zeroes is the one the last carry-over launch read
    h->redo_parity = h->lane_step & 1;
    if (h->evC_live) { KID_HIP(h, hipStreamWaitEvent(h->stream, h->evC, 0)); }
  }
  PrepExtras ex;
  ex.ssh_copy = (src[KID_F_SSH] && src[KID_F_SSH] != h->d_forcing[KID_F_SSH]) ? h->d_forcing[KID_F_SSH] : nullptr;
  ex.acc = h->d_acc; ex.zero_planes = nacc_active(h);
  ex.cnt0 = h->d_redo_cnt[0][h->redo_parity]; ex.cnt1 = h->d_redo_cnt[1][h->redo_parity];
  h->have_forcing = true;
  const int rc = pack_forcing(h, src, ex);
  if (rc) return rc;
  h->acc_prezeroed = true; h->redo_prezeroed = true;
  return KID_OK;
}

#include "kid_ingest.inc"

int kid_upload_bergs(kid_handle *h, const kid_berg_soa *host) {
  if (!h || !host || host->n < 0) return KID_EINVAL;
  if (host->n > h->capacity) { h->err = "more bergs than capacity"; return KID_ECAPACITY; }
  KID_HIP(h, hipSetDevice(h->device));
  { const int rc_j = lanes_drain(h); if (rc_j) return rc_j; }
  h->static_rows_n = -1;   // the cached order by the static `inorder` keys belongs to the previous population
  const size_t n = (size_t)host->n;
  bool any_static = false, any_fl = false;
  for (int f = 0; f < KID_NB_F64; ++f) {
    if (host->f64[f]) KID_HIP(h, hipMemcpyAsync(h->bp.f[f], host->f64[f], n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    else KID_HIP(h, hipMemsetAsync(h->bp.f[f], 0, n * sizeof(double), h->stream));
  }
  for (int f = 0; f < KID_NB_I32; ++f) {
    if (host->i32[f]) KID_HIP(h, hipMemcpyAsync(h->bp.i[f], host->i32[f], n * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    else if (f == KID_BI_ALIVE) { if (n) hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->bp.i[f], 1, (long long)n); }
    else KID_HIP(h, hipMemsetAsync(h->bp.i[f], 0, n * sizeof(int32_t), h->stream));
  }
  if (host->id) KID_HIP(h, hipMemcpyAsync(h->bp.id, host->id, n * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
  else KID_HIP(h, hipMemsetAsync(h->bp.id, 0, n * sizeof(int64_t), h->stream));
  // cell indices are trusted by the kernels: reject anything outside the computational domain + first halo row
  if (host->i32[KID_BI_INE] && host->i32[KID_BI_JNE]) {
    for (size_t k = 0; k < n; ++k) {
      const int i = host->i32[KID_BI_INE][k], j = host->i32[KID_BI_JNE][k];
      if (i < h->gd.isc - 1 || i > h->gd.iec + 1 || j < h->gd.jsc - 1 || j > h->gd.jec + 1) {
        h->err = "berg cell index (ine,jne) outside the computational domain"; return KID_EINVAL;
      }
    }
  } else if (n > 0) { h->err = "ine/jne are required"; return KID_EINVAL; }
  // which optional per-berg fields can matter at all (avoids streaming all-zero arrays through the kernel)
  for (size_t k = 0; k < n; ++k) {
    if (host->f64[KID_B_STATIC_BERG] && host->f64[KID_B_STATIC_BERG][k] != 0.) any_static = true;
    if (host->f64[KID_B_HALO_BERG] && host->f64[KID_B_HALO_BERG][k] != 0.) any_static = true;
    if (host->f64[KID_B_MASS_OF_FL_BITS] && host->f64[KID_B_MASS_OF_FL_BITS][k] != 0.) any_fl = true;
    if (host->f64[KID_B_MASS_OF_FL_BERGY_BITS] && host->f64[KID_B_MASS_OF_FL_BERGY_BITS][k] != 0.) any_fl = true;
    if (host->f64[KID_B_FL_K] && host->f64[KID_B_FL_K][k] != 0.) any_fl = true;
  }
  h->flags.has_static = any_static ? 1 : 0;
  h->flags.has_fl = (any_fl || h->params.footloose) ? 1 : 0;
  for (int f = 0; f < KID_NB_F64; ++f) {
    bool nz = false;
    if (host->f64[f]) for (size_t k = 0; k < n && !nz; ++k) nz = host->f64[f][k] != 0.;
    h->uploaded_nonzero[f] = nz;
  }
  h->tail_valid = false; h->env_ever_stored = false;
  if (h->bp.orient) { h->bp.orient = nullptr; h->tables_dirty = true; }
  h->n = host->n;
  h->visited = false; h->have_bonds = false;
  KID_HIP(h, hipStreamSynchronize(h->stream));
  return KID_OK;
}

int kid_download_bergs(kid_handle *h, kid_berg_soa *host) {
  if (!h || !host) return KID_EINVAL;
  if (host->n < h->n) { h->err = "host SoA too small"; return KID_EINVAL; }
  KID_HIP(h, hipSetDevice(h->device));
  { const int rc_j = join_side(h); if (rc_j) return rc_j; }
  const size_t n = (size_t)h->n;
  for (int f = 0; f < KID_NB_F64; ++f)
    if (host->f64[f]) KID_HIP(h, hipMemcpyAsync(host->f64[f], h->bp.f[f], n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  for (int f = 0; f < KID_NB_I32; ++f)
    if (host->i32[f]) KID_HIP(h, hipMemcpyAsync(host->i32[f], h->bp.i[f], n * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  if (host->id) KID_HIP(h, hipMemcpyAsync(host->id, h->bp.id, n * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  KID_HIP(h, hipStreamSynchronize(h->stream));
  host->n = h->n;
  return KID_OK;
}

// which rows are occupied, for tail drops, compaction and re-binning: `alive`, plus (decomposed runs) the rows still waiting
// to be packed for a neighbour
static int layout_flags(kid_handle *h, const int32_t **flags) {
  *flags = h->bp.i[KID_BI_ALIVE];
  if (!h->mig_mode || h->n == 0) return KID_OK;
  if (!h->d_keep) KID_HIP(h, hipMalloc(&h->d_keep, (size_t)h->capacity * sizeof(int32_t)));
  const KeepSel ks{h->gd.isc, h->gd.iec, h->gd.jsc, h->gd.jec, h->flags.has_static ? h->bp.f[KID_B_HALO_BERG] : nullptr};
  hipLaunchKernelGGL(keep_flags_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, h->stream, h->bp.i[KID_BI_ALIVE], h->bp.i[KID_BI_INE], h->bp.i[KID_BI_JNE], ks, h->d_keep, (long long)h->n);
  KID_HIP(h, hipGetLastError());
  *flags = h->d_keep;
  return KID_OK;
}

int kid_num_bergs(kid_handle *h, int64_t *n_slots, int64_t *n_alive) {
  if (!h) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  { const int rc_j = join_side(h); if (rc_j) return rc_j; }
  if (n_slots) *n_slots = h->n;
  if (n_alive) {
    unsigned long long cnt = 0;
    KID_HIP(h, hipMemsetAsync(h->d_count, 0, sizeof(cnt), h->stream));
    if (h->n > 0) hipLaunchKernelGGL(count_alive_kernel, dim3((unsigned)std::min<long long>((h->n + 255) / 256, 1024)), dim3(256), 0, h->stream, h->bp.i[KID_BI_ALIVE], (long long)h->n, h->d_count);
    KID_HIP(h, hipMemcpyAsync(&cnt, h->d_count, sizeof(cnt), hipMemcpyDeviceToHost, h->stream));
    KID_HIP(h, hipStreamSynchronize(h->stream));
    *n_alive = (int64_t)cnt;
    if (h->tail_valid && h->mig_mode && (int64_t)cnt < h->n) {   // decomposed run: the tail is what is neither alive nor waiting to be packed
      const int32_t *flags = nullptr;
      { const int rc = layout_flags(h, &flags); if (rc) return rc; }
      KID_HIP(h, hipMemsetAsync(h->d_count, 0, sizeof(cnt), h->stream));
      hipLaunchKernelGGL(count_alive_kernel, dim3((unsigned)std::min<long long>((h->n + 255) / 256, 1024)), dim3(256), 0, h->stream, flags, (long long)h->n, h->d_count);
      KID_HIP(h, hipMemcpyAsync(&cnt, h->d_count, sizeof(cnt), hipMemcpyDeviceToHost, h->stream));
      KID_HIP(h, hipStreamSynchronize(h->stream));
    }
    if (h->tail_valid && (int64_t)cnt < h->n) {  // a re-binning left the dead at the tail: drop them now that the count is known
      h->n = (int64_t)cnt;
      if (n_slots) *n_slots = h->n;
    }
  }
  return KID_OK;
}

int kid_compact_bergs(kid_handle *h) {
  if (!h) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  { const int rc_j = lanes_drain(h); if (rc_j) return rc_j; }
  if (h->n == 0) return KID_OK;
  h->static_rows_n = -1;   // rows move: the cached traversal order (mts_build_order) is of the old rows
  const long long n = h->n;
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (!h->d_spare_f64) {
    KID_HIP(h, hipMalloc(&h->d_spare_f64, (size_t)h->capacity * sizeof(double)));
    KID_HIP(h, hipMalloc(&h->d_pos, (size_t)h->capacity * sizeof(unsigned)));
    KID_HIP(h, hipMalloc(&h->d_flag, (size_t)h->capacity * sizeof(unsigned)));
    size_t tmp = 0;
    KID_HIP(h, rocprim::exclusive_scan(nullptr, tmp, h->d_flag, h->d_pos, 0u, (size_t)h->capacity, rocprim::plus<unsigned>(), h->stream));
    h->scan_tmp_bytes = tmp;
    KID_HIP(h, hipMalloc(&h->d_scan_tmp, tmp));
  }
  const int32_t *live = nullptr;   // alive, or alive + waiting to be packed (decomposed runs)
  { const int rc_l = layout_flags(h, &live); if (rc_l) return rc_l; }
  hipLaunchKernelGGL(alive_to_u32_kernel, dim3(nb), dim3(256), 0, h->stream, live, h->d_flag, n);
  size_t tmp = h->scan_tmp_bytes;
  KID_HIP(h, rocprim::exclusive_scan(h->d_scan_tmp, tmp, h->d_flag, h->d_pos, 0u, (size_t)n, rocprim::plus<unsigned>(), h->stream));
  int64_t n_alive = 0;
  // (the count of rows that stay)
  {
    unsigned long long cnt = 0;
    KID_HIP(h, hipMemsetAsync(h->d_count, 0, sizeof(cnt), h->stream));
    hipLaunchKernelGGL(count_alive_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 1024)), dim3(256), 0, h->stream, live, n, h->d_count);
    KID_HIP(h, hipMemcpyAsync(&cnt, h->d_count, sizeof(cnt), hipMemcpyDeviceToHost, h->stream));
    KID_HIP(h, hipStreamSynchronize(h->stream));
    n_alive = (int64_t)cnt;   // rows that stay
  }
  for (int f = 0; f < KID_NB_F64; ++f) {
    hipLaunchKernelGGL(compact_f64_kernel, dim3(nb), dim3(256), 0, h->stream, h->bp.f[f], h->d_spare_f64, live, h->d_pos, n);
    std::swap(h->bp.f[f], h->d_spare_f64);
  }
  {
    int64_t *spare = (int64_t *)h->d_spare_f64;  // same element size
    hipLaunchKernelGGL(compact_i64_kernel, dim3(nb), dim3(256), 0, h->stream, h->bp.id, spare, live, h->d_pos, n);
    double *old = (double *)h->bp.id; h->bp.id = spare; h->d_spare_f64 = old;
  }
  const bool keeps_dead = (live != h->bp.i[KID_BI_ALIVE]);   // rows waiting to be packed stay dead: their flag moves with them
  for (int f = 0; f < KID_NB_I32; ++f) {
    if (f == KID_BI_ALIVE && !keeps_dead) continue;
    int32_t *spare = (int32_t *)h->d_flag;  // reuse the flag buffer as int32 spare
    hipLaunchKernelGGL(compact_i32_kernel, dim3(nb), dim3(256), 0, h->stream, h->bp.i[f], spare, live, h->d_pos, n);
    unsigned *old = (unsigned *)h->bp.i[f]; h->bp.i[f] = spare; h->d_flag = old;
  }
  if (!keeps_dead && n_alive > 0) hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)((n_alive + 255) / 256)), dim3(256), 0, h->stream, h->bp.i[KID_BI_ALIVE], 1, (long long)n_alive);
  KID_HIP(h, hipGetLastError());
  KID_HIP(h, hipStreamSynchronize(h->stream));
  h->n = n_alive;
  h->tables_dirty = true;
  return KID_OK;
}

// Fields no kernel of the current configuration ever stores to: if such a field was uploaded as zeros it is zeros
// forever and a re-binning need not move it (config 2: 32 of the 52 fp64 fields).  Conservative: anything that can
// append bergs (footloose) or the MTS path writes everything.
static bool field_never_written(const kid_handle *h, int f) {
  const kid_params &p = h->params;
  if (p.footloose || p.mts || h->calving_on) return false;
  switch (f) {
    case KID_B_AXN_FAST: case KID_B_AYN_FAST: case KID_B_BXN_FAST: case KID_B_BYN_FAST: case KID_B_ANG_VEL: case KID_B_ANG_ACCEL: case KID_B_ROT:
    case KID_B_UVEL_OLD: case KID_B_VVEL_OLD: case KID_B_LON_OLD: case KID_B_LAT_OLD:
      return p.periodic_reentry == 0;   // a berg that re-enters across the seam gets them reset (FW:3573-3577)
    case KID_B_HALO_BERG: case KID_B_STATIC_BERG: case KID_B_START_LON: case KID_B_START_LAT: case KID_B_START_MASS: case KID_B_HEAT_DENSITY:
      return true;
    case KID_B_UVEL_PREV: case KID_B_VVEL_PREV:
      return p.Runge_not_Verlet != 0;
    case KID_B_MASS_OF_FL_BITS: case KID_B_MASS_OF_FL_BERGY_BITS: case KID_B_FL_K: case KID_B_START_DAY: case KID_B_MASS_SCALING:
      return h->flags.has_fl == 0;
    case KID_B_UO: case KID_B_VO: case KID_B_UI: case KID_B_VI: case KID_B_UA: case KID_B_VA: case KID_B_SSH_X: case KID_B_SSH_Y:
    case KID_B_SST: case KID_B_SSS: case KID_B_CN: case KID_B_HI: case KID_B_OD:
      return !h->env_ever_stored;
    default:
      return false;
  }
}

// move_berg_between_cells (IB:5437, FW:1758-1797): re-bin the bergs after they moved.  With per-cell linked lists
// that is list surgery; on the SoA it is a device counting sort by cell index (j-major, i-minor: the reference's
// traversal order, IB:7106) followed by ONE launch that gathers every field into a second set of arrays, which then
// becomes the SoA.  Dead bergs sort to the end and are dropped.  The order inside a cell is arbitrary (the bonded /
// MTS path, where it matters, builds its own order every step and never re-bins); KID_STABLE_RESORT=1 in the
// environment selects a stable comparison sort instead.
// Correctness never depends on it (the kernels accept any order); speed does: the hot build shares LDS cell
// packets and atomics between the lanes of a wave that sit in the same cell.
static int rebin_core(kid_handle *h, bool with_lane, int *list, const int *list_count);
int kid_move_berg_between_cells(kid_handle *h) {
  if (!h) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  { const int rc_j = lanes_drain(h); if (rc_j) return rc_j; }
  h->steps_since_sort = 0;
  if (h->n == 0) return KID_OK;
  if (h->have_bonds) { h->err = "bergs with bonds keep their rows: no re-binning while bond tables exist"; return KID_EUNSUPPORTED; }
  const int rc = rebin_core(h, false, nullptr, nullptr);
  if (rc) return rc;
  // the dead sorted to the tail; they are dropped the next time the host asks for the count (no synchronisation here)
  h->tail_valid = true;
  return KID_OK;
}
// with_lane: the lane array moves with the rows and `list` (row numbers, *list_count of them) is translated
static int rebin_core(kid_handle *h, bool with_lane, int *list, const int *list_count) {
  h->static_rows_n = -1;   // rows move: the cached traversal order (mts_build_order) is of the old rows
  const long long n = h->n;
  const unsigned nb = (unsigned)((n + 255) / 256);
  const unsigned dead_key = (unsigned)h->ncell;  // larger than any cell index
  if (!h->d_key[0]) {
    h->stable_resort = h->dbg.stable_resort;  // a stable comparison sort keeps the row order inside a cell
    for (int q = 0; q < 2; ++q) {
      KID_HIP(h, hipMalloc(&h->d_key[q], (size_t)h->capacity * sizeof(unsigned)));
      KID_HIP(h, hipMalloc(&h->d_idx[q], (size_t)h->capacity * sizeof(unsigned)));
    }
    size_t tmp = 0;
    KID_HIP(h, rocprim::radix_sort_pairs(nullptr, tmp, h->d_key[0], h->d_key[1], h->d_idx[0], h->d_idx[1], (size_t)h->capacity, 0u, 32u, h->stream));
    h->sort_tmp_bytes = tmp;
    KID_HIP(h, hipMalloc(&h->d_sort_tmp, tmp));
    KID_HIP(h, hipMalloc(&h->d_cell_hist, ((size_t)h->ncell + 1) * sizeof(unsigned)));
    tmp = 0;
    KID_HIP(h, rocprim::exclusive_scan(nullptr, tmp, h->d_cell_hist, h->d_cell_hist, 0u, (size_t)h->ncell + 1, rocprim::plus<unsigned>(), h->stream));
    h->cscan_tmp_bytes = tmp;
    KID_HIP(h, hipMalloc(&h->d_cscan_tmp, tmp ? tmp : 8));
    for (int f = 0; f < KID_NB_F64; ++f) KID_HIP(h, hipMalloc(&h->bp_alt.f[f], (size_t)h->capacity * sizeof(double)));
    for (int f = 0; f < KID_NB_I32; ++f) KID_HIP(h, hipMalloc(&h->bp_alt.i[f], (size_t)h->capacity * sizeof(int32_t)));
    KID_HIP(h, hipMalloc(&h->bp_alt.id, (size_t)h->capacity * sizeof(int64_t)));
  }
  const int32_t *live = nullptr;   // alive, or alive + waiting to be packed (decomposed runs): those keep their cell as key
  { const int rc_l = layout_flags(h, &live); if (rc_l) return rc_l; }
  const unsigned *perm = nullptr;
  if (h->stable_resort) {
    unsigned bits = 1; while ((1ull << bits) <= (unsigned long long)dead_key) ++bits;
    hipLaunchKernelGGL(cell_key_kernel, dim3(nb), dim3(256), 0, h->stream, h->bp.i[KID_BI_INE], h->bp.i[KID_BI_JNE], live,
                       h->gd.isd, h->gd.jsd, h->ni, dead_key, h->d_key[0], h->d_idx[0], n);
    size_t tmp = h->sort_tmp_bytes;
    KID_HIP(h, rocprim::radix_sort_pairs(h->d_sort_tmp, tmp, h->d_key[0], h->d_key[1], h->d_idx[0], h->d_idx[1], (size_t)n, 0u, bits, h->stream));
    perm = h->d_idx[1];
  } else {
    KID_HIP(h, hipMemsetAsync(h->d_cell_hist, 0, ((size_t)h->ncell + 1) * sizeof(unsigned), h->stream));
    hipLaunchKernelGGL(cell_rank_kernel, dim3(nb), dim3(256), 0, h->stream, h->bp.i[KID_BI_INE], h->bp.i[KID_BI_JNE], live,
                       h->gd.isd, h->gd.jsd, h->ni, dead_key, h->d_key[0], h->d_key[1], h->d_cell_hist, n);
    size_t tmp = h->cscan_tmp_bytes;
    KID_HIP(h, rocprim::exclusive_scan(h->d_cscan_tmp, tmp, h->d_cell_hist, h->d_cell_hist, 0u, (size_t)h->ncell + 1, rocprim::plus<unsigned>(), h->stream));
    hipLaunchKernelGGL(cell_place_kernel, dim3(nb), dim3(256), 0, h->stream, h->d_key[0], h->d_key[1], h->d_cell_hist, h->d_idx[1], h->d_idx[0], n);
    perm = h->d_idx[1];
  }
  PermTable t{};
  int moved_f[KID_NB_F64], nmf = 0;
  for (int f = 0; f < KID_NB_F64; ++f) {
    if (!h->uploaded_nonzero[f] && field_never_written(h, f)) continue;  // all zeros, before and after
    t.src[t.n8] = h->bp.f[f]; t.dst[t.n8] = h->bp_alt.f[f]; ++t.n8; moved_f[nmf++] = f;
  }
  t.src[t.n8] = h->bp.id; t.dst[t.n8] = h->bp_alt.id; ++t.n8;
  for (int f = 0; f < KID_NB_I32; ++f) { t.src[t.n8 + t.n4] = h->bp.i[f]; t.dst[t.n8 + t.n4] = h->bp_alt.i[f]; ++t.n4; }
  if (with_lane) {
    if (h->stable_resort) { h->err = "KID_STABLE_RESORT has no inverse permutation for the slow-lane re-binning"; return KID_EUNSUPPORTED; }
    if (!h->d_lane_alt) { KID_HIP(h, hipMalloc(&h->d_lane_alt, (size_t)h->capacity * sizeof(int))); KID_HIP(h, hipMemsetAsync(h->d_lane_alt, 0, (size_t)h->capacity * sizeof(int), h->stream)); }
    t.src[t.n8 + t.n4] = h->d_lane; t.dst[t.n8 + t.n4] = h->d_lane_alt; ++t.n4;
  }
  hipLaunchKernelGGL(permute_all_kernel, dim3(nb), dim3(256), 0, h->stream, t, perm, n);
  if (with_lane) {
    std::swap(h->d_lane, h->d_lane_alt);
    if (list) hipLaunchKernelGGL(translate_list_kernel, dim3(64), dim3(256), 0, h->stream, list, list_count, h->d_idx[0]);
  }
  for (int q = 0; q < nmf; ++q) std::swap(h->bp.f[moved_f[q]], h->bp_alt.f[moved_f[q]]);
  std::swap(h->bp.id, h->bp_alt.id);
  for (int f = 0; f < KID_NB_I32; ++f) std::swap(h->bp.i[f], h->bp_alt.i[f]);
  KID_HIP(h, hipGetLastError());
  if (h->tables_dirty) return refresh_tables(h);
  // only the field pointers changed: one table, not four (a launch on the side stream may still be reading it)
  { const int rc = join_side(h); if (rc) return rc; }
  hipLaunchKernelGGL(set_berg_table_kernel, dim3(1), dim3(64), 0, h->stream, h->bp, h->d_bp);
  KID_HIP(h, hipGetLastError());
  return KID_OK;
}
int kid_set_resort_interval(kid_handle *h, int steps) {
  if (!h || steps < 0) return KID_EINVAL;
  h->resort_interval = steps;
  return KID_OK;
}

// With .not.old_interp_flds_order the stored environment (berg%uo ... od) is an input of evolve_icebergs and thermodynamics as
// the reference calls them, one after the other; the fused step (kid_run_step / kid_step_local) interpolates it for itself
// and only writes it -- 104 B per berg-step that nothing inside the library reads back.  Switching the store off is
// therefore allowed there too; the entry points that do read it then refuse to run (launch_berg) instead of reading values
// of some earlier step.
int kid_set_store_environment(kid_handle *h, int on) {
  if (!h) return KID_EINVAL;
  h->flags.store_env = on ? 1 : 0;
  h->env_off_requested = !on;
  return KID_OK;
}

int kid_zero_accumulators(kid_handle *h) {
  if (!h) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  const size_t planes = (size_t)nacc_active(h);
  KID_HIP(h, hipMemsetAsync(h->d_acc, 0, planes * h->ncell * sizeof(double), h->stream));
  // the KID_NSCALAR words behind the planes hold this step's increments of the running totals the reference
  // keeps on `bergs` (net_heat_to_ocean, nbergs_melted, ...); the gather folds them into the totals and clears them.
  return KID_OK;
}

}  // extern "C"

static int refresh_tables(kid_handle *h);
// May the hot build be the plain one (berg_kernel<..., K = 1>, kid_device.hpp)?  Only when every switch of KID_SWITCHES
// has the value that build folds in, and none of the handle's flags is set.
static bool plain_namelist(const kid_handle *h) {
  const kid_params &p = h->params;
  bool same = true;
#define KID_X(name, plain, flp) same = same && (p.name == (decltype(p.name))(plain));
  KID_SWITCHES(KID_X)
#undef KID_X
  const Flags &f = h->flags;
  return same && h->gd.grid_is_latlon && !p.pass_fields_to_ocean_model && !f.has_static && !f.has_fl && !f.footprint && !f.no_diag &&
         !h->dbg.no_plain_build;
}
// ... and K = 2 with the footloose profile's (Cartesian grid, footloose state, no static bergs, no footprint planes)
static bool fl_profile_namelist(const kid_handle *h) {
  const kid_params &p = h->params;
  bool same = true;
#define KID_X(name, plain, flp) same = same && (p.name == (decltype(p.name))(flp));
  KID_SWITCHES(KID_X)
#undef KID_X
  const Flags &f = h->flags;
  return same && !h->gd.grid_is_latlon && !p.pass_fields_to_ocean_model && !f.has_static && f.has_fl && !f.footprint && !f.no_diag && !h->dbg.no_plain_build;
}
template <unsigned PH>
static int launch_berg(kid_handle *h, long long range_k0 = 0, long long range_len = -1) {   // range: the rows to step (default all)
  if (!h->have_forcing) { h->err = "kid_set_forcing must be called before stepping"; return KID_EINVAL; }
  if (h->n == 0 || range_len == 0) return KID_OK;
  const bool rk = h->params.Runge_not_Verlet != 0, old = h->params.old_interp_flds_order != 0;
  if (!old && !(PH & PH_INTERP) && (PH & (PH_EVOLVE | PH_THERMO)) && !h->flags.store_env) {
    h->err = "this entry point reads the bergs' stored environment, which kid_set_store_environment has switched off (use kid_run_step, or switch it on)";
    return KID_EINVAL;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
{ int rc_t = lanes_drain(h); if (rc_t) return rc_t; }
{ int rc_t = refresh_tables(h); if (rc_t) return rc_t; }
  const DevGrid *gtab = h->forc_parity ? h->d_grid2 : h->d_grid;
  h->tail_valid = false;
  if (h->flags.store_env || !old) h->env_ever_stored = true;
  // pass 1: every berg through the specialised build; pass 2: the general build over the bergs pass 1 queued
  // (a few per cent: cell crossings, coast bounces, polar cells).  Pass 2 is sized for the worst case and its
  // surplus workgroups exit on the device-side count.
  // Pipelined (a side stream is set): two halves; the general build of a half runs on the side stream while the main
  // stream is already in the hot build of the other half (or of the next step): the ~85 us single-wave latency of the
  // general build leaves the critical path.  Events order a half's hot build behind its own previous general build.
  const bool plain = plain_namelist(h);   // the hot build of the default namelist carries none of the other branches
  const bool flprof = fl_profile_namelist(h);   // ... nor does the one of the footloose profile
  const int nparts = (h->pipelined && !h->params.mts && !h->params.footloose && h->n >= 4096 && range_len < 0) ? 2 : 1;
  const long long half = ((h->n / 2 + 255) / 256) * 256;
  for (int part = 0; part < nparts; ++part) {
    const long long k0 = (nparts == 1) ? range_k0 : (part == 0 ? 0 : half);
    const long long klen = (nparts == 1) ? (range_len < 0 ? h->n : range_len) : (part == 0 ? half : h->n - half);
    const unsigned nbp = (unsigned)((klen + KID_HOT_WG - 1) / KID_HOT_WG);
    const Redo redo{part == 0 ? h->d_redo_list : h->d_redo_list2, h->d_redo_cnt[part][h->redo_parity], k0, klen, nullptr, 0,
                    h->d_fl_cursor, h->d_iceberg_counter, (long long)h->capacity, h->gd.iec - h->gd.isc + 1, h->fl_step};
    hipStream_t gs = (nparts == 2) ? h->side_stream : h->stream;
    if (h->evG_live[part]) KID_HIP(h, hipStreamWaitEvent(h->stream, h->evG[part], 0));
    if (nparts == 1 && h->evG_live[1]) KID_HIP(h, hipStreamWaitEvent(h->stream, h->evG[1], 0));
    if (!h->redo_prezeroed) KID_HIP(h, hipMemsetAsync(redo.count, 0, sizeof(int), h->stream));
    if (h->profile) {  // one (start, stop) pair around every hot-build launch
      KID_HIP(h, hipEventCreate(&e0)); KID_HIP(h, hipEventCreate(&e1));
      KID_HIP(h, hipEventRecord(e0, h->stream));
    }
#define KID_LAUNCH(RKV, OLDV)                                                                                                   \
  do {                                                                                                                          \
    if (PH == (PH_EVOLVE | PH_THERMO | PH_SPREAD) && RKV && OLDV && plain)                                                       \
      (void)kid::launch_hot_plain(h->flags.store_env ? 3 : 1, nbp, (void *)h->stream, gtab, h->d_params, h->d_bp, (long long)h->n, h->d_acc, h->ncell, &h->flags, &redo);  \
    else if (PH == (PH_INTERP | PH_EVOLVE | PH_FL | PH_THERMO | PH_SPREAD) && !RKV && !OLDV && flprof)                           \
      hipLaunchKernelGGL((berg_kernel<RKV, OLDV, PH, true, (PH == (PH_INTERP | PH_EVOLVE | PH_FL | PH_THERMO | PH_SPREAD) && !RKV && !OLDV) ? 2 : 0>), dim3(nbp), dim3(KID_HOT_WG), 0, h->stream, gtab, h->d_params, h->d_bp, (long long)h->n, h->d_acc, h->ncell, h->flags, redo);  \
    else                                                                                                                        \
    hipLaunchKernelGGL((berg_kernel<RKV, OLDV, PH, true>), dim3(nbp), dim3(KID_HOT_WG), 0, h->stream, gtab, h->d_params, h->d_bp, (long long)h->n, h->d_acc, h->ncell, h->flags, redo);  \
    if (h->profile) { (void)hipEventRecord(e1, h->stream); h->pending.emplace_back(e0, e1); h->berg_launches++; } /* the timed kernel is the hot build (pass 1) */ \
    if (nparts == 2) { (void)hipEventRecord(h->evF[part], h->stream); (void)hipStreamWaitEvent(gs, h->evF[part], 0); }          \
    hipLaunchKernelGGL((berg_kernel<RKV, OLDV, PH, false>), dim3((unsigned)std::min<long long>((klen + 63) / 64, 2048)), dim3(64), 0, gs, gtab, h->d_params, h->d_bp, (long long)h->n, h->d_acc, h->ncell, h->flags, redo); \
    if (nparts == 2) { (void)hipEventRecord(h->evG[part], gs); h->evG_live[part] = true; } else h->evG_live[part] = false;      \
  } while (0)
    if (rk && old) KID_LAUNCH(true, true);
    else if (rk && !old) KID_LAUNCH(true, false);
    else if (!rk && old) KID_LAUNCH(false, true);
    else KID_LAUNCH(false, false);
#undef KID_LAUNCH
  }
  if (nparts == 1) h->evG_live[1] = false;
  h->redo_prezeroed = false;
  KID_HIP(h, hipGetLastError());
  return KID_OK;
}

// The "slow lane" schedule of the fused step (kid_set_side_stream mode 2).  The general build is a single-wave latency
// (~65 us for the few per cent of bergs that cross a cell edge or bounce) during which the chip idles.  Here it never
// sits between two hot builds: a berg the hot build of step s hands over is stepped by general-build launches on the
// side stream for steps s and s + 1 and returns to the hot build at s + 2:
//   main: prepass(s) | hot(s) ..................................... | prepass(s+1) | hot(s+1) ...
//   side:            | carry(s): step s of the bergs handed over at s-1 | new(s): step s of those handed over at s | gather(s)
// carry(s) starts with hot(s) and is long finished when hot(s+1) starts (event evC, normally already signalled);
// new(s) runs under hot(s+1), which skips its bergs (lane[k] >= s+1).  Per-step sums go to the accumulator block bound
// at launch; the caller alternates two blocks and launches the gather on the side stream (PipelinedStepper).
static bool lanes_eligible(const kid_handle *h) {
  const kid_params &p = h->params;
  return h->side_mode == 2 && h->side_stream && !p.static_icebergs && !p.mts && !p.interactive_icebergs_on &&
         !p.footloose && !(p.grounding_fraction > 0.) && !p.find_melt_using_spread_mass && h->n >= 4096;
}
template <bool OLDV>
static int launch_berg_lanes(kid_handle *h) {
  constexpr unsigned PH = OLDV ? (PH_EVOLVE | PH_THERMO | PH_SPREAD) : (PH_INTERP | PH_EVOLVE | PH_THERMO | PH_SPREAD);
  if (!h->have_forcing) { h->err = "kid_set_forcing must be called before stepping"; return KID_EINVAL; }
  { int rc_t = refresh_tables(h); if (rc_t) return rc_t; }
  const bool rk = h->params.Runge_not_Verlet != 0;
  const DevGrid *gtab = h->forc_parity ? h->d_grid2 : h->d_grid;
  hipStream_t M = h->stream, S = h->side_stream;
  const int s = h->lane_step, par = s & 1;
  h->tail_valid = false;
  if (h->flags.store_env || !OLDV) h->env_ever_stored = true;
  h->lanes_active = true;
  for (int q = 0; q < 2; ++q) if (h->evG_live[q] && h->pipelined) { KID_HIP(h, hipStreamWaitEvent(M, h->evG[q], 0)); h->evG_live[q] = false; }
  if (!h->redo_prezeroed) {  // no prepass: zero this step's counter here (it was last read by the previous carry-over launch)
    if (h->evC_live) KID_HIP(h, hipStreamWaitEvent(M, h->evC, 0));
    KID_HIP(h, hipMemsetAsync(h->d_redo_cnt[0][par], 0, sizeof(int), M));
  }
  int *list_new = par ? h->d_redo_list2 : h->d_redo_list, *list_old = par ? h->d_redo_list : h->d_redo_list2;
  const Redo hot{list_new, h->d_redo_cnt[0][par], 0, h->n, h->d_lane, s};
  const Redo carry{list_old, h->d_redo_cnt[0][par ^ 1], 0, h->n, nullptr, 0};
  const unsigned nbp = (unsigned)((h->n + KID_HOT_WG - 1) / KID_HOT_WG), nbg = (unsigned)std::min<long long>((h->n + 63) / 64, 2048);
  // Every record / wait is a barrier packet of ~5 us on the stream it goes to: the main stream gets one wait (in the
  // prepass) and two records per step; with profiling on, the (start, stop) pair of the hot build doubles as the two.
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (h->profile) { KID_HIP(h, hipEventCreate(&e0)); KID_HIP(h, hipEventCreate(&e1)); }
  hipEvent_t evP = h->profile ? e0 : h->evP, evF = h->profile ? e1 : h->evF[0];
  KID_HIP(h, hipEventRecord(evP, M));  // forcing records and accumulator block of this step are ready
  KID_HIP(h, hipStreamWaitEvent(S, evP, 0));
#define KID_LAUNCH_LANES(RKV)                                                                                                   \
  do {                                                                                                                          \
    if (h->carry_valid)                                                                                                         \
      hipLaunchKernelGGL((berg_kernel<RKV, OLDV, PH, false>), dim3(nbg), dim3(64), 0, S, gtab, h->d_params, h->d_bp, (long long)h->n, h->d_acc, h->ncell, h->flags, carry); \
    /* recorded even without a carry-over launch: everything enqueued on the side stream so far (the gather of the step  \
       before last included) is complete once the next prepass has waited for it */                                          \
    (void)hipEventRecord(h->evC, S); h->evC_live = true;                                                                        \
    if (RKV && OLDV && plain)                                                                                                   \
      (void)kid::launch_hot_plain(h->flags.store_env ? 3 : 1, nbp, (void *)M, gtab, h->d_params, h->d_bp, (long long)h->n, h->d_acc, h->ncell, &h->flags, &hot); \
    else                                                                                                                        \
    hipLaunchKernelGGL((berg_kernel<RKV, OLDV, PH, true>), dim3(nbp), dim3(KID_HOT_WG), 0, M, gtab, h->d_params, h->d_bp, (long long)h->n, h->d_acc, h->ncell, h->flags, hot); \
    (void)hipEventRecord(evF, M);                                                                                               \
    if (h->profile) { h->pending.emplace_back(e0, e1); h->berg_launches++; }                                                    \
    if (rebin_now) {                                                                                                            \
      /* re-binning inside the step: between the hot build and the general build of the bergs it handed over, whose list  \
         and lanes move with the rows; nothing is drained and no latency is exposed (the carry-over launch of this step   \
         is long finished, the main stream only has to say so before it moves rows) */                                      \
      (void)hipStreamWaitEvent(M, h->evC, 0);                                                                                   \
      rebin_rc = rebin_core(h, true, list_new, h->d_redo_cnt[0][par]);                                                          \
      if (!h->evR) (void)hipEventCreateWithFlags(&h->evR, hipEventDisableTiming);                                               \
      (void)hipEventRecord(h->evR, M); (void)hipStreamWaitEvent(S, h->evR, 0);                                                  \
    } else (void)hipStreamWaitEvent(S, evF, 0);                                                                                 \
    hipLaunchKernelGGL((berg_kernel<RKV, OLDV, PH, false>), dim3(nbg), dim3(64), 0, S, gtab, h->d_params, h->d_bp, (long long)h->n, h->d_acc, h->ncell, h->flags, hot); \
    (void)hipEventRecord(h->evG[0], S); h->evG_live[0] = true;                                                                  \
  } while (0)
  const bool plain = plain_namelist(h);
  const bool rebin_now = h->resort_interval > 0 && ++h->steps_since_sort >= h->resort_interval && !h->have_bonds && !h->dbg.stable_resort;
  int rebin_rc = KID_OK;
  if (rk) KID_LAUNCH_LANES(true); else KID_LAUNCH_LANES(false);
#undef KID_LAUNCH_LANES
  if (rebin_rc) return rebin_rc;
  if (rebin_now) h->steps_since_sort = 0;
  h->carry_valid = true;
  h->lane_step = s + 1;
  h->redo_prezeroed = false;
  KID_HIP(h, hipGetLastError());
  return KID_OK;
}

extern "C" {

static int mts_depth(kid_handle *h);
static int bond_orientations(kid_handle *h);
int kid_evolve_icebergs_mts(kid_handle *h);
int kid_evolve_icebergs_interactive(kid_handle *h);
int kid_interp_gridded_fields_to_bergs(kid_handle *h) {
  if (!h) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  int rc = launch_berg<PH_INTERP>(h);
  if (rc || !h->params.mts) return rc;
  return mts_depth(h);
}
int kid_evolve_icebergs(kid_handle *h) {
  if (!h) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  if (h->params.static_icebergs) return KID_OK;  // IB:5428
  if (h->params.mts) return kid_evolve_icebergs_mts(h);  // IB:5431
  if (h->params.interactive_icebergs_on) return kid_evolve_icebergs_interactive(h);
  return launch_berg<PH_EVOLVE>(h);
}
int kid_thermodynamics(kid_handle *h) {
  if (!h) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  // IB:2872-2873
  KID_HIP(h, hipMemsetAsync(h->d_acc + (size_t)KID_A_UVEL_ON_OCEAN * h->ncell, 0, 18 * h->ncell * sizeof(double), h->stream));
  return launch_berg<PH_THERMO>(h);
}
static int refresh_tables(kid_handle *h) {
  if (h->tables_dirty) {
    const int rc = join_side(h);  // a general-build launch on the side stream may still be reading the tables
    if (rc) return rc;
    hipLaunchKernelGGL(set_berg_table_kernel, dim3(1), dim3(64), 0, h->stream, h->bp, h->d_bp);
    hipLaunchKernelGGL(set_params_kernel, dim3(1), dim3(64), 0, h->stream, h->params, h->d_params);
    { const int par = h->forc_parity;
      h->forc_parity = 0; hipLaunchKernelGGL(set_grid_kernel, dim3(1), dim3(64), 0, h->stream, dev_grid(h), h->d_grid);
      h->forc_parity = 1; hipLaunchKernelGGL(set_grid_kernel, dim3(1), dim3(64), 0, h->stream, dev_grid(h), h->d_grid2);
      h->forc_parity = par; }
    KID_HIP(h, hipGetLastError());
    h->tables_dirty = false;
  }
  return KID_OK;
}
// ids of the children a footloose pass appended to rows [n_old, n_old + m): the per-cell counter values in the order the
// reference's loop would have met the events (kid_footloose.hpp, fl_assign_ids_*)
// second half of calve_fl_icebergs for the children of one pass (kid_footloose.hpp): positions, copied members, constants
static int fl_place_children(kid_handle *h, long long n_old, int m, unsigned step) {
  if (!h->fl_place_warm) {   // the kernel's out-of-line helpers need stack scratch, which the runtime allocates at the FIRST launch
    // (1.6 s at 1e7 bergs' worth of device memory in use): an empty launch at the first footloose pass, not at the first event
    h->fl_place_warm = true;
    hipLaunchKernelGGL(fl_place_children_kernel<BergPtrs>, dim3(1), dim3(64), 0, h->stream, dev_grid(h), (const kid_params *)h->d_params, (const BergPtrs *)h->d_bp, n_old, 0, step);
    KID_HIP(h, hipGetLastError());
  }
  if (m <= 0) return KID_OK;
  hipLaunchKernelGGL(fl_place_children_kernel<BergPtrs>, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, h->stream, dev_grid(h), (const kid_params *)h->d_params, (const BergPtrs *)h->d_bp, n_old, m, step);
  KID_HIP(h, hipGetLastError());
  return KID_OK;
}
static int fl_assign_ids(kid_handle *h, long long n_old, int m) {
  if (m <= 0) return KID_OK;
  if (!h->d_fl_head) {
    KID_HIP(h, hipMalloc(&h->d_fl_head, h->ncell * sizeof(int32_t)));
    KID_HIP(h, hipMemsetAsync(h->d_fl_head, 0xff, h->ncell * sizeof(int32_t), h->stream));   // -1: empty lists
  }
  if (m > h->fl_ev_capacity) {
    if (h->d_fl_next) (void)hipFree(h->d_fl_next);
    if (h->d_fl_newid) (void)hipFree(h->d_fl_newid);
    h->fl_ev_capacity = std::max<long long>(2ll * m, 4096);
    KID_HIP(h, hipMalloc(&h->d_fl_next, (size_t)h->fl_ev_capacity * sizeof(int32_t)));
    KID_HIP(h, hipMalloc(&h->d_fl_newid, (size_t)h->fl_ev_capacity * sizeof(int64_t)));
  }
  const FlIdCtx x{h->d_fl_head, h->d_fl_next, h->d_fl_newid, h->d_iceberg_counter, n_old, m, id_iNg(h->gd), id_ij0(h->gd)};
  const dim3 grid((unsigned)((m + 255) / 256)), block(256);
  const DevGrid g = dev_grid(h);
  hipLaunchKernelGGL(fl_assign_ids_push<BergPtrs>, grid, block, 0, h->stream, g, (const BergPtrs *)h->d_bp, x);
  hipLaunchKernelGGL(fl_assign_ids_rank<BergPtrs>, grid, block, 0, h->stream, g, (const BergPtrs *)h->d_bp, x);
  hipLaunchKernelGGL(fl_assign_ids_store<BergPtrs>, grid, block, 0, h->stream, g, (const BergPtrs *)h->d_bp, x);
  KID_HIP(h, hipGetLastError());
  return KID_OK;
}
int kid_footloose_calving(kid_handle *h) {
  if (!h) return KID_EINVAL;
  if (!h->params.footloose || h->n == 0) return KID_OK;
  KID_HIP(h, hipSetDevice(h->device));
  int rc = refresh_tables(h);
  if (rc) return rc;
  h->flags.has_fl = 1;
  KID_HIP(h, hipMemsetAsync(h->d_fl_cursor, 0, sizeof(int), h->stream));
  const unsigned fl_step_now = h->fl_step;
  FlChildCtx cx{h->d_fl_cursor, h->d_iceberg_counter, (long long)h->n, (long long)h->capacity, h->gd.iec - h->gd.isc + 1, h->fl_step++};
  hipLaunchKernelGGL(footloose_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, h->stream, dev_grid(h), h->d_params, h->d_bp, cx, h->d_acc, h->ncell);
  KID_HIP(h, hipGetLastError());
  int appended = 0;  // the population grew: the host needs the new size before the next launch
  KID_HIP(h, hipMemcpyAsync(&appended, h->d_fl_cursor, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  KID_HIP(h, hipStreamSynchronize(h->stream));
  const int64_t room = h->capacity - h->n;
  if (appended > room) {
    h->n = h->capacity;
    h->err = "footloose calving ran out of capacity: create the handle with room for child bergs";
    return KID_ECAPACITY;
  }
  rc = fl_place_children(h, h->n, appended, fl_step_now);
  if (rc) return rc;
  rc = fl_assign_ids(h, h->n, appended);
  h->n += appended;
  return rc;
}
double kid_footloose_uniform(int32_t seed, int64_t berg_id, int64_t step, int32_t draw) {   // the number a child placement uses (host side of kid_rng.h)
  return kid_fl_uniform((uint32_t)seed, berg_id, (uint32_t)step, (uint32_t)draw);
}
int kid_set_footloose_step(kid_handle *h, int64_t step) {   // restart: continue the child-placement sequence (kid_rng.h)
  if (!h || step < 0) return KID_EINVAL;
  h->fl_step = (unsigned)step;
  return KID_OK;
}
int kid_get_footloose_step(kid_handle *h, int64_t *step) {
  if (!h || !step) return KID_EINVAL;
  *step = (int64_t)h->fl_step;
  return KID_OK;
}
int kid_set_iceberg_counter(kid_handle *h, const int32_t *counter) {  // grd%iceberg_counter_grd, (isd:ied,jsd:jed)
  if (!h || !counter) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  KID_HIP(h, hipMemcpy(h->d_iceberg_counter, counter, h->ncell * sizeof(int32_t), hipMemcpyHostToDevice));
  return KID_OK;
}
int kid_get_iceberg_counter(kid_handle *h, int32_t *counter) {
  if (!h || !counter) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  KID_HIP(h, hipStreamSynchronize(h->stream));
  KID_HIP(h, hipMemcpy(counter, h->d_iceberg_counter, h->ncell * sizeof(int32_t), hipMemcpyDeviceToHost));
  return KID_OK;
}

static int launch_gather(kid_handle *h) {
  { const int rc_j = join_side(h); if (rc_j) return rc_j; }
  const DevGrid g = dev_grid(h);
  const int ncomp = (h->gd.iec - h->gd.isc + 1) * (h->gd.jec - h->gd.jsc + 1);
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((ncomp + 255) / 256)), dim3(256), 0, h->stream, g, h->params, h->d_acc, h->d_out, h->ncell, h->d_totals,
                     (const double *)(h->params.find_melt_using_spread_mass ? h->d_spread_mass_old : nullptr),
                     (const double *)((h->params.find_melt_using_spread_mass && h->params.Iceberg_melt_without_decay) ? h->d_spread_mass_old + h->ncell : nullptr));
  KID_HIP(h, hipGetLastError());
  return KID_OK;
}
int kid_create_gridded_icebergs_fields(kid_handle *h) {
  if (!h) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  KID_HIP(h, hipMemsetAsync(h->d_acc + (size_t)KID_A_MASS_ON_OCEAN * h->ncell, 0, 36 * h->ncell * sizeof(double), h->stream));  // IB:4984-4987
  int rc = launch_berg<PH_SPREAD>(h);
  if (rc) return rc;
  return launch_gather(h);
}

#include "kid_mts_host.inc"
#include "kid_calving.inc"

int kid_step_local(kid_handle *h) {
  if (!h) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  int rc = h->acc_prezeroed ? KID_OK : kid_zero_accumulators(h);  // kid_step_prepare has done it for this step
  h->acc_prezeroed = false;
  if (rc) return rc;
  const kid_params &p = h->params;
  if (p.mts) {  // IB:5409-5512 with mts=T
    if (!h->visited) { rc = mts_first_visit(h); if (rc) return rc; }
    if (!p.static_icebergs) { rc = kid_evolve_icebergs_mts(h); if (rc) return rc; }
    rc = kid_interp_gridded_fields_to_bergs(h);   // IB:5458
    if (rc) return rc;
    rc = kid_set_conglom_ids(h);                  // transfer_mts_bergs, IB:5459
    if (rc) return rc;
    rc = bond_orientations(h);
    if (rc) return rc;
    return launch_berg<PH_THERMO | PH_SPREAD>(h);
  }
  if (p.interactive_icebergs_on) {  // single-time-step scheme with interactions, IB:5409-5512
    const bool contact = (p.contact_distance > 0.) || (p.contact_spring_coef != p.spring_coef);
    if (!h->visited) { rc = sts_ia_first_visit(h); if (rc) return rc; }
    if (!p.old_interp_flds_order) { rc = launch_berg<PH_INTERP>(h); if (rc) return rc; }
    if (!p.static_icebergs) { rc = kid_evolve_icebergs_interactive(h); if (rc) return rc; }
    if (contact) { rc = kid_set_conglom_ids(h); if (rc) return rc; }   // IB:5470-5471
    rc = bond_orientations(h);
    if (rc) return rc;
    return p.old_interp_flds_order ? launch_berg<PH_THERMO | PH_SPREAD>(h) : launch_berg<PH_INTERP | PH_THERMO | PH_SPREAD>(h);
  }
  if (p.footloose && !p.static_icebergs && !h->dbg.fl_unfused) {
    // calving sits between evolve and thermodynamics (IB:5453): fused into the per-berg launch (PH_FL), one pass over the
    // SoA instead of three (320 B per berg-step instead of 530).  The children it appends are new rows; they
    // owe this step's thermodynamics and spreading, which a second, short launch over those rows delivers.
    h->flags.has_fl = 1;
    KID_HIP(h, hipMemsetAsync(h->d_fl_cursor, 0, sizeof(int), h->stream));
    const long long n_old = h->n;
    rc = p.old_interp_flds_order ? launch_berg<PH_EVOLVE | PH_FL | PH_THERMO | PH_SPREAD>(h) : launch_berg<PH_INTERP | PH_EVOLVE | PH_FL | PH_THERMO | PH_SPREAD>(h);
    const unsigned fl_step_now = h->fl_step;
    h->fl_step += 1u;   // one footloose pass (hot and general build of this launch share the step word)
    if (rc) return rc;
    int appended = 0;  // the population grew: the host needs the new size before the next launch
    KID_HIP(h, hipMemcpyAsync(&appended, h->d_fl_cursor, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    KID_HIP(h, hipStreamSynchronize(h->stream));
    if (appended > h->capacity - n_old) {
      h->n = h->capacity;
      h->err = "footloose calving ran out of capacity: create the handle with room for child bergs";
      return KID_ECAPACITY;
    }
    h->n = n_old + appended;
    rc = fl_place_children(h, n_old, appended, fl_step_now);
    if (rc) return rc;
    rc = fl_assign_ids(h, n_old, appended);
    if (rc) return rc;
    return p.old_interp_flds_order ? launch_berg<PH_THERMO | PH_SPREAD>(h, n_old, appended) : launch_berg<PH_INTERP | PH_THERMO | PH_SPREAD>(h, n_old, appended);
  }
  if (p.footloose) {  // phase by phase (KID_FL_UNFUSED, static bergs): three launches
    rc = launch_berg<PH_INTERP | PH_EVOLVE>(h);
    if (rc) return rc;
    rc = kid_footloose_calving(h);
    if (rc) return rc;
    return launch_berg<PH_INTERP | PH_THERMO | PH_SPREAD>(h);
  }
  if (p.find_melt_using_spread_mass) {
    // IB:5490-5503: the gridded mass BEFORE the thermodynamics (calculate_mass_on_ocean without diagnostics, the 'mass'
    // gather into grd%spread_mass_old, planes reset), then thermodynamics + create_gridded_icebergs_fields as usual; the
    // gather replaces floating_melt by (spread_mass_old - spread_mass)/dt (IB:3436-3445)
    if (!h->d_spread_mass_old) {
      KID_HIP(h, hipMalloc(&h->d_spread_mass_old_own, 2 * h->ncell * sizeof(double)));   // + spread_mass_tmp
      KID_HIP(h, hipMemsetAsync(h->d_spread_mass_old_own, 0, 2 * h->ncell * sizeof(double), h->stream));
      h->d_spread_mass_old = h->d_spread_mass_old_own;
    }
    if (!p.static_icebergs) { rc = p.old_interp_flds_order ? launch_berg<PH_EVOLVE>(h) : launch_berg<PH_INTERP | PH_EVOLVE>(h); if (rc) return rc; }
    const size_t on_ocean = (size_t)KID_A_MASS_ON_OCEAN * h->ncell, on_bytes = 36 * h->ncell * sizeof(double);
    KID_HIP(h, hipMemsetAsync(h->d_acc + on_ocean, 0, on_bytes, h->stream));
    const Flags keep = h->flags;
    h->flags.no_diag = 1; h->flags.footprint = 0;
    rc = launch_berg<PH_SPREAD>(h);
    h->flags = keep;
    if (rc) return rc;
    { const int ncomp = (h->gd.iec - h->gd.isc + 1) * (h->gd.jec - h->gd.jsc + 1);
      hipLaunchKernelGGL(mass_gather_kernel, dim3((unsigned)((ncomp + 255) / 256)), dim3(256), 0, h->stream, dev_grid(h), (const double *)h->d_acc, h->d_spread_mass_old, h->ncell,
                         h->params.periodic_reentry != 0 && h->gd.Lx > 0.); }
    KID_HIP(h, hipMemsetAsync(h->d_acc + on_ocean, 0, on_bytes, h->stream));
    if (p.Iceberg_melt_without_decay) {
      // the bergs do not decay, so the gridded mass AFTER the melt is what thermodynamics itself spreads (IB:3219-3238);
      // it is gathered into spread_mass_tmp (IB:3411-3413) before create_gridded_icebergs_fields spreads the unchanged bergs
      rc = p.old_interp_flds_order ? launch_berg<PH_THERMO | PH_TSPREAD>(h) : launch_berg<PH_INTERP | PH_THERMO | PH_TSPREAD>(h);
      if (rc) return rc;
      { const int ncomp = (h->gd.iec - h->gd.isc + 1) * (h->gd.jec - h->gd.jsc + 1);
        hipLaunchKernelGGL(mass_gather_kernel, dim3((unsigned)((ncomp + 255) / 256)), dim3(256), 0, h->stream, dev_grid(h), (const double *)h->d_acc, h->d_spread_mass_old + h->ncell, h->ncell,
                           h->params.periodic_reentry != 0 && h->gd.Lx > 0.); }
      KID_HIP(h, hipMemsetAsync(h->d_acc + on_ocean, 0, on_bytes, h->stream));
      return launch_berg<PH_SPREAD>(h);
    }
    return p.old_interp_flds_order ? launch_berg<PH_THERMO | PH_SPREAD>(h) : launch_berg<PH_INTERP | PH_THERMO | PH_SPREAD>(h);
  }
  if (p.old_interp_flds_order) {
    if (p.static_icebergs) rc = launch_berg<PH_THERMO | PH_SPREAD>(h);
    else if (lanes_eligible(h)) rc = launch_berg_lanes<true>(h);
    else rc = launch_berg<PH_EVOLVE | PH_THERMO | PH_SPREAD>(h);
  } else {
    // IB:5423: interpolate, evolve; IB:5473: interpolate again at the new position, then thermodynamics.  Per berg that
    // is one chain, so one launch (the second interpolation sits between the phases inside berg_kernel)
    if (lanes_eligible(h)) return launch_berg_lanes<false>(h);
    if (!p.static_icebergs && !h->dbg.new_order_unfused) return launch_berg<PH_INTERP | PH_EVOLVE | PH_THERMO | PH_SPREAD>(h);
    rc = p.static_icebergs ? launch_berg<PH_INTERP>(h) : launch_berg<PH_INTERP | PH_EVOLVE>(h);
    if (rc) return rc;
    rc = launch_berg<PH_INTERP | PH_THERMO | PH_SPREAD>(h);
  }
  return rc;
}
int kid_step_gather(kid_handle *h) {
  if (!h) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  int rc = launch_gather(h);
  if (rc) return rc;
  return KID_OK;
}
int kid_run_step(kid_handle *h, int nsteps) {
  if (!h || nsteps < 0) return KID_EINVAL;
  for (int s = 0; s < nsteps; ++s) {
    int rc = kid_step_local(h);
    if (rc) return rc;
    rc = kid_step_gather(h);
    if (rc) return rc;
    if (!h->lanes_active && !h->params.mts && !h->params.interactive_icebergs_on && h->resort_interval > 0 && ++h->steps_since_sort >= h->resort_interval) {  // IB:5437, amortised; bonded bergs keep their rows
      rc = kid_move_berg_between_cells(h);
      if (rc) return rc;
    }
  }
  return KID_OK;
}

int kid_get_accumulators(kid_handle *h, double *acc, double *out, double *scalars) {
  if (!h) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  { const int rc_j = join_side(h); if (rc_j) return rc_j; }
  if (acc) KID_HIP(h, hipMemcpyAsync(acc, h->d_acc, (size_t)KID_NACC * h->ncell * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (out) KID_HIP(h, hipMemcpyAsync(out, h->d_out, (size_t)KID_NOUT * h->ncell * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (scalars) KID_HIP(h, hipMemcpyAsync(scalars, h->d_totals, KID_NSCALAR * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  KID_HIP(h, hipStreamSynchronize(h->stream));
  if (scalars) {
    int64_t alive = 0;
    int rc = kid_num_bergs(h, nullptr, &alive);
    if (rc) return rc;
    scalars[KID_S_NBERGS_ALIVE] = (double)alive;
  }
  return KID_OK;
}
int kid_accum_device_ptr(kid_handle *h, void **dev_ptr, int64_t *count) {
  if (!h || !dev_ptr || !count) return KID_EINVAL;
  *dev_ptr = h->d_acc - KID_NSCALAR;
  *count = (int64_t)((size_t)KID_NACC * h->ncell + KID_NSCALAR);
  return KID_OK;
}
int kid_accum_live_count(kid_handle *h, int64_t *count) {
  if (!h || !count) return KID_EINVAL;
  *count = (int64_t)((size_t)nacc_active(h) * h->ncell + KID_NSCALAR);
  return KID_OK;
}
int kid_bind_accum_buffer(kid_handle *h, void *dev_ptr, int64_t count) {
  if (!h) return KID_EINVAL;
  if (!dev_ptr) { h->d_acc = h->d_acc_own + KID_NSCALAR; h->acc_prezeroed = false; return KID_OK; }
  if (count < (int64_t)((size_t)KID_NACC * h->ncell + KID_NSCALAR)) { h->err = "accumulator buffer too small"; return KID_EINVAL; }
  h->d_acc = (double *)dev_ptr + KID_NSCALAR;
  h->acc_prezeroed = false;
  return KID_OK;
}

int kid_bind_spread_mass_old(kid_handle *h, void *dev_ptr, int64_t count) {
  if (!h) return KID_EINVAL;
  if (!dev_ptr) { h->d_spread_mass_old = h->d_spread_mass_old_own; return KID_OK; }
  if (count < (int64_t)(2 * h->ncell)) { h->err = "spread_mass_old buffer too small (2 planes)"; return KID_EINVAL; }
  h->d_spread_mass_old = (double *)dev_ptr;
  return KID_OK;
}
int kid_last_redo_count(kid_handle *h, int64_t *count) {
  if (!h || !count) return KID_EINVAL;
  KID_HIP(h, hipSetDevice(h->device));
  { const int rc_j = join_side(h); if (rc_j) return rc_j; }
  int c[4] = {0, 0, 0, 0};
  KID_HIP(h, hipMemcpyAsync(c, h->d_redo_count2, 4 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  KID_HIP(h, hipStreamSynchronize(h->stream));
  *count = (int64_t)c[0 + h->redo_parity] + (h->pipelined ? (int64_t)c[2 + h->redo_parity] : 0);
  return KID_OK;
}
int kid_profile_enable(kid_handle *h, int on) {
  if (!h) return KID_EINVAL;
  h->profile = on != 0;
  h->berg_ms = 0.; h->all_ms = 0.; h->berg_launches = 0;
  return KID_OK;
}
int kid_profile_get(kid_handle *h, double *berg_ms, int64_t *launches, double *all_ms) {
  if (!h) return KID_EINVAL;
  // the events were only recorded while stepping (no host/device sync in the timed loop); resolve them now
  KID_HIP(h, hipSetDevice(h->device));
  KID_HIP(h, hipStreamSynchronize(h->stream));
  for (auto &pe : h->pending) {
    float ms = 0.f;
    KID_HIP(h, hipEventElapsedTime(&ms, pe.first, pe.second));
    h->berg_ms += ms;
    (void)hipEventDestroy(pe.first); (void)hipEventDestroy(pe.second);
  }
  h->pending.clear();
  if (berg_ms) *berg_ms = h->berg_ms;
  if (launches) *launches = h->berg_launches;
  if (all_ms) *all_ms = h->all_ms;
  return KID_OK;
}

}  // extern "C"

#include "kid_restart.inc"
#include "kid_traj.inc"
#include "kid_migrate.inc"
#include "kid_chksum.inc"

// kid_device.hpp -- gfx950 device functions of the KID evolve loop (one wavefront lane per berg, fp64, no MFMA).
//
// Written for CDNA4 directly: the ocean/atmosphere/ice fields are repacked into 64-byte per-cell records so
// that one bilinear interpolation of all eight B-grid fields is four 64-byte record loads (two 128-byte
// lines, shared by the neighbouring lanes of a cell-sorted wave) instead of 32 scattered 8-byte loads, and the
// per-call grid arithmetic of the reference (sea-surface-slope stencils, ocean_depth+ssh) is hoisted into a
// per-cell prepass.  Arithmetic that reaches a berg is kept in the reference's operation order; each function
// cites the reference lines it implements (IB = src/icebergs.F90, FW = src/icebergs_framework.F90).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/kid_types.h"

namespace kid {

// Measurement-only macros (KID_EXP_*) change what the library computes or how it is laid out; they are honoured only
// together with -DKID_EXPERIMENTS, which kid_version() reports, so that a stray -D cannot ship wrong answers silently.
#if !defined(KID_EXPERIMENTS) && (defined(KID_EXP_MARKERS) || defined(KID_EXP_NO_ATOMICS) || defined(KID_EXP_MAXRUN) || defined(KID_EXP_CHUNK) || defined(KID_EXP_NUM_VGPR) || \
                                  defined(KID_EXP_MTS_NOPAIR) || defined(KID_EXP_TIMING))
#error "KID_EXP_* macros are measurement-only: build with -DKID_EXPERIMENTS to use them"
#endif
// Keeps the machine scheduler from interleaving two long phases (each wants ~100 VGPRs for its own loads in
// flight); without it the RK4 stage body needs ~300 registers and spills, with it the kernel fits 2-3 waves/SIMD.
#ifdef KID_EXP_MARKERS
#define KID_MARK(name) asm volatile("; KIDMARK " name)
#else
#define KID_MARK(name) ((void)0)
#endif
#define KID_PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
// -DKID_EXPERIMENTS -DKID_EXP_TIMING: where a wave of the per-berg kernel spends its lifetime.  KID_TICK(n) charges the cycles
// since the previous tick (s_memtime) to segment n in a per-wave LDS table; the wave adds its table to kid_tprof[] when it
// ends (kid_exp_timing() reads it out, tools/profiling/time_segments.py prints the shares).  Measurement only.
#ifdef KID_EXP_TIMING
enum { KID_TPROF_WAVES = 262144 };
__device__ unsigned long long kid_tprof[KID_TPROF_WAVES * 16];   // one row per wave of a launch (no atomics: nothing shared between waves)
__device__ __forceinline__ void kid_tick(int idx, int idx2 = 0) {   // idx < 0: start the clock; idx == 99: add the wave's table to its row
  __shared__ unsigned long long tl[4][16];
  const unsigned long long now = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    if (idx < 0) { for (int q = 0; q < 16; ++q) tl[w][q] = 0ull; }
    else if (idx == 99) {
      const unsigned long long row = (unsigned long long)blockIdx.x * 4ull + (unsigned long long)w;
#ifdef KID_EXP_TIMING_GENERAL   // the general build's launches instead (a grid of at most 2048 waves)
      if (gridDim.x <= 4096u && row < (unsigned long long)KID_TPROF_WAVES) {
#else
      if ((gridDim.x > 4096u || idx2) && row < (unsigned long long)KID_TPROF_WAVES) {
#endif   // (the hot build's launches only: the general build's grid is small; idx2: any grid)
        unsigned long long *r = kid_tprof + row * 16ull;
        r[0] += 1ull;
        for (int q = 1; q < 16; ++q) r[q] += tl[w][q];
      }
    } else tl[w][1 + idx] += now - tl[w][0];
    tl[w][0] = now;
  }
}
#define KID_TICK(n) kid_tick(n)
#define KID_TICK_ANY(n) kid_tick(n, 1)
#else
#define KID_TICK(n) ((void)0)
#define KID_TICK_ANY(n) ((void)0)
#endif

// ---------------------------------------------------------------------------------------------------------
// Namelist switches as the device code sees them.  K = 0: read from kid_params at run time (any namelist).
// K = 1: the "plain" namelist -- every switch below at the value icebergs_nml gives it by default (FW:686-822; BASELINE
// configs 1-2: drag + Coriolis + melt, no bonds, no footloose, no diagnostics planes) -- folded at compile time, so that
// the hot build of that namelist carries none of the other branches (code size, scalar registers).  The host picks
// K = 1 only when every one of these switches has exactly this value (plain_namelist() in kid_hip.hip).
// ---------------------------------------------------------------------------------------------------------
// K = 3: the plain namelist again, for a handle that stores the bergs' environment (kid_set_store_environment on: the members
// berg%uo .. hi are written every step, IB:2890-2894) -- the same folded switches; K = 1 is the build that does not store it.
// K = 2: the footloose profile (tests/footloose_tests/input.nml, BASELINE config 3: Verlet on a regular Cartesian grid, footloose
// calving with bergy bits, the corrected rolling scheme, new spreading switched off by passive mode) folded the same way.
// X(switch, value in the plain namelist, value in the footloose profile)
#define KID_SWITCHES(X)                                                                                                      \
  X(old_bug_bilin, 1, 0) X(coastal_drift, 0., 0.) X(cdrag_grounding, 0., 0.) X(use_new_predictive_corrective, 0, 1)          \
  X(iceberg_bonds_on, 0, 0) X(internal_bergs_for_drag, 0, 0) X(hexagonal_icebergs, 0, 0) X(speed_limit, 0., 0.)              \
  X(override_iceberg_velocities, 0, 0) X(use_f_plane, 0, 1) X(use_updated_rolling_scheme, 0, 1) X(tip_parameter, 0., 0.)     \
  X(use_mixed_melting, 0, 0) X(melt_icebergs_as_ice_shelf, 0, 0) X(set_melt_rates_to_zero, 0, 0) X(use_operator_splitting, 1, 1) \
  X(footloose, 0, 1) X(bergy_bit_erosion_fraction, 0., 1.) X(diag_mask, 0, 0) X(allow_bergs_to_roll, 1, 1)                   \
  X(Iceberg_melt_without_decay, 0, 0) X(grounding_fraction, 0., 0.) X(clipping_depth, 0., 0.) X(use_old_spreading, 1, 0)      \
  X(add_weight_to_ocean, 1, 0) X(time_average_weight, 0, 0) X(find_melt_using_spread_mass, 0, 0) X(mts, 0, 0) X(dem, 0, 0)
template <int K> struct Sw {
#define KID_X(name, plain, flp) static __device__ __forceinline__ auto name(const kid_params &p) -> decltype(p.name) { if constexpr (K == 1 || K == 3) return (decltype(p.name))(plain); else if constexpr (K == 2) return (decltype(p.name))(flp); else return p.name; }
  KID_SWITCHES(KID_X)
#undef KID_X
};

// Division and square root.  The IEEE expansions cost ~12 (division) and ~16 (square root) instructions each, a fifth of
// the hot build.  Unless -DKID_EXACT_MATH, a quotient is a product with a Newton-refined v_rcp_f64 (error <= ~1.5 ulp
// instead of 0.5), so that divisors which repeat (M, the cell area, dt, W+L, ...) are inverted once; results then differ
// from the oracle's at the 1e-16 level (tolerance 1e-10), like the other trims.
#ifdef KID_EXACT_MATH
struct Rcp { double d; };
__device__ __forceinline__ Rcp kid_rcp(double b) { return Rcp{b}; }
__device__ __forceinline__ double operator*(double a, Rcp r) { return a / r.d; }
__device__ __forceinline__ double kid_div(double a, double b) { return a / b; }
#else
struct Rcp { double r; };
__device__ __forceinline__ Rcp kid_rcp(double b) {
  double r = __builtin_amdgcn_rcp(b);          // ~2^-26 relative
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);                  // <= 1 ulp
  return Rcp{r};
}
__device__ __forceinline__ double operator*(double a, Rcp r) { return a * r.r; }
__device__ __forceinline__ double kid_div(double a, double b) {  // one more correction: <= ~0.6 ulp
  const Rcp r = kid_rcp(b);
  const double q = a * r.r;
  return __builtin_fma(__builtin_fma(-b, q, a), r.r, q);
}
#endif
// kid_div(a, b) with the refined reciprocal of b already in hand (a divisor that repeats: the same bits as kid_div)
#ifdef KID_EXACT_MATH
__device__ __forceinline__ double kid_div_r(double a, double b, Rcp) { return a / b; }
#else
__device__ __forceinline__ double kid_div_r(double a, double b, Rcp r) {
  const double q = a * r.r;
  return __builtin_fma(__builtin_fma(-b, q, a), r.r, q);
}
#endif
// a*b + c.  The library is compiled with -ffp-contract=off and stays so: left to the compiler, contraction differs between
// the hot and the general build of the same source line, and a berg's result would depend on which of them stepped it.
// Where the hot loop spends its multiplies and adds (bilinear interpolation, rotations, sums of squares) the fused form is
// spelled out instead -- the same instruction in every build -- unless -DKID_EXACT_MATH.
#ifdef KID_EXACT_MATH
__device__ __forceinline__ double kid_fma(double a, double b, double c) { return a * b + c; }
#else
__device__ __forceinline__ double kid_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
#endif
// Square root of a speed or a length: one Goldschmidt step on v_rsq_f64 (~23 bits) and one residual correction, <= 1 ulp;
// without the range scaling of the IEEE expansion (the arguments are sums of squares of velocities, areas, ...: far from the
// ends of the exponent range) -- 11 instructions instead of 17.  0 and negative arguments as sqrt() treats them.
#ifdef KID_EXACT_MATH
__device__ __forceinline__ double kid_sqrt(double x) { return sqrt(x); }
#else
__device__ __forceinline__ double kid_sqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  return (x > 0.) ? g : ((x == 0.) ? x : __builtin_nan(""));
}
#endif

// The same for a sum of squares of speeds (never negative, never infinite): the seed is taken of max(x, 1e-300), so x = 0 comes
// out as 0 through the arithmetic itself (g = 0 * 1e150 = 0 in every step) and the three selects of the general form go; the
// result is the same bits as kid_sqrt(x) for every x that is 0 or >= 1e-300.
#ifdef KID_EXACT_MATH
__device__ __forceinline__ double kid_sqrt_nn(double x) { return sqrt(x); }
#else
__device__ __forceinline__ double kid_sqrt_nn(double x) {
  const double y = __builtin_amdgcn_rsq(__builtin_fmax(x, 1.e-300));
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
  return __builtin_fma(__builtin_fma(-g, g, x), h, g);
}
#endif

// 1/sqrt(x), x > 0 finite: v_rsq_f64 (~23 bits) and two Newton steps y <- y + y (1 - x y^2) / 2, <= 1 ulp
#ifndef KID_EXACT_MATH
__device__ __forceinline__ double kid_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double e = __builtin_fma(-x * y, y, 1.0);
  y = __builtin_fma(y, 0.5 * e, y);
  e = __builtin_fma(-x * y, y, 1.0);
  return __builtin_fma(y, 0.5 * e, y);
}
#endif

// reference module constants, IB:68-80
constexpr double RHO_ICE = 916.7, RHO_AIR = 1.1, RHO_SEAWATER = 1025.0, GRAVITY = 9.8;
constexpr double CD_AV = 1.3, CD_AH = 0.0055, CD_WV = 0.9, CD_WH = 0.0012, CD_IV = 0.9;

// B-grid corner record (i,j): everything bilin() reads in interp_flds (IB:4757-4765)
struct alignas(64) VelRec { double cosr, sinr, uo, vo, ui, vi, ua, va; };
// A-grid cell record (i,j): PCM tracers (IB:4815-4818), od (IB:4897) and the hoisted slope stencils (IB:4903-4926)
struct alignas(64) TrcRec { double sst, sss, cn, hi, od, ddx, ddy, msk; };
// corner geometry + cell area/mask (FW:6325-6332, IB:7945-7975, IB:3114)
struct alignas(32) GeoRec { double lon, lat, area, msk; };

struct DevGrid {
  int isd, ied, jsd, jed, isc, iec, jsc, jec, ni, nj;
  int latlon, regular;
  double Lx;
  const VelRec *vel;
  const TrcRec *trc;
  const GeoRec *geo;
  const double *dx, *dy, *ocean_depth, *ssh;
  // 1.0 where the hot build may step a berg of this cell: all four corner cells inside the data domain, no corner at the
  // pole, and the corners a strictly convex quadrilateral -- then "(xi, yj) inside the unit square" and the reference's
  // point-in-cell test (FW:6076-6160) are the same statement and the hot build tests the former (pack_static_kernel)
  const double *hotok;
  // the cell packets themselves (PK_* layout, PK_GSTRIDE doubles per cell), gathered from the three record arrays once per
  // forcing change (pack_packets_kernel): the hot build stages a cell with two contiguous loads per lane instead of
  // resolving, per lane, which record of which neighbour cell element q lives in and loading from 64 different records
  const double *pkt;
  // (sin, cos) of the latitude of every cell's north-east corner (pack_static_kernel): the first RK4 stage gets sin / cos of the
  // berg's own latitude from them by the angle-addition formulas (lat_terms_cell); null on a Cartesian grid
  const double *latref;
  // Parameter-only subexpressions of the hot loop, evaluated once on the host with the same IEEE operations (so the
  // results are the ones every lane used to compute for itself, per RK4 stage): sin(pi/180*lat_ref) alone was 7 % of
  // the step
  double sin_lat_ref;  // f-plane / Cartesian Coriolis, IB:2043-2047
  double pi_180, r180_pi, dydl, rho_ratio;  // pi/180, 180/pi, (180/pi)/Rearth (IB:462-477), rho_bergs/rho_seawater (IB:2052)
  double fl_e1;  // exp(pi/4) of the footloose foot length (IB:2538)
  __device__ __forceinline__ int idx(int i, int j) const { return (i - isd) + (j - jsd) * ni; }
};

// the plain build (K = 1) is for lat-lon grids (grid_is_latlon is the namelist default), the footloose profile (K = 2) for Cartesian ones
template <int K> __device__ __forceinline__ bool grid_latlon(const DevGrid &g) { if constexpr (K == 1 || K == 3) return true; else if constexpr (K == 2) return false; else return g.latlon != 0; }

struct Env { double uo, vo, ui, vi, ua, va, ssh_x, ssh_y, sst, sss, cn, hi, od; };

// ---------------------------------------------------------------------------------------------------------
// Cell views: everything a berg reads from the grid around its cell (i,j).
//   GlbCell reads the record arrays in global memory (general build).
//   PkCell  reads a 68-double "cell packet" that the wave staged in LDS once per step (hot build): a berg of the
//           hot build never leaves its cell (it bails out otherwise), so the four RK4 stages, the five cell
//           searches and the thermodynamics interpolation all hit the same packet, and the cell-sorted lanes of a
//           wave share a handful of packets.  This is the LDS staging of the forcing fields for the bilinear
//           interpolation: ~270 per-lane global loads per berg-step become ~10 cooperative loads per wave.
// Packet layout (doubles): corner records k = 0:(i-1,j-1) 1:(i,j-1) 2:(i-1,j) 3:(i,j)
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(3))) int lds_int;
enum { PK_VEL = 0, PK_CORNER = 32, PK_T0 = 40, PK_DDX = 45, PK_DDY = 51, PK_AREA = 57, PK_MSK = 58, PK_HOTOK = 67, PK_SIZE = 68, PK_STRIDE = 68,
       PK_GSTRIDE = 72 };   // doubles between two cells' packets in DevGrid::pkt (576 B: whole 64-byte lines)
struct Corners { double lon00, lat00, lon10, lat10, lon11, lat11, lon01, lat01; };

struct GlbCell {
  const DevGrid &g;
  int c;
  __device__ __forceinline__ int corner_cell(int k) const { return c - ((k & 1) ? 0 : 1) - ((k & 2) ? 0 : g.ni); }
  __device__ __forceinline__ double vel(int k, int f) const { return reinterpret_cast<const double *>(g.vel + corner_cell(k))[f]; }
  __device__ __forceinline__ Corners corners() const {
    const GeoRec a = g.geo[c - g.ni - 1], b = g.geo[c - g.ni], d = g.geo[c], e = g.geo[c - 1];
    return Corners{a.lon, a.lat, b.lon, b.lat, d.lon, d.lat, e.lon, e.lat};
  }
  __device__ __forceinline__ double corner_lat11() const { return g.geo[c].lat; }
  __device__ __forceinline__ double t0(int f) const { return reinterpret_cast<const double *>(g.trc + c)[f]; }
  // ddx_ssh at (0,+1),(0,0),(0,-1),(-1,+1),(-1,0),(-1,-1); ddy_ssh at (+1,0),(0,0),(-1,0),(+1,-1),(0,-1),(-1,-1)
  __device__ __forceinline__ double ddx(int k) const { return g.trc[c - (k / 3) + (1 - (k % 3)) * g.ni].ddx; }
  __device__ __forceinline__ double ddy(int k) const { return g.trc[c + (1 - (k % 3)) - (k / 3) * g.ni].ddy; }
  __device__ __forceinline__ double area() const { return g.geo[c].area; }
  __device__ __forceinline__ double msk(int di, int dj) const { return g.geo[c + di + dj * g.ni].msk; }
  __device__ __forceinline__ bool unrot() const {
    return vel(0, 0) == 1. && vel(1, 0) == 1. && vel(2, 0) == 1. && vel(3, 0) == 1. && vel(0, 1) == 0. && vel(1, 1) == 0. && vel(2, 1) == 0. && vel(3, 1) == 0.;
  }
};
struct PkCell {
  const lds_double *pk;
  __device__ __forceinline__ double vel(int k, int f) const { return pk[PK_VEL + k * 8 + f]; }
  __device__ __forceinline__ Corners corners() const {
    return Corners{pk[PK_CORNER + 0], pk[PK_CORNER + 1], pk[PK_CORNER + 2], pk[PK_CORNER + 3],
                   pk[PK_CORNER + 6], pk[PK_CORNER + 7], pk[PK_CORNER + 4], pk[PK_CORNER + 5]};
  }
  __device__ __forceinline__ double corner_lat11() const { return pk[PK_CORNER + 7]; }
  __device__ __forceinline__ double t0(int f) const { return pk[PK_T0 + f]; }
  __device__ __forceinline__ double ddx(int k) const { return pk[PK_DDX + k]; }
  __device__ __forceinline__ double ddy(int k) const { return pk[PK_DDY + k]; }
  __device__ __forceinline__ double area() const { return pk[PK_AREA]; }
  __device__ __forceinline__ double msk(int di, int dj) const { return pk[PK_MSK + (di + 1) + 3 * (dj + 1)]; }
  // PK_HOTOK holds +-(1 + 2 [sides along the axes, pack_static_kernel] + 4 [unrotated: cos = 1 and sin = 0 at the four corners]),
  // positive where the hot build may step a berg of the cell: DevGrid::hotok and no NaN among the sea-surface-slope stencil
  // values (pack_packets_kernel), so that the hot evolve needs no NaN test of its own (IB:4869-4870); 0 on the data domain's rim
  __device__ __forceinline__ bool hotok() const { return pk[PK_HOTOK] > 0.; }
  __device__ __forceinline__ bool rect() const { const double f = pk[PK_HOTOK]; return f == 3. || f == 7.; }
  __device__ __forceinline__ bool unrot() const { return fabs(pk[PK_HOTOK]) >= 5.; }
};
template <bool FAST> struct CellOf;
template <> struct CellOf<true> {
  typedef PkCell type;
  static __device__ __forceinline__ PkCell make(const DevGrid &, const lds_double *pk, int, int) { return PkCell{pk}; }
};
// The general build reads the same gathered packet, from memory: one base address and fixed offsets per cell (576 contiguous
// bytes) instead of three record arrays indexed by neighbour -- its interpolation and cell search were chains of dependent
// loads (a third of a general-build wave's 83 us, tools/profiling/time_segments.py with KID_EXP_TIMING_GENERAL).  Every cell a
// berg can sit in has a packet (kid_upload_bergs admits the computational domain and the first halo row; the halo is two
// cells wide at least); -DKID_EXP_GENERAL_RECORDS keeps the record arrays.
typedef __attribute__((address_space(1))) double kid_gdouble;
struct GlbPkCell {
  const kid_gdouble *pk;
  __device__ __forceinline__ double vel(int k, int f) const { return pk[PK_VEL + k * 8 + f]; }
  __device__ __forceinline__ Corners corners() const {
    return Corners{pk[PK_CORNER + 0], pk[PK_CORNER + 1], pk[PK_CORNER + 2], pk[PK_CORNER + 3],
                   pk[PK_CORNER + 6], pk[PK_CORNER + 7], pk[PK_CORNER + 4], pk[PK_CORNER + 5]};
  }
  __device__ __forceinline__ double corner_lat11() const { return pk[PK_CORNER + 7]; }
  __device__ __forceinline__ double t0(int f) const { return pk[PK_T0 + f]; }
  __device__ __forceinline__ double ddx(int k) const { return pk[PK_DDX + k]; }
  __device__ __forceinline__ double ddy(int k) const { return pk[PK_DDY + k]; }
  __device__ __forceinline__ double area() const { return pk[PK_AREA]; }
  __device__ __forceinline__ double msk(int di, int dj) const { return pk[PK_MSK + (di + 1) + 3 * (dj + 1)]; }
  __device__ __forceinline__ bool unrot() const { return fabs(pk[PK_HOTOK]) >= 5.; }
};
template <> struct CellOf<false> {
#ifdef KID_EXP_GENERAL_RECORDS
  typedef GlbCell type;
  static __device__ __forceinline__ GlbCell make(const DevGrid &g, const lds_double *, int i, int j) { return GlbCell{g, g.idx(i, j)}; }
#else
  typedef GlbPkCell type;
  static __device__ __forceinline__ GlbPkCell make(const DevGrid &g, const lds_double *, int i, int j) {
    return GlbPkCell{(const kid_gdouble *)g.pkt + (size_t)g.idx(i, j) * (size_t)PK_GSTRIDE};
  }
#endif
};
// One wave stages the packet of a cell into LDS: lane q fetches packet element q (and q+64).  Where element q
// lives (which record array, which neighbour cell, which field) is fixed, so it is resolved once per lane:
// address of element q for cell c = base + c * stride.
struct PacketSrc { const char *base; long long stride; };
__device__ __forceinline__ PacketSrc packet_source(const DevGrid &g, int q) {
  const char *arr; long long stride, cell_off, field;
  if (q < PK_CORNER) {            // 4 corner velocity records x 8 fields
    const int k = q >> 3;
    arr = reinterpret_cast<const char *>(g.vel); stride = sizeof(VelRec); field = q & 7;
    cell_off = -((k & 1) ? 0 : 1) - ((k & 2) ? 0 : g.ni);
  } else if (q < PK_T0) {         // corner lon/lat in the order 00,10,01,11
    const int k = (q - PK_CORNER) >> 1;
    arr = reinterpret_cast<const char *>(g.geo); stride = sizeof(GeoRec); field = (q - PK_CORNER) & 1;
    cell_off = -((k & 1) ? 0 : 1) - ((k & 2) ? 0 : g.ni);
  } else if (q < PK_DDX) {
    arr = reinterpret_cast<const char *>(g.trc); stride = sizeof(TrcRec); field = q - PK_T0; cell_off = 0;
  } else if (q < PK_DDY) {
    const int k = q - PK_DDX;
    arr = reinterpret_cast<const char *>(g.trc); stride = sizeof(TrcRec); field = 5; cell_off = -(k / 3) + (1 - (k % 3)) * g.ni;
  } else if (q < PK_AREA) {
    const int k = q - PK_DDY;
    arr = reinterpret_cast<const char *>(g.trc); stride = sizeof(TrcRec); field = 6; cell_off = (1 - (k % 3)) - (k / 3) * g.ni;
  } else if (q == PK_AREA) {
    arr = reinterpret_cast<const char *>(g.geo); stride = sizeof(GeoRec); field = 2; cell_off = 0;
  } else if (q < PK_HOTOK) {
    const int m = q - PK_MSK;
    arr = reinterpret_cast<const char *>(g.geo); stride = sizeof(GeoRec); field = 3; cell_off = (m % 3 - 1) + (m / 3 - 1) * g.ni;
  } else {
    arr = reinterpret_cast<const char *>(g.hotok); stride = sizeof(double); field = 0; cell_off = 0;
  }
  return PacketSrc{arr + cell_off * stride + field * 8, stride};
}

// v_min_f64 / v_max_f64 (one instruction each; `a < b ? a : b` is a compare and two selects).  Same result unless an
// operand is a NaN, which the path never feeds them.
__device__ __forceinline__ double dmin(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ double dmax(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ double sign1(double b) { return copysign(1.0, b); }

// Fortran MODULO(a,p), p>0 (FW:6568).  A longitude handed to it lies inside the window [0,p) in all but the
// rarest cases, where the exact result is the argument itself; everything else goes out of line.
__device__ __noinline__ double f_modulo_slow(double a, double p) {
  if (a >= 0.0) { if (a < 2.0 * p) return a - p; }  // exact (Sterbenz)
  else if (a > -p) return a + p;
  double r = fmod(a, p);
  if (r != 0.0 && (r < 0.0)) r += p;
  return r;
}
__device__ __forceinline__ double f_modulo(double a, double p) {
  double r = a;
  if (__builtin_expect(!(a >= 0.0 && a < p), 0)) r = f_modulo_slow(a, p);
  return r;
}
__device__ __forceinline__ double mod_around(double x, double y, double Lx) {  // FW:6558-6573
  if (Lx > 0.) {
    const double lo = y - Lx / 2.;
    return f_modulo(x - lo, Lx) + lo;
  }
  return x;
}

// ---------------------------------------------------------------------------------------------------------
// point-in-cell tests, FW:6076-6296
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool sum_sign_dot_prod4(double x0, double y0, double x1, double y1, double x2, double y2,
                                                   double x3, double y3, double x, double y, double Lx) {
  const double xx = mod_around(x, x0, Lx), xx0 = mod_around(x0, x0, Lx), xx1 = mod_around(x1, x0, Lx);
  const double xx2 = mod_around(x2, x0, Lx), xx3 = mod_around(x3, x0, Lx);
  const double l0 = (xx - xx0) * (y1 - y0) - (y - y0) * (xx1 - xx0);
  const double l1 = (xx - xx1) * (y2 - y1) - (y - y1) * (xx2 - xx1);
  const double l2 = (xx - xx2) * (y3 - y2) - (y - y2) * (xx3 - xx2);
  const double l3 = (xx - xx3) * (y0 - y3) - (y - y3) * (xx0 - xx3);
  // south and east edges belong to the cell, north and west do not (FW:6199-6206)
  const double p0 = (l0 == 0.) ? -0.5 : sign1(l0), p1 = (l1 == 0.) ? 0.5 : sign1(l1);
  const double p2 = (l2 == 0.) ? 0.5 : sign1(l2), p3 = (l3 == 0.) ? -0.5 : sign1(l3);
  return ((fabs(p0) + fabs(p2)) + (fabs(p1) + fabs(p3))) == fabs((p0 + p2) + (p1 + p3));
}
__device__ __noinline__ bool sum_sign_dot_prod5(double x0, double y0, double x1, double y1, double x2, double y2,
                                                double x3, double y3, double x4, double y4, double x, double y, double Lx) {
  const double xx = mod_around(x, x0, Lx), xx0 = mod_around(x0, x0, Lx), xx1 = mod_around(x1, x0, Lx);
  const double xx2 = mod_around(x2, x0, Lx), xx3 = mod_around(x3, x0, Lx), xx4 = mod_around(x4, x0, Lx);
  const double l0 = (xx - xx0) * (y1 - y0) - (y - y0) * (xx1 - xx0);
  const double l1 = (xx - xx1) * (y2 - y1) - (y - y1) * (xx2 - xx1);
  const double l2 = (xx - xx2) * (y3 - y2) - (y - y2) * (xx3 - xx2);
  const double l3 = (xx - xx3) * (y4 - y3) - (y - y3) * (xx4 - xx3);
  const double l4 = (xx - xx4) * (y0 - y4) - (y - y4) * (xx0 - xx4);
  const double p0 = (l0 == 0.) ? 0. : sign1(l0), p1 = (l1 == 0.) ? 0. : sign1(l1), p2 = (l2 == 0.) ? 0. : sign1(l2);
  const double p3 = (l3 == 0.) ? 0. : sign1(l3), p4 = (l4 == 0.) ? 0. : sign1(l4);
  return (((fabs(p0) + fabs(p2)) + (fabs(p1) + fabs(p3))) + fabs(p4) - fabs(((p0 + p2) + (p1 + p3)) + p4)) < 0.5;
}

__device__ __forceinline__ bool cell_in_data_domain(const DevGrid &g, int i, int j) {
  return !(i - 1 < g.isd || i > g.ied || j - 1 < g.jsd || j > g.jed);
}

template <bool FAST = false>
__device__ __forceinline__ bool is_point_in_cell(const DevGrid &g, const Corners &q, double x, double y) {  // FW:6102-6158
  const double Lx = g.Lx;
  const double a = mod_around(q.lon00, x, Lx), b = mod_around(q.lon10, x, Lx);
  const double c = mod_around(q.lon01, x, Lx), d = mod_around(q.lon11, x, Lx);
  const double xlo = dmin(dmin(dmin(a, b), c), d), xhi = dmax(dmax(dmax(a, b), c), d);
  const double tol = 0.1;
  const double ylo = dmin(dmin(dmin(q.lat00, q.lat10), q.lat01), q.lat11);
  const double yhi = dmax(dmax(dmax(q.lat00, q.lat10), q.lat01), q.lat11);
  // the reference returns early on the crude bounds (FW:6118, 6122); evaluated without branches here
  const bool crude = !(x < (xlo - tol) || x > (xhi + tol)) && !(y < ylo || y > yhi);
  if (!FAST && crude && g.latlon && yhi > 89.999) {  // one corner at the pole: five-sided polygon (cold path; FAST bails earlier)
    if (q.lat11 > 89.999)
      return sum_sign_dot_prod5(q.lon00, q.lat00, q.lon10, q.lat10, q.lon10, q.lat11, q.lon01, q.lat11, q.lon01, q.lat01, x, y, Lx);
    else if (q.lat01 > 89.999)
      return sum_sign_dot_prod5(q.lon00, q.lat00, q.lon10, q.lat10, q.lon11, q.lat11, q.lon11, q.lat01, q.lon00, q.lat01, x, y, Lx);
    else if (q.lat00 > 89.999)
      return sum_sign_dot_prod5(q.lon01, q.lat00, q.lon10, q.lat00, q.lon10, q.lat10, q.lon11, q.lat11, q.lon01, q.lat01, x, y, Lx);
    else if (q.lat10 > 89.999)
      return sum_sign_dot_prod5(q.lon00, q.lat00, q.lon00, q.lat10, q.lon11, q.lat10, q.lon11, q.lat11, q.lon01, q.lat01, x, y, Lx);
  }
  return crude & sum_sign_dot_prod4(q.lon00, q.lat00, q.lon10, q.lat10, q.lon11, q.lat11, q.lon01, q.lat01, x, y, Lx);
}

// FW:6439-6534.  Returns false on the reference's FATAL paths.
__device__ __forceinline__ bool calc_xiyj(double x1, double x2, double x3, double x4, double y1, double y2, double y3, double y4,
                                          double x, double y, double &xi, double &yj, double Lx) {
  const double alpha = x2 - x1, delta = y2 - y1, beta = x4 - x1, epsilon = y4 - y1;
  const double gamma = (x3 - x1) - (alpha + beta), kappa = (y3 - y1) - (delta + epsilon);
  double a = (kappa * beta - gamma * epsilon);
  const double dx = mod_around(x, x1, Lx) - x1, dy = y - y1;
  double b = (delta * beta - alpha * epsilon) - (kappa * dx - gamma * dy);
  double c = (alpha * dy - delta * dx);
  bool ok = true;
  if (fabs(a) > 1.e-12) {
    const double d = 0.25 * (b * b) - a * c;
    if (d >= 0.) {
      const double sd = kid_sqrt(d);
      const Rcp ra = kid_rcp(a);
      const double yy1 = -(0.5 * b + sd) * ra, yy2 = -(0.5 * b - sd) * ra;
      yj = (fabs(yy1 - 0.5) < fabs(yy2 - 0.5)) ? yy1 : yy2;
    } else { ok = false; yj = -999.; }
  } else {
    yj = (b != 0.) ? kid_div(-c, b) : 0.;
  }
  a = (alpha + gamma * yj);
  b = (delta + kappa * yj);
  if (a != 0.) xi = kid_div(dx - beta * yj, a);
  else if (b != 0.) xi = kid_div(dy - epsilon * yj, b);
  else {
    c = (epsilon * alpha - beta * delta) + (epsilon * gamma - beta * kappa) * yj;
    if (c != 0.) xi = (epsilon * dx - beta * dy) / c; else { ok = false; xi = -999.; }
  }
  return ok;
}

// FW:6359-6404: a cell with a corner at the pole is searched on the co-latitude tangent plane
__device__ __noinline__ void pos_within_polar_cell(const DevGrid &g, const kid_params &p, const Corners &q, double x, double y,
                                                   double &xi, double &yj, int &err) {
  const double pi_180 = p.pi / 180.;
  const double xx = (90. - y) * cos(x * pi_180), yy = (90. - y) * sin(x * pi_180);
  const double x1 = (90. - q.lat00) * cos(q.lon00 * pi_180), y1 = (90. - q.lat00) * sin(q.lon00 * pi_180);
  const double x2 = (90. - q.lat10) * cos(q.lon10 * pi_180), y2 = (90. - q.lat10) * sin(q.lon10 * pi_180);
  const double x3 = (90. - q.lat11) * cos(q.lon11 * pi_180), y3 = (90. - q.lat11) * sin(q.lon11 * pi_180);
  const double x4 = (90. - q.lat01) * cos(q.lon01 * pi_180), y4 = (90. - q.lat01) * sin(q.lon01 * pi_180);
  if (!calc_xiyj(x1, x2, x3, x4, y1, y2, y3, y4, xx, yy, xi, yj, g.Lx)) err = 1;
  if (is_point_in_cell(g, q, x, y)) {
    if (!((xi >= 0. && xi < 1.) && (yj >= 0. && yj < 1.))) {
      double fac = 2.1 * dmax(fabs(xi - 0.5), fabs(yj - 0.5)); fac = dmax(1., fac);
      xi = 0.5 + (xi - 0.5) / fac;
      yj = 0.5 + (yj - 0.5) / fac;
    }
  } else if (fabs(xi - 0.5) < 0.5 && fabs(yj - 0.5) < 0.5) err = 1;
}

// FW:6299-6436 (debug=.false.).  err is set on the reference's FATAL paths.
// FAST: the specialised hot-path build.  Anything rare (polar cells, a berg leaving its cell, the polar tangent
// plane) sets `bail` instead of being handled; the kernel then leaves that berg untouched and queues it for the
// general (FAST=false) build of the same code, which runs on the short list of such bergs.
// A berg of the hot build sits in a cell with hotok = 1 (checked once per step), so "in the cell" is "(xi, yj) in the unit
// square"; within HOT_EDGE of an edge, where rounding could make the two tests differ, the berg goes to the general build.
constexpr double HOT_EDGE = 1.e-8;
// A berg of the hot build searches the same cell five times a step; for a cell with sides along the axes the two divisors of
// that search (alpha = the cell's width, b = -alpha * its height) are inverted once (the same refined reciprocals kid_div
// would form each time: the same bits).
struct RectInv { double alpha, b; Rcp ra, rb; };
__device__ __forceinline__ RectInv rect_inv(const Corners &q) {
  RectInv r;
  r.alpha = q.lon10 - q.lon00;
  r.b = -(r.alpha * (q.lat01 - q.lat00));
  r.ra = kid_rcp(r.alpha); r.rb = kid_rcp(r.b);
  return r;
}
template <bool FAST, int K = 0, bool HAVE_RI = false, class CELL>
__device__ __forceinline__ bool pos_within_cell(const DevGrid &g, const kid_params &p, const CELL &cell, double x, double y, int i, int j,
                                                double &xi, double &yj, int &err, bool &bail, const RectInv ri = RectInv{}) {
  if constexpr (FAST) {
    const Corners q = cell.corners();
    if (!grid_latlon<K>(g) && g.regular) {
      const double ddx = fabs(q.lon11 - q.lon01), ddy = fabs(q.lat11 - q.lat10);
      const double x1 = q.lon11 - (ddx / 2), y1 = q.lat11 - (ddy / 2);
      xi = kid_div(mod_around(x, x1, g.Lx) - x1, ddx) + 0.5;
      yj = kid_div(y - y1, ddy) + 0.5;
    } else if (cell.rect()) {
      // calc_xiyj on a cell whose sides lie along the axes: beta, delta, gamma, kappa and the quadratic coefficient are exact
      // zeros, and what is left of its linear branch is this -- the same operations on the same values, bit for bit
      const double dx = mod_around(x, q.lon00, g.Lx) - q.lon00, dy = y - q.lat00;
      if constexpr (HAVE_RI) {   // (alpha and b are not 0: a berg of a degenerate cell has bailed, rk4_step)
        yj = kid_div_r(-(ri.alpha * dy), ri.b, ri.rb);
        xi = kid_div_r(dx, ri.alpha, ri.ra);
      } else {
        const double alpha = q.lon10 - q.lon00, epsilon = q.lat01 - q.lat00;
        const double b = -(alpha * epsilon), c = alpha * dy;
        yj = (b != 0.) ? kid_div(-c, b) : 0.;
        if (alpha != 0.) xi = kid_div(dx, alpha); else { err = 1; xi = -999.; }
      }
    } else if (!calc_xiyj(q.lon00, q.lon10, q.lon11, q.lon01, q.lat00, q.lat10, q.lat11, q.lat01, x, y, xi, yj, g.Lx)) err = 1;
    const bool inside = (dmin(xi, yj) > HOT_EDGE) && (dmax(xi, yj) < 1. - HOT_EDGE);
    if (!inside) bail = true;
    return inside;
  }
  xi = -999.; yj = -999.;
  if (!cell_in_data_domain(g, i, j)) return false;
  const Corners q = cell.corners();
  if (!g.latlon && g.regular) {
    const double ddx = fabs(q.lon11 - q.lon01), ddy = fabs(q.lat11 - q.lat10);
    const double x1 = q.lon11 - (ddx / 2), y1 = q.lat11 - (ddy / 2);
    const double Delta_x = mod_around(x, x1, g.Lx) - x1;
    xi = kid_div(Delta_x, ddx) + 0.5;
    yj = kid_div(y - y1, ddy) + 0.5;
  } else if (!g.latlon || dmax(dmax(dmax(q.lat00, q.lat10), q.lat11), q.lat01) < 89.999) {
    if (!calc_xiyj(q.lon00, q.lon10, q.lon11, q.lon01, q.lat00, q.lat10, q.lat11, q.lat01, x, y, xi, yj, g.Lx)) err = 1;
  } else {  // polar cell: co-latitude tangent plane (FW:6359-6404), cold and out of line
    if (FAST) { bail = true; return false; }
    else pos_within_polar_cell(g, p, q, x, y, xi, yj, err);
  }
  return is_point_in_cell<FAST>(g, q, x, y);
}

// FW:7071-7088 on the corner coordinates (used when a berg bounces, IB:7990-7991, 8050-8051)
__device__ __forceinline__ void bilin_lonlat(const DevGrid &g, const kid_params &p, int i, int j, double xi, double yj,
                                             double &lon, double &lat) {
  const Corners q = GlbCell{g, g.idx(i, j)}.corners();
  if (p.old_bug_bilin) {
    lon = (q.lon11 * (1. - xi) + q.lon01 * xi) * (1. - yj) + (q.lon10 * (1. - xi) + q.lon00 * xi) * yj;
    lat = (q.lat11 * (1. - xi) + q.lat01 * xi) * (1. - yj) + (q.lat10 * (1. - xi) + q.lat00 * xi) * yj;
  } else {
    lon = (q.lon11 * xi + q.lon01 * (1. - xi)) * yj + (q.lon10 * xi + q.lon00 * (1. - xi)) * (1. - yj);
    lat = (q.lat11 * xi + q.lat01 * (1. - xi)) * yj + (q.lat10 * xi + q.lat00 * (1. - xi)) * (1. - yj);
  }
}

// ---------------------------------------------------------------------------------------------------------
// IB:4718-4900 interp_flds (non-MTS; tidal_drift = 0)
// ---------------------------------------------------------------------------------------------------------
// need_ice = false (wave-uniform): no berg of the wave sits in a cell with sea ice (hi = 0: c_ice = 0, IB:2129), so the ice
// velocity multiplies 0 wherever it goes and is not interpolated -- bitwise the same accelerations
// NANFREE: the caller has checked that the cell's stencil values hold no NaN (hot evolve: PkCell::hotok)
template <int K = 0, bool NANFREE = false, class CELL>
__device__ __forceinline__ void interp_flds(const kid_params &p, const CELL &cell, double xi, double yj, Env &e, bool need_ice = true) {
  double wx1, wx0, wy1, wy0;  // weights of columns i / i-1 and rows j / j-1 (FW:7081-7087)
  if (Sw<K>::old_bug_bilin(p)) { wx1 = 1. - xi; wx0 = xi; wy1 = 1. - yj; wy0 = yj; }
  else { wx1 = xi; wx0 = 1. - xi; wy1 = yj; wy0 = 1. - yj; }
#define KID_BIL(f) kid_fma(kid_fma(cell.vel(3, f), wx1, cell.vel(2, f) * wx0), wy1, kid_fma(cell.vel(1, f), wx1, cell.vel(0, f) * wx0) * wy0)
  // A cell whose four corners carry cos = 1, sin = 0 (every cell of an unrotated grid; elsewhere all but the displaced-pole
  // patches) is not rotated at all: the reference interpolates cos to 1 or 1 - 2^-53 (the weights' own rounding) and multiplies
  // by it; here such a cell's velocities pass through, in both builds alike (a per-cell property: the result does not depend on
  // which other bergs share the wave).  -DKID_EXACT_MATH keeps the interpolated rotation.
#ifdef KID_EXACT_MATH
  const bool unrot = false;
#else
  const bool unrot = cell.unrot();
#endif
  const bool all_unrot = __ballot(!unrot) == 0ull;   // wave-uniform: nothing to rotate in this wave
  double cos_rot = 1., sin_rot = 0.;
  if (!all_unrot) { cos_rot = KID_BIL(0); sin_rot = KID_BIL(1); }
  double uo = KID_BIL(2), vo = KID_BIL(3), ui = 0., vi = 0., ua = KID_BIL(6), va = KID_BIL(7);
  if (need_ice) { ui = KID_BIL(4); vi = KID_BIL(5); }
#undef KID_BIL
  if (Sw<K>::coastal_drift(p) > 0.) {  // IB:4769-4776
    const double cd = Sw<K>::coastal_drift(p);
    const double mE = cell.msk(1, 0), mW = cell.msk(-1, 0), mN = cell.msk(0, 1), mS = cell.msk(0, -1), m0 = cell.msk(0, 0);
    uo = uo + cd * (mE - mW) * m0;
    ui = ui + cd * (mE - mW) * m0;
    vo = vo + cd * (mN - mS) * m0;
    vi = vi + cd * (mN - mS) * m0;
  }
  // sea-surface slope from the hoisted per-cell stencils (IB:4830-4860).  The reference's two branches (yj >= 0.5 or not) are
  // the same expression on neighbouring stencil rows with weights shifted by one: the row and the two constants are picked per
  // lane and the expression is evaluated once (the same operations on the same values as the branch the lane would have taken)
  double ssh_x, ssh_y;
  {
    const bool up = yj >= 0.5;
    const int k = up ? 0 : 1;
    const double wa = yj + (up ? -0.5 : 0.5), wb = (up ? 1.5 : 0.5) - yj;
    const double hxp = kid_fma(wa, cell.ddx(k), wb * cell.ddx(k + 1));
    const double hxm = kid_fma(wa, cell.ddx(k + 3), wb * cell.ddx(k + 4));
    ssh_x = kid_fma(xi, hxp, (1. - xi) * hxm);
  }
  {
    const bool up = xi >= 0.5;
    const int k = up ? 0 : 1;
    const double wa = xi + (up ? -0.5 : 0.5), wb = (up ? 1.5 : 0.5) - xi;
    const double hyp = kid_fma(wa, cell.ddy(k), wb * cell.ddy(k + 1));
    const double hym = kid_fma(wa, cell.ddy(k + 3), wb * cell.ddy(k + 4));
    ssh_y = kid_fma(yj, hyp, (1. - yj) * hym);
  }
  // rotate to lat-lon (IB:4953-4967)
  if (!all_unrot) {
    double t, r0, r1;
    t = uo; r0 = kid_fma(cos_rot, t, sin_rot * vo); r1 = kid_fma(cos_rot, vo, -(sin_rot * t)); if (!unrot) { uo = r0; vo = r1; }
    if (need_ice) { t = ui; r0 = kid_fma(cos_rot, t, sin_rot * vi); r1 = kid_fma(cos_rot, vi, -(sin_rot * t)); if (!unrot) { ui = r0; vi = r1; } }
    t = ua; r0 = kid_fma(cos_rot, t, sin_rot * va); r1 = kid_fma(cos_rot, va, -(sin_rot * t)); if (!unrot) { ua = r0; va = r1; }
    t = ssh_x; r0 = kid_fma(cos_rot, t, sin_rot * ssh_y); r1 = kid_fma(cos_rot, ssh_y, -(sin_rot * t)); if (!unrot) { ssh_x = r0; ssh_y = r1; }
  }
  if constexpr (!NANFREE) {   // IB:4869-4870
    if (ssh_x != ssh_x) ssh_x = 0.;
    if (ssh_y != ssh_y) ssh_y = 0.;
  }
  e.uo = uo; e.vo = vo; e.ui = ui; e.vi = vi; e.ua = ua; e.va = va; e.ssh_x = ssh_x; e.ssh_y = ssh_y;
  e.sst = cell.t0(0); e.sss = cell.t0(1); e.cn = cell.t0(2); e.hi = cell.t0(3); e.od = cell.t0(4);
}

// ---------------------------------------------------------------------------------------------------------
// IB:1950-2442 accel, non-interactive bergs
// ---------------------------------------------------------------------------------------------------------
struct BergGeom { double M, T, W, L; int n_bonds; };

// What accel computes from the berg's size and from the tracer-point values of its cell (hi, od) alone: the drag
// coefficients, the grounding drag, the size factors of the wave-radiation force (IB:2050-2130).  A berg of the hot
// build stays in its cell, so these are the same in all four RK4 stages and are evaluated once per step (the general
// build, whose berg may change cells between stages, evaluates them per stage) -- the same expressions either way.
#ifdef KID_EXACT_MATH
struct AccelPre { double c_gnd, c_ocn, c_atm, c_ice, pref_wave, F, WL2, L; Rcp rWpL; };
#else
struct AccelPre { double c_gnd, c_ocn, c_atm, c_ice, wave_q, F, L; };   // wave_q = 0.5 rho_sw/M * g * 2WL/(W+L): one register instead of three
#endif
template <int K = 0>
__device__ __forceinline__ AccelPre accel_pre(const DevGrid &g, const kid_params &p, const BergGeom &bg, double hi_cell, double od) {
  AccelPre a;
  const double M = bg.M, T = bg.T, W = bg.W, L = bg.L;
  const Rcp rM = kid_rcp(M);
  const double D = g.rho_ratio * T, F = T - D;
  const double hi = dmin(hi_cell, D), D_hi = dmax(0., D - hi);
  a.c_gnd = 0.;
  if (Sw<K>::cdrag_grounding(p) != 0.) {  // grounding drag IB:2068-2082 (cdrag_grounding = 0: c_gnd = 0 whatever groundfrac is)
    double groundfrac;
    if (p.h_to_init_grounding > 0.0) {
      groundfrac = 1.0 - kid_div(od - D, p.h_to_init_grounding);
      groundfrac = dmax(groundfrac, 0.0); groundfrac = dmin(groundfrac, 1.0);
    } else groundfrac = (D > od) ? 1.0 : 0.0;
    if (groundfrac > 0.0) a.c_gnd = (Sw<K>::cdrag_grounding(p) * W * L * groundfrac) * rM;
  }
  double dragfrac = 1.0;
  if (Sw<K>::iceberg_bonds_on(p) && Sw<K>::internal_bergs_for_drag(p)) {
    const double N_max = Sw<K>::hexagonal_icebergs(p) ? 6.0 : 4.0;
    dragfrac = ((N_max - (double)bg.n_bonds) / N_max);
  }
  a.c_ocn = RHO_SEAWATER * rM * p.ocean_drag_scale * (0.5 * CD_WV * dragfrac * W * (D_hi) + CD_WH * W * L);
  a.c_atm = RHO_AIR * rM * (0.5 * CD_AV * dragfrac * W * F + CD_AH * W * L);
  a.c_ice = (fabs(hi) == 0.) ? 0. : RHO_ICE * rM * (0.5 * CD_IV * dragfrac * W * hi);
#ifdef KID_EXACT_MATH
  a.pref_wave = (0.5 * RHO_SEAWATER) * rM;
  a.F = F; a.WL2 = 2. * W * L; a.L = L; a.rWpL = kid_rcp(W + L);
#else
  a.wave_q = (0.5 * RHO_SEAWATER) * rM * GRAVITY * (2. * W * L) * kid_rcp(W + L);
  a.F = F; a.L = L;
#endif
  return a;
}

template <bool RK, int K = 0>
__device__ __forceinline__ void accel(const DevGrid &g, const kid_params &p, const AccelPre &ap, const Env &e,
                                      int i, int j, double sin_lat, double uvel, double vvel, double uvel0, double vvel0,
                                      double dt, double &ax, double &ay, double &axn, double &ayn, double &bxn, double &byn,
                                      unsigned &tickets) {
  // RK: alpha=0, C_N=0, predictive-corrective per namelist; Verlet: alpha=C_N=1, predictive-corrective forced (IB:2002-2013)
  const bool new_pc = RK ? (Sw<K>::use_new_predictive_corrective(p) != 0) : true;
  const double u_star = kid_fma(axn, dt / 2., uvel0), v_star = kid_fma(ayn, dt / 2., vvel0);
  const double uo = e.uo, vo = e.vo, ui = e.ui, vi = e.vi, ua = e.ua, va = e.va;
  const double f_cori = (2. * p.omega) * sin_lat;  // IB:2043-2047, the caller picks lat or lat_ref
  const double c_gnd = ap.c_gnd, c_ocn = ap.c_ocn, c_atm = ap.c_atm;
  // wave radiation IB:2085-2102
  double uwave = ua - uo, vwave = va - vo;
  double wmod = kid_fma(uwave, uwave, vwave * vwave);
  const double ampl = 0.5 * 0.02025 * wmod, Lwavelength = 0.32 * wmod;
  const double Lcutoff = 0.125 * Lwavelength, Ltop = 0.25 * Lwavelength;
#ifdef KID_EXACT_MATH
  const double Cr = 0.06 * dmin(dmax(0., kid_div(ap.L - Lcutoff, (Ltop - Lcutoff) + 1.e-30)), 1.);
  double wave_rad = ap.pref_wave * Cr * GRAVITY * ampl * dmin(ampl, ap.F) * ap.WL2 * ap.rWpL;
  wmod = kid_sqrt(kid_fma(ua, ua, va * va));
  if (wmod != 0.) { const Rcp rw = kid_rcp(wmod); uwave = ua * rw; vwave = va * rw; } else { uwave = 0.; vwave = 0.; wave_rad = 0.; }
#else
  // min(max(0, q), 1) of q = (L - Lcutoff) / (Ltop - Lcutoff + 1e-30) is 1 exactly when the numerator is not below the
  // denominator -- any berg longer than a quarter of the wave length, i.e. nearly all of them: the division is done only in waves
  // that hold a shorter one (each lane's value is the same either way)
  const double cr_num = ap.L - Lcutoff, cr_den = (Ltop - Lcutoff) + 1.e-30;
  double cr_q = 1.;
  if (__ballot(!(cr_num >= cr_den)) != 0ull) cr_q = dmin(dmax(0., kid_div(cr_num, cr_den)), 1.);
  const double Cr = 0.06 * cr_q;
  const double wave_rad = ap.wave_q * Cr * ampl * dmin(ampl, ap.F);
  // the unit vector of the wind from one Newton-refined reciprocal square root.  No wind: ua = va = 0, and 0 times the (finite)
  // root of the floor is the reference's uwave = vwave = 0; its wave_rad = 0 only ever multiplies them
  wmod = kid_fma(ua, ua, va * va);
  { const double rw = kid_rsqrt(__builtin_fmax(wmod, 1.e-300)); uwave = ua * rw; vwave = va * rw; }
#endif
  double c_ice = ap.c_ice;
  // no lane of the wave has sea ice (c_ice = 0 everywhere, e.g. config 2): drag_ice = 0 * (...) is 0 whatever the
  // speeds are, so its square roots and its terms of the sums below are skipped -- the same numbers
  const bool any_ice = __ballot(c_ice != 0.) != 0ull;
  if (any_ice) { if (fabs(ui) + fabs(vi) == 0.) c_ice = 0.; }
  const bool has_gnd = Sw<K>::cdrag_grounding(p) != 0.;   // (c_gnd = 0 otherwise, accel_pre)
  (void)has_gnd;
  const double ex = -GRAVITY * e.ssh_x + wave_rad * uwave, ey = -GRAVITY * e.ssh_y + wave_rad * vwave;  // IB:2142-2149
  double axn_l, ayn_l, bxn_l, byn_l;
  if (RK) { axn_l = 0.; ayn_l = 0.; bxn_l = ex + f_cori * vvel; byn_l = ey - f_cori * uvel; }          // IB:2172-2173
  else    { axn_l = ex + f_cori * v_star; ayn_l = ey - f_cori * u_star; bxn_l = 0.; byn_l = 0.; }      // IB:2165-2166
  double uveln = new_pc ? uvel0 : uvel, vveln = new_pc ? vvel0 : vvel;
  // the |V0 - V_x| halves of the predictive-corrective drag do not change between the two passes
  double s0o = 0., s0a = 0., s0i = 0.;
  if (new_pc) {
    s0o = kid_sqrt_nn(kid_fma((uvel0 - uo), (uvel0 - uo), (vvel0 - vo) * (vvel0 - vo)));
    s0a = kid_sqrt_nn(kid_fma((uvel0 - ua), (uvel0 - ua), (vvel0 - va) * (vvel0 - va)));
    if (any_ice) s0i = kid_sqrt_nn(kid_fma((uvel0 - ui), (uvel0 - ui), (vvel0 - vi) * (vvel0 - vi)));
  }
  const double A12_0 = RK ? -0. * dt * f_cori : (-1. * dt * f_cori) / 2.;  // IB:2244-2251 (alpha, C_N)
  const double A21_0 = RK ? 0. * dt * f_cori : (1. * dt * f_cori) / 2.;
  ax = 0.; ay = 0.;
#pragma unroll
  for (int itloop = 1; itloop <= 2; ++itloop) {  // IB:2183-2277
    double drag_ocn, drag_atm, drag_ice;
    if (new_pc) {
      drag_ocn = c_ocn * 0.5 * (kid_sqrt_nn(kid_fma((uveln - uo), (uveln - uo), (vveln - vo) * (vveln - vo))) + s0o);
      drag_atm = c_atm * 0.5 * (kid_sqrt_nn(kid_fma((uveln - ua), (uveln - ua), (vveln - va) * (vveln - va))) + s0a);
      drag_ice = any_ice ? c_ice * 0.5 * (kid_sqrt_nn(kid_fma((uveln - ui), (uveln - ui), (vveln - vi) * (vveln - vi))) + s0i) : 0.;
    } else {
#ifdef KID_EXACT_MATH
      const double us = 0.5 * (uveln + uvel), vs = 0.5 * (vveln + vvel);
#else
      // first pass: uveln is uvel itself, and 0.5 (u + u) = u exactly
      const double us = (itloop == 1) ? uvel : 0.5 * (uveln + uvel), vs = (itloop == 1) ? vvel : 0.5 * (vveln + vvel);
#endif
      drag_ocn = c_ocn * kid_sqrt_nn(kid_fma((us - uo), (us - uo), (vs - vo) * (vs - vo)));
      drag_atm = c_atm * kid_sqrt_nn(kid_fma((us - ua), (us - ua), (vs - va) * (vs - va)));
      drag_ice = any_ice ? c_ice * kid_sqrt_nn(kid_fma((us - ui), (us - ui), (vs - vi) * (vs - vi))) : 0.;
    }
    const double drag_gnd = c_gnd;
#ifdef KID_EXACT_MATH
    double RHS_x = (axn_l / 2) + bxn_l, RHS_y = (ayn_l / 2) + byn_l;
    RHS_x = RHS_x - drag_ocn * (u_star - uo) - drag_atm * (u_star - ua) - drag_ice * (u_star - ui) - drag_gnd * u_star;  // beta=1
    RHS_y = RHS_y - drag_ocn * (v_star - vo) - drag_atm * (v_star - va) - drag_ice * (v_star - vi) - drag_gnd * v_star;
    const double lambda = drag_ocn + drag_atm + drag_ice + drag_gnd;
#else
    // the same sums in the same order, without the terms that are 0 times something (RK: axn = 0; no ice in the wave; no
    // grounding drag in the namelist): x - 0 * y = x
    double RHS_x = RK ? bxn_l : (axn_l / 2) + bxn_l, RHS_y = RK ? byn_l : (ayn_l / 2) + byn_l;
    RHS_x = RHS_x - drag_ocn * (u_star - uo) - drag_atm * (u_star - ua);  // beta=1
    RHS_y = RHS_y - drag_ocn * (v_star - vo) - drag_atm * (v_star - va);
    double lambda = drag_ocn + drag_atm;
    // (the empty asm keeps each block a branch: speculated, the compiler evaluates the terms anyway and selects)
    if (any_ice) { asm volatile(""); RHS_x = RHS_x - drag_ice * (u_star - ui); RHS_y = RHS_y - drag_ice * (v_star - vi); lambda = lambda + drag_ice; }
    if (has_gnd) { asm volatile(""); RHS_x = RHS_x - drag_gnd * u_star; RHS_y = RHS_y - drag_gnd * v_star; lambda = lambda + drag_gnd; }
#endif
    const double A11 = kid_fma(dt, lambda, 1.), A22 = A11;
#ifdef KID_EXACT_MATH
    constexpr bool diagonal = false;
#else
    constexpr bool diagonal = RK;   // alpha = C_N = 0 (IB:2002-2013): A12 = A21 = 0 and A11 = A22, the 2x2 solve is a division by A11
#endif
    if constexpr (diagonal) {
      const Rcp rA = kid_rcp(A11);
      ax = RHS_x * rA; ay = RHS_y * rA;
    } else {
      const double detA = 1. * kid_rcp((A11 * A22) - (A12_0 * A21_0));
      ax = detA * kid_fma(A22, RHS_x, -(A12_0 * RHS_y));
      ay = detA * kid_fma(A11, RHS_y, -(A21_0 * RHS_x));
    }
    uveln = kid_fma(dt, ax, u_star);
    vveln = kid_fma(dt, ay, v_star);
  }
  if (RK) { axn = 0.; ayn = 0.; }                                               // IB:2286
  else    { axn = ex + f_cori * vveln; ayn = ey - f_cori * uveln; }              // IB:2288-2297
  bxn = ax - (axn / 2); byn = ay - (ayn / 2);
  if (Sw<K>::speed_limit(p) > 0. || Sw<K>::speed_limit(p) == -1.) {  // IB:2304-2323: only the ticket counter survives
    const double speed = kid_sqrt(uveln * uveln + vveln * vveln);
    if (speed > 0.) {
      const int c = g.idx(i, j);
      const double loc_dx = dmin(0.5 * (g.dx[c] + g.dx[c - g.ni]), 0.5 * (g.dy[c] + g.dy[c - 1]));
      const double new_speed = loc_dx / dt * Sw<K>::speed_limit(p);
      if (new_speed < speed && Sw<K>::speed_limit(p) > 0.) tickets += 1u;
    }
  }
  if (Sw<K>::override_iceberg_velocities(p)) { ax = 0.; ay = 0.; axn = 0.; ayn = 0.; bxn = 0.; byn = 0.; }
}

// ---------------------------------------------------------------------------------------------------------
// IB:7819-8063 adjust_index_and_ground (debug=.false.)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void adjust_index_slow(const DevGrid &g, const kid_params &p, double &lon, double &lat,
                                              int &i, int &j, double &xi, double &yj, int &err) {
  // cold path: hop cells / bounce off land.  (The second pos_within_cell(i0,j0) of IB:7940 repeats the caller's.)
  constexpr double posn_eps = 0.05;
  const int i0 = i, j0 = j;
  bool bounced = false, lret = false, unused_bail = false;
  for (int icount = 0; !lret && icount < 4; ++icount) {
    if (xi < 0.) {
      if (i > g.isd) { if (g.geo[g.idx(i - 1, j)].msk > 0.) { if (i > g.isd + 1) i = i - 1; } else bounced = true; }
    } else if (xi >= 1.) {
      if (i < g.ied) { if (g.geo[g.idx(i + 1, j)].msk > 0.) { if (i < g.ied) i = i + 1; } else bounced = true; }
    }
    if (yj < 0.) {
      if (j > g.jsd) { if (g.geo[g.idx(i, j - 1)].msk > 0.) { if (j > g.jsd + 1) j = j - 1; } else bounced = true; }
    } else if (yj >= 1.) {
      if (j < g.jed) { if (g.geo[g.idx(i, j + 1)].msk > 0.) { if (j < g.jed) j = j + 1; } else bounced = true; }
    }
    if (bounced) {
      if (xi >= 1.) xi = 1. - posn_eps;
      if (xi < 0.) xi = posn_eps;
      if (yj >= 1.) yj = 1. - posn_eps;
      if (yj < 0.) yj = posn_eps;
      bilin_lonlat(g, p, i, j, xi, yj, lon, lat);
    }
    lret = pos_within_cell<false>(g, p, GlbCell{g, g.idx(i, j)}, lon, lat, i, j, xi, yj, err, unused_bail);
  }
  if (!bounced && lret && g.geo[g.idx(i, j)].msk > 0.) return;
  if (!bounced && !lret) {
    if (abs(i - i0) + abs(j - j0) == 0 && p.use_roundoff_fix) {  // IB:8024-8034
      xi = (xi - 0.5) * (1. - posn_eps) + 0.5;
      yj = (yj - 0.5) * (1. - posn_eps) + 0.5;
    }
  }
  if (xi >= 1.) xi = 1. - posn_eps;   // asymmetric clamps IB:8045-8049
  if (xi < 0.) xi = posn_eps;
  if (yj > 1.) yj = 1. - posn_eps;
  if (yj <= 0.) yj = posn_eps;
  bilin_lonlat(g, p, i, j, xi, yj, lon, lat);
  (void)pos_within_cell<false>(g, p, GlbCell{g, g.idx(i, j)}, lon, lat, i, j, xi, yj, err, unused_bail);
}
template <bool FAST, int K = 0, bool HAVE_RI = false>
__device__ __forceinline__ void adjust_index_and_ground(const DevGrid &g, const kid_params &p, const lds_double *pk, double &lon, double &lat,
                                                        int &i, int &j, double &xi, double &yj, int &err, bool &bail, const RectInv ri = RectInv{}) {
  if (pos_within_cell<FAST, K, HAVE_RI>(g, p, CellOf<FAST>::make(g, pk, i, j), lon, lat, i, j, xi, yj, err, bail, ri)) return;  // the common case: still in its cell
  if (FAST) bail = true;
  else adjust_index_slow(g, p, lon, lat, i, j, xi, yj, err);
}

// ---------------------------------------------------------------------------------------------------------
// polar tangent plane, IB:7767-7816, 8066-8099 (cold: lat > 89)
// ---------------------------------------------------------------------------------------------------------
__device__ __noinline__ void rotpos_to_tang(const kid_params &p, double lon, double lat, double &x, double &y) {
  const double pi_180 = p.pi / 180.;
  const double r = p.Rearth * ((90. - lat) * pi_180);
  x = r * cos(lon * pi_180); y = r * sin(lon * pi_180);
}
__device__ __noinline__ void rotpos_from_tang(const kid_params &p, double x, double y, double &lon, double &lat) {
  const double r180_pi = 180. / p.pi;
  const double r = sqrt(x * x + y * y);
  lat = 90. - (r180_pi * r / p.Rearth);
  lon = r180_pi * acos(x / r) * sign1(y);
}
__device__ __noinline__ void rotvec_to_tang(const kid_params &p, double lon, double uvel, double vvel, double &xdot, double &ydot) {
  const double pi_180 = p.pi / 180.;
  const double clon = cos(lon * pi_180), slon = sin(lon * pi_180);
  xdot = -slon * uvel - clon * vvel; ydot = clon * uvel - slon * vvel;
}
__device__ __noinline__ void rotvec_from_tang(const kid_params &p, double lon, double xdot, double ydot, double &uvel, double &vvel) {
  const double pi_180 = p.pi / 180.;
  const double clon = cos(lon * pi_180), slon = sin(lon * pi_180);
  uvel = -slon * xdot + clon * ydot; vvel = -clon * xdot - slon * ydot;
}

// sin(lat) for Coriolis (IB:2043-2047) and the metric dlon/dx (IB:462-477) share one argument reduction
struct LatTerms { double sin_f, dxdl, s, c; };
template <int K = 0>
__device__ __forceinline__ LatTerms lat_terms(const DevGrid &g, const kid_params &p, double lat, double sin_ref) {
  LatTerms t;
  if (grid_latlon<K>(g)) {
    double s, c;
    sincos(lat * g.pi_180, &s, &c);
    t.dxdl = g.r180_pi * kid_rcp(p.Rearth * c);
    t.sin_f = Sw<K>::use_f_plane(p) ? sin_ref : s;
    t.s = s; t.c = c;
  } else { t.dxdl = 1.; t.sin_f = sin_ref; t.s = 0.; t.c = 1.; }
  return t;
}

// The first RK4 stage: the berg sits within one cell of its cell's north-east corner, whose sin / cos are tabulated
// (DevGrid::latref): sin / cos of lat = lat_c + d from the angle-addition formulas with a 7th / 8th-order Taylor series in d
// (|d| < 0.02 rad: truncation < 1e-18) -- ~15 instructions instead of a sincos with argument reduction
// (~120, 6 % of the step's vector instructions).  Both builds use it (the same numbers for a berg whichever build steps
// it); not bitwise equal to sincos(lat) (differences ~2e-16); -DKID_EXACT_MATH keeps sincos.
template <int K = 0>
__device__ __forceinline__ LatTerms lat_terms_cell(const DevGrid &g, const kid_params &p, double lat, double sin_ref,
                                                   double lat_c, double s_c, double c_c, bool have_ref) {
#ifdef KID_EXACT_MATH
  (void)lat_c; (void)s_c; (void)c_c; (void)have_ref;
  return lat_terms<K>(g, p, lat, sin_ref);
#else
  LatTerms t;
  if (!grid_latlon<K>(g)) { t.dxdl = 1.; t.sin_f = sin_ref; t.s = 0.; t.c = 1.; return t; }
  const double d = (lat - lat_c) * g.pi_180;
  double s, c;
  if (have_ref && fabs(d) < 0.02) {   // (1.15 degrees: the next terms of the two series are < 1e-18 there; taller cells take sincos)
    const double d2 = d * d;   // Horner with fused multiply-adds: sin d = d + d^3 (-1/6 + d^2 (1/120 - d^2/5040)), cos d = 1 + d^2 (-1/2 + d^2 (1/24 + d^2 (-1/720 + d^2/40320)))
    const double sd = kid_fma(d * d2, kid_fma(d2, kid_fma(d2, -1. / 5040., 1. / 120.), -1. / 6.), d);
    const double cd = kid_fma(d2, kid_fma(d2, kid_fma(d2, kid_fma(d2, 1. / 40320., -1. / 720.), 1. / 24.), -0.5), 1.);
    s = kid_fma(s_c, cd, c_c * sd); c = kid_fma(c_c, cd, -(s_c * sd));
  } else sincos(lat * g.pi_180, &s, &c);
  t.dxdl = g.r180_pi * kid_rcp(p.Rearth * c);
  t.sin_f = Sw<K>::use_f_plane(p) ? sin_ref : s;
  t.s = s; t.c = c;
  return t;
#endif
}

// RK4 stages 2-4 sit within a few hundred metres of stage 1: sin/cos of lat1 + d come from the angle-addition formulas
// with a 5th/4th-order Taylor series in d (|d| < 2e-3 rad: truncation < 1e-17), ~20 instructions instead of a sincos
// with argument reduction (~150).  Not bitwise equal to sincos(lat) (differences ~1e-16); -DKID_EXACT_MATH keeps sincos.
template <int K = 0>
__device__ __forceinline__ LatTerms lat_terms_near(const DevGrid &g, const kid_params &p, double lat, double sin_ref,
                                                   double lat1, double s1, double c1) {
#ifdef KID_EXACT_MATH
  (void)lat1; (void)s1; (void)c1;
  return lat_terms<K>(g, p, lat, sin_ref);
#else
  LatTerms t;
  if (!grid_latlon<K>(g)) { t.dxdl = 1.; t.sin_f = sin_ref; t.s = 0.; t.c = 1.; return t; }
  const double d = (lat - lat1) * g.pi_180;
  double s, c;
  if (fabs(d) < 2.e-3) {
    const double d2 = d * d;   // sin d = d + d^3 (-1/6 + d^2/120), cos d = 1 + d^2 (-1/2 + d^2/24), Horner with fused multiply-adds
    const double sd = kid_fma(d * d2, kid_fma(d2, 1. / 120., -1. / 6.), d);
    const double cd = kid_fma(d2, kid_fma(d2, 1. / 24., -0.5), 1.);
    s = kid_fma(s1, cd, c1 * sd); c = kid_fma(c1, cd, -(s1 * sd));
  } else sincos(lat * g.pi_180, &s, &c);
  t.dxdl = g.r180_pi * kid_rcp(p.Rearth * c);
  t.sin_f = Sw<K>::use_f_plane(p) ? sin_ref : s;
  t.s = s; t.c = c;
  return t;
#endif
}

// footloose beam constants (IB:2538-2547, 3013-3015, 3376-3378): 1/(g rho_sw) and 1/(12 (1 - 0.3**2)) are literals of the
// source, folded at compile time with the same IEEE operations; x**3 and x**0.25 of the buoyancy length are x*x*x and
// sqrt(sqrt(x)) (within an ulp of pow; -DKID_EXACT_MATH keeps pow)
constexpr double FL_LW_C = 1. / (GRAVITY * RHO_SEAWATER), FL_B_C1 = 1. / (12. * (1. - 0.3 * 0.3));
#ifdef KID_EXACT_MATH
__device__ __forceinline__ double kid_cube(double x) { return pow(x, 3.); }
__device__ __forceinline__ double kid_root4(double x) { return pow(x, 0.25); }
#else
__device__ __forceinline__ double kid_cube(double x) { return x * x * x; }
__device__ __forceinline__ double kid_root4(double x) { return sqrt(sqrt(x)); }
#endif

// per-berg dynamic state carried through one step
struct BergDyn {
  double lon, lat, uvel, vvel, axn, ayn, bxn, byn, xi, yj, uvel_prev, vvel_prev;
  int ine, jne;
};

// ---------------------------------------------------------------------------------------------------------
// IB:7331-7679 Runge_Kutta_stepping.  env is only read when .not.old_interp_flds_order.
//
// The general build runs the four stages as ONE loop body; the hot build unrolls them (33 KB of code, inside the 64 KB
// instruction cache a CU pair shares): with `s` a constant the stage selects and the loop-carried copies go, 203 -> 160 registers.
// The reference's sums (u1+u4)+2(u2+u3) are kept bit-for-bit by carrying the two partial sums A=(x1[+x4]) and
// B=(x2[+x3]) per quantity.  On the polar tangent plane (lat>89, IB:7393) the same loop advances the
// tangent-plane position/velocity instead; its rot* helpers are out of line (cold).
// ---------------------------------------------------------------------------------------------------------
template <bool OLD_ORDER, bool FAST, int K = 0>
__device__ __forceinline__ void rk4_step(const DevGrid &g, const kid_params &p, const BergGeom &bg, const Env &stored,
                                         BergDyn &d, unsigned &tickets, int &err, bool &bail, const lds_double *pk,
                                         double latref_s = 0., double latref_c = 1.) {
  const double dt = p.dt, dt_2 = 0.5 * dt, dt_6 = dt / 6.;
  const double sin_ref = g.sin_lat_ref;
  const double dydl = grid_latlon<K>(g) ? g.dydl : 1.;
  const int i1 = d.ine, j1 = d.jne;
  const double xi1 = d.xi, yj1 = d.yj, lon1 = d.lon, lat1 = d.lat, uvel1 = d.uvel, vvel1 = d.vvel;
  const bool on_tang = FAST ? false : ((lat1 > 89.) && g.latlon);
  if (FAST && (lat1 > 89.) && grid_latlon<K>(g)) { bail = true; return; }
  Env e = stored;
  // the size- and cell-dependent factors of accel: once per step where the berg cannot change cells between stages (hot
  // build) or carries its environment with it (.not.old_interp_flds_order), per stage otherwise
  constexpr bool PRE_ONCE = FAST || !OLD_ORDER;
  AccelPre ap;
  RectInv ri = {};
#ifdef KID_EXACT_MATH
  constexpr bool HAVE_RI = false;
#else
  constexpr bool HAVE_RI = FAST;
#endif
  if constexpr (FAST) {
    const PkCell cell{pk};
    if (!cell.hotok()) { bail = true; return; }
    if (OLD_ORDER) ap = accel_pre<K>(g, p, bg, cell.t0(3), cell.t0(4));
    if constexpr (HAVE_RI) {
      ri = rect_inv(cell.corners());   // (used by the lanes whose cell is rect())
      if (cell.rect() && (ri.alpha == 0. || ri.b == 0.)) { bail = true; return; }   // a degenerate cell: the general build reports it
    }
  }
  if (!OLD_ORDER) ap = accel_pre<K>(g, p, bg, stored.hi, stored.od);
  // hot build: no lane of the wave in a cell with sea ice -> the ice velocity is not interpolated (interp_flds)
  const bool need_ice = (FAST && OLD_ORDER && !(Sw<K>::coastal_drift(p) > 0.)) ? (__ballot(ap.c_ice != 0.) != 0ull) : true;
  double bxn = 0., byn = 0.;
  double x1 = 0., y1 = 0., xdot1 = 0., ydot1 = 0.;
  if (on_tang) { rotpos_to_tang(p, lon1, lat1, x1, y1); rotvec_to_tang(p, lon1, uvel1, vvel1, xdot1, ydot1); }
  // partial sums: A = q1 (+ q4), B = q2 (+ q3) for q in {u, v, ax, ay, axn, ayn} (or their tangent-plane twins)
  double Au = 0., Bu = 0., Av = 0., Bv = 0., Aax = 0., Bax = 0., Aay = 0., Bay = 0., Aaxn = 0., Baxn = 0., Aayn = 0., Bayn = 0.;
  double lon_s = lon1, lat_s = lat1, uvel_s = uvel1, vvel_s = vvel1, xdot_s = xdot1, ydot_s = ydot1;
  int i = i1, j = j1; double xi = xi1, yj = yj1;
  double s_lat1 = 0., c_lat1 = 1.;
  // hot build: the four stages unrolled -- `s` is a constant in each copy (no stage selects, no loop-carried copies: 203 -> 174
  // registers); the general build keeps one copy of the body
#if defined(KID_EXACT_MATH) || defined(KID_EXP_RK_ROLLED)
#pragma unroll 1
#else
#pragma unroll (FAST ? 4 : 1)
#endif
  for (int s = 0; s < 4; ++s) {
    KID_MARK("loop_top"); KID_TICK(s == 0 ? 0 : 5);
    if (s > 0) { i = i1; j = j1; xi = xi1; yj = yj1; adjust_index_and_ground<FAST, K, HAVE_RI>(g, p, pk, lon_s, lat_s, i, j, xi, yj, err, bail, ri); }  // IB:7430-7431
    KID_PHASE_FENCE();
    KID_MARK("after_adjust"); KID_TICK(1);
    LatTerms lt;
    if (s == 0) {   // (the cell's reference: the latitude of its north-east corner from the packet, its sin / cos from DevGrid::latref)
      const typename CellOf<FAST>::type c0 = CellOf<FAST>::make(g, pk, i1, j1);
      lt = lat_terms_cell<K>(g, p, lat_s, sin_ref, c0.corner_lat11(), latref_s, latref_c, g.latref != nullptr);
      s_lat1 = lt.s; c_lat1 = lt.c;
    }
    else lt = lat_terms_near<K>(g, p, lat_s, sin_ref, lat1, s_lat1, c_lat1);
    double qu = uvel_s * lt.dxdl, qv = vvel_s * dydl;          // u_k, v_k  IB:7412
    KID_PHASE_FENCE();
    double axn_s = d.axn, ayn_s = d.ayn, ax, ay;               // IB:7400-7401
    KID_MARK("after_latterms"); KID_TICK(2);
    if (OLD_ORDER) interp_flds<K, FAST>(p, CellOf<FAST>::make(g, pk, i, j), xi, yj, e, need_ice);
    if constexpr (!PRE_ONCE) ap = accel_pre<K>(g, p, bg, e.hi, e.od);
    KID_PHASE_FENCE();
    KID_MARK("after_interp"); KID_TICK(3);
    accel<true, K>(g, p, ap, e, i, j, lt.sin_f, uvel_s, vvel_s, uvel1, vvel1, (s < 2) ? dt_2 : dt, ax, ay, axn_s, ayn_s, bxn, byn, tickets);
    KID_PHASE_FENCE();
    KID_MARK("after_accel"); KID_TICK(4);
    double qax = ax, qay = ay, qaxn = axn_s, qayn = ayn_s;
    if (on_tang) {
      qu = xdot_s; qv = ydot_s;
      rotvec_to_tang(p, lon_s, ax, ay, qax, qay);
      rotvec_to_tang(p, lon_s, axn_s, ayn_s, qaxn, qayn);
    }
#ifdef KID_EXACT_MATH
    if (s == 0)      { Au = qu; Av = qv; Aax = qax; Aay = qay; Aaxn = qaxn; Aayn = qayn; }
    else if (s == 1) { Bu = qu; Bv = qv; Bax = qax; Bay = qay; Baxn = qaxn; Bayn = qayn; }
    else if (s == 2) { Bu = Bu + qu; Bv = Bv + qv; Bax = Bax + qax; Bay = Bay + qay; Baxn = Baxn + qaxn; Bayn = Bayn + qayn; }
    else             { Au = Au + qu; Av = Av + qv; Aax = Aax + qax; Aay = Aay + qay; Aaxn = Aaxn + qaxn; Aayn = Aayn + qayn; }
#else
    {  // one running sum q1 + 2 q2 + 2 q3 + q4 per quantity (kept in A, B stays 0): half the registers of the (q1+q4), (q2+q3) pairs
      const double wq = (s == 0 || s == 3) ? 1. : 2.;
      Au = Au + wq * qu; Av = Av + wq * qv; Aax = Aax + wq * qax; Aay = Aay + wq * qay; Aaxn = Aaxn + wq * qaxn; Aayn = Aayn + wq * qayn;
      // (unrolled stages: the sums are formed here and now -- left to itself the scheduler sinks the four additions of every
      // quantity to the end of the step and keeps, or spills, the terms until then)
      if constexpr (FAST) asm volatile("" : "+v"(Au), "+v"(Av), "+v"(Aax), "+v"(Aay));
    }
#endif
    if (s < 3) {  // X_{k+1} = X1 + c V_k ; V_{k+1} = V1 + c A_k   (c = dt/2, dt/2, dt)
      const double c = (s < 2) ? dt_2 : dt;
      if (on_tang) {
        const double xs = x1 + c * qu, ys = y1 + c * qv;
        xdot_s = xdot1 + c * qax; ydot_s = ydot1 + c * qay;
        rotpos_from_tang(p, xs, ys, lon_s, lat_s);
        rotvec_from_tang(p, lon_s, xdot_s, ydot_s, uvel_s, vvel_s);
      } else {
        lon_s = lon1 + c * qu; lat_s = lat1 + c * qv;
        uvel_s = uvel1 + c * ax; vvel_s = vvel1 + c * ay;
      }
    }
  }
  KID_MARK("loop_end"); KID_TICK(5);
  // combine IB:7597-7616
  double lonn, latn, uveln, vveln, axn, ayn;
  if (on_tang) {
    const double xn = x1 + dt_6 * (Au + 2. * Bu), yn = y1 + dt_6 * (Av + 2. * Bv);
    const double xdotn = xdot1 + dt_6 * (Aax + 2. * Bax), ydotn = ydot1 + dt_6 * (Aay + 2. * Bay);
    const double xddotn = (Aaxn + 2. * Baxn) / 6., yddotn = (Aayn + 2. * Bayn) / 6.;
    rotpos_from_tang(p, xn, yn, lonn, latn);
    rotvec_from_tang(p, lonn, xdotn, ydotn, uveln, vveln);
    rotvec_from_tang(p, lonn, xddotn, yddotn, axn, ayn);  // bxn,byn stay as the 4th accel left them
  } else {
    lonn = lon1 + dt_6 * (Au + 2. * Bu);
    latn = lat1 + dt_6 * (Av + 2. * Bv);
    uveln = uvel1 + dt_6 * (Aax + 2. * Bax);
    vveln = vvel1 + dt_6 * (Aay + 2. * Bay);
    axn = (Aaxn + 2. * Baxn) / 6.;
    ayn = (Aayn + 2. * Bayn) / 6.;
    bxn = ((Aax + 2. * Bax) / 6) - (axn / 2);
    byn = ((Aay + 2. * Bay) / 6) - (ayn / 2);
  }
  i = i1; j = j1; xi = xi1; yj = yj1;
  KID_PHASE_FENCE();
  adjust_index_and_ground<FAST, K, HAVE_RI>(g, p, pk, lonn, latn, i, j, xi, yj, err, bail, ri);
  if (Sw<K>::override_iceberg_velocities(p)) { uveln = p.u_override; vveln = p.v_override; }  // IB:7151-7154
  d.lon = lonn; d.lat = latn; d.uvel = uveln; d.vvel = vveln; d.axn = axn; d.ayn = ayn; d.bxn = bxn; d.byn = byn;
  d.xi = xi; d.yj = yj; d.ine = i; d.jne = j;
}

// ---------------------------------------------------------------------------------------------------------
// IB:7203-7328 verlet_stepping + IB:7684-7764 update_verlet_position (non-interactive)
// ---------------------------------------------------------------------------------------------------------
template <bool OLD_ORDER, bool FAST, int K = 0>
__device__ __forceinline__ void verlet_step(const DevGrid &g, const kid_params &p, const BergGeom &bg, const Env &stored,
                                            BergDyn &d, unsigned &tickets, int &err, bool &bail, const lds_double *pk) {
  const double dt = p.dt, dt_2 = 0.5 * dt;
  const double sin_ref = g.sin_lat_ref;
  const double dydl = grid_latlon<K>(g) ? g.dydl : 1.;
  const double lon1 = d.lon, lat1 = d.lat, uvel1 = d.uvel, vvel1 = d.vvel;
  double axn = d.axn, ayn = d.ayn, bxn = d.bxn, byn = d.byn;
  KID_TICK(0);
  d.uvel_prev = d.uvel - dt_2 * d.bxn; d.vvel_prev = d.vvel - dt_2 * d.byn;       // IB:7256
  const double uvel3 = uvel1 + (dt_2 * axn), vvel3 = vvel1 + (dt_2 * ayn);         // IB:7259-7260
  if constexpr (FAST) { if (!PkCell{pk}.hotok() || ((lat1 > 89.) && g.latlon)) { bail = true; return; } }
  const LatTerms lt = lat_terms<K>(g, p, lat1, sin_ref);
  Env e = stored;
  if (OLD_ORDER) interp_flds<K, FAST>(p, CellOf<FAST>::make(g, pk, d.ine, d.jne), d.xi, d.yj, e);
  const AccelPre ap = accel_pre<K>(g, p, bg, e.hi, e.od);
  KID_PHASE_FENCE();
  KID_TICK(3);
  double ax1, ay1, uveln, vveln;
  accel<false, K>(g, p, ap, e, d.ine, d.jne, lt.sin_f, uvel1, vvel1, uvel1, vvel1, dt, ax1, ay1, axn, ayn, bxn, byn, tickets);
  KID_TICK(4);
  const bool on_tang = FAST ? false : ((lat1 > 89.) && g.latlon);
  if (on_tang) {
    double xdot3, ydot3, xddot1, yddot1;
    rotvec_to_tang(p, lon1, uvel3, vvel3, xdot3, ydot3);
    rotvec_to_tang(p, lon1, ax1, ay1, xddot1, yddot1);
    rotvec_from_tang(p, lon1, xdot3 + (dt * xddot1), ydot3 + (dt * yddot1), uveln, vveln);
  } else { uveln = uvel3 + (dt * ax1); vveln = vvel3 + (dt * ay1); }
  if (Sw<K>::override_iceberg_velocities(p)) { uveln = p.u_override; vveln = p.v_override; }
  // update_verlet_position reads berg%uvel AFTER the write-back (IB:7161 then 7720): u_new + dt/2 (axn+bxn)
  const double uvel2 = uveln + (dt_2 * axn) + (dt_2 * bxn), vvel2 = vveln + (dt_2 * ayn) + (dt_2 * byn);
  double lonn, latn;
  if (on_tang) {
    double x1, y1, xdot2, ydot2;
    rotpos_to_tang(p, lon1, lat1, x1, y1);
    rotvec_to_tang(p, lon1, uvel2, vvel2, xdot2, ydot2);
    rotpos_from_tang(p, x1 + (dt * xdot2), y1 + (dt * ydot2), lonn, latn);
  } else {
    const double u2 = uvel2 * lt.dxdl, v2 = vvel2 * dydl;
    lonn = lon1 + (dt * u2); latn = lat1 + (dt * v2);
  }
  KID_PHASE_FENCE();
  adjust_index_and_ground<FAST, K>(g, p, pk, lonn, latn, d.ine, d.jne, d.xi, d.yj, err, bail);
  KID_TICK(1);
  d.lon = lonn; d.lat = latn; d.uvel = uveln; d.vvel = vveln; d.axn = axn; d.ayn = ayn; d.bxn = bxn; d.byn = byn;
}

// ---------------------------------------------------------------------------------------------------------
// IB:3307-3364 rolling, IB:3370-3387 fl_bits_dimensions
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void swapd(double &x, double &y) { const double t = x; x = y; y = t; }
template <int K = 0>
__device__ __forceinline__ void rolling(const kid_params &p, double &Tn, double &Wn, double &Ln) {
  const double q = p.rho_bergs / RHO_SEAWATER;
  const double Dn = q * Tn;
  if (Dn > 0.) {
    if (!Sw<K>::use_updated_rolling_scheme(p) && Sw<K>::tip_parameter(p) < 999.) {          // scheme 3 (default)
      if (dmax(Wn, Ln) < kid_sqrt_nn(0.92 * (Dn * Dn) + 58.32 * Dn)) { swapd(Tn, Wn); if (Wn > Ln) swapd(Wn, Ln); }
    } else {
      if (Wn > Ln) swapd(Ln, Wn);
      if (!Sw<K>::use_updated_rolling_scheme(p) && Sw<K>::tip_parameter(p) >= 999.) {       // scheme 2
        if (Wn < sqrt((6.0 * q * (1 - q) * (Tn * Tn)) - (12 * 6.0 * q * Tn))) { swapd(Tn, Wn); if (Wn > Ln) swapd(Wn, Ln); }
      }
      if (Sw<K>::use_updated_rolling_scheme(p)) {                                   // scheme 1
        const double tip = (Sw<K>::tip_parameter(p) > 0.) ? Sw<K>::tip_parameter(p) : sqrt(6 * q * (1 - q));
        if ((tip * Tn) > Wn) { swapd(Tn, Wn); if (Wn > Ln) swapd(Wn, Ln); }
      }
    }
  }
}
template <int K = 0>
__device__ __forceinline__ void fl_bits_dimensions_inl(const kid_params &p, double thickness, double &L_fl, double &W_fl, double &T_fl) {
  const double l_c = p.pi / (2. * sqrt(2.));
  const double l_w = kid_root4(FL_LW_C * p.fl_youngs * FL_B_C1 * kid_cube(thickness));
  const double l_b = l_c * l_w;
  L_fl = 3. * l_b; W_fl = l_b; T_fl = thickness;
  rolling<K>(p, T_fl, W_fl, L_fl);
}
// (out of line for the cold callers; the thermodynamics of a footloose run inlines it: a call in the middle of that phase makes
// the compiler spill what is live across it)
__device__ __noinline__ void fl_bits_dimensions(const kid_params &p, double thickness, double &L_fl, double &W_fl, double &T_fl) {
  fl_bits_dimensions_inl<0>(p, thickness, L_fl, W_fl, T_fl);
}

// IB:3492-3785 find_basal_melt (cold: only with melt_icebergs_as_ice_shelf / use_mixed_melting)
__device__ __noinline__ double find_basal_melt(const DevGrid &g, const kid_params &p, double dvo, double lat, double salt,
                                               double temp, bool three_eq, double thickness) {
  const double VK = 0.40, ZETA_N = 0.052, RC = 0.20, c2_3 = 2.0 / 3.0;
  const double dR0_dT = -0.038357, dR0_dS = 0.805876, RHO_T0_S0 = 999.910681, Salin_Ice = 0.0;
  const double kd_molec_salt = 8.02e-10, kd_molec_temp = 1.41e-7, kv_molec = 1.95e-6;
  const double Cp_ml = 3974.0, LF = 3.335e5, p_atm = 101325;
  const double dTFr_dp = -7.53E-08, dTFr_dS = -0.0573, TFr_S0_P0 = 0.0832;
  const double density_ice = p.rho_bergs, Rho0 = RHO_SEAWATER, Hml = 10.;
  const double p_int = p_atm + (GRAVITY * thickness * density_ice);
  const double Rhoml = RHO_T0_S0 + dR0_dT * temp + dR0_dS * salt;
  const double I_ZETA_N = 1.0 / ZETA_N, I_LF = 1.0 / LF, I_VK = 1.0 / VK, RhoCp = Rho0 * Cp_ml;
  const double Gam_mol_t = 12.5 * pow(kv_molec / kd_molec_temp, c2_3) - 6, Gam_mol_s = 12.5 * pow(kv_molec / kd_molec_salt, c2_3) - 6;
  const double ustar_h = dmax(p.ustar_icebergs_bg, sqrt(p.cdrag_icebergs * (dvo * dvo + p.utide_icebergs * p.utide_icebergs)));
  const double f_cori = (2. * p.omega) * sin((p.pi / 180.) * ((g.latlon && !p.use_f_plane) ? lat : p.lat_ref));
  const double absf = fabs(f_cori);
  const double hBL_neut = ((absf * Hml <= VK * ustar_h) || (absf == 0.)) ? Hml : (VK * ustar_h) / absf;
  const double hBL_neut_h_molec = ZETA_N * ((hBL_neut * ustar_h) / (5.0 * kv_molec));
  const double ln_neut = (hBL_neut_h_molec > 1.0) ? log(hBL_neut_h_molec) : 0.0;
  double lprec = 0., I_Gam_T = 0., I_Gam_S = 0., wT_flux = 0.;
  bool out_of_bounds = false;
  if (three_eq) {
    double Sbdry = salt, Sb_max = 0, Sb_min = 0;
    bool Sb_max_set = false, Sb_min_set = false;
    const double dB_dS = (GRAVITY / Rhoml) * dR0_dS, dB_dT = (GRAVITY / Rhoml) * dR0_dT;
    for (int it1 = 1; it1 <= 20; ++it1) {
      const double tfreeze = (TFr_S0_P0 + dTFr_dS * Sbdry) + dTFr_dp * p_int;
      const double dT_ustar = (temp - tfreeze) * ustar_h, dS_ustar = (salt - Sbdry) * ustar_h;
      if (p.const_gamma) { I_Gam_T = p.Gamma_T_3EQ; I_Gam_S = p.Gamma_T_3EQ / 35.; }
      else {
        const double Gam_turb = I_VK * (ln_neut + (0.5 * I_ZETA_N - 1.0));
        I_Gam_T = 1.0 / (Gam_mol_t + Gam_turb); I_Gam_S = 1.0 / (Gam_mol_s + Gam_turb);
      }
      wT_flux = dT_ustar * I_Gam_T;
      const double wB_flux = dB_dS * (dS_ustar * I_Gam_S) + dB_dT * wT_flux;
      if (wB_flux > 0.0) {  // the Newton iterate is never fed back (IB:3661-3698): wB_flux is fixed
        const double n_star_term = (ZETA_N / RC) * (hBL_neut * VK) / (ustar_h * ustar_h * ustar_h);
        const double I_n_star = sqrt(1.0 + n_star_term * wB_flux);
        double Gam_turb;
        if (hBL_neut_h_molec > I_n_star * I_n_star) Gam_turb = I_VK * ((ln_neut - 2.0 * log(I_n_star)) + (0.5 * I_ZETA_N * I_n_star - 1.0));
        else Gam_turb = I_VK * (0.5 * I_ZETA_N * I_n_star - 1.0);
        if (p.const_gamma) { I_Gam_T = p.Gamma_T_3EQ; I_Gam_S = p.Gamma_T_3EQ / 35.; }
        else { I_Gam_T = 1.0 / (Gam_mol_t + Gam_turb); I_Gam_S = 1.0 / (Gam_mol_s + Gam_turb); }
        wT_flux = dT_ustar * I_Gam_T;
      }
      const double t_flux = RhoCp * wT_flux;
      lprec = I_LF * t_flux;
      const double mass_exch = (ustar_h * I_Gam_S) * Rho0;
      const double Sbdry_it = (salt * mass_exch + Salin_Ice * lprec) / (mass_exch + lprec);
      const double dS_it = Sbdry_it - Sbdry;
      if (fabs(dS_it) < 1e-4 * (0.5 * (salt + Sbdry + 1.e-10))) break;
      if (dS_it < 0.0) {
        if (Sb_max_set && (Sbdry > Sb_max)) { out_of_bounds = true; break; }
        Sb_max = Sbdry; Sb_max_set = true;
      } else {
        if (Sb_min_set && (Sbdry < Sb_min)) { out_of_bounds = true; break; }
        Sb_min = Sbdry; Sb_min_set = true;
      }
      Sbdry = Sbdry_it;  // IB:3755
    }
  }
  if (!three_eq || out_of_bounds) {
    const double tfreeze = (TFr_S0_P0 + dTFr_dS * salt) + dTFr_dp * p_int;
    const double Gam_turb = I_VK * (ln_neut + (0.5 * I_ZETA_N - 1.0));
    I_Gam_T = 1.0 / (Gam_mol_t + Gam_turb);
    wT_flux = (ustar_h * I_Gam_T) * (temp - tfreeze);
    lprec = I_LF * (RhoCp * wT_flux);
  }
  return lprec / density_ice;
}

}  // namespace kid

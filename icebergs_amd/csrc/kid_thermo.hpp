// kid_thermo.hpp -- thermodynamics, mass spreading and the wavefront-segmented scatter-add (gfx950).
//
// The reference accumulates per-cell sums by walking each cell's linked list (IB:2885-3299, IB:4989-5009).
// Here one lane owns one berg and the per-cell sums are built in two levels: lanes of a wave that sit in the
// same cell (the SoA is kept cell-sorted, so that is most of them) are first summed with a segmented
// shuffle scan, and only the last lane of each run issues the hardware fp64 atomic.  That cuts the atomic
// count by the mean run length (about 14 at 1e6 bergs on 72 000 cells) and keeps each atomic instruction's
// addresses few and clustered, which is what the memory-side atomic units want (MI355X_MICROARCH, "Global
// float atomics").
#pragma once
#include "kid_device.hpp"

namespace kid {

// ---- scatter-add of per-berg values into per-cell planes ---------------------------------------------------------
// Measured on MI355X (1e6 cell-sorted bergs, ~14 per cell): fp64 atomics execute at the memory side, one fabric
// request per lane when the lanes of a wave instruction hit the same address, and cost 0.45 ms of a 1.1 ms step when
// issued per berg and value; LDS ds_add_f64 into a workgroup window was no better (0.53 ms, same-address
// serialisation).  What is cheap is to let each wave sum its own runs of equal cells first:
//   * make_runs(): one ballot finds the runs of equal cell index among the 64 lanes (the SoA is cell-sorted, so a
//     wave holds ~5 runs; unsorted input degrades to 64 runs of one, still correct);
//   * cell_add() only parks a value in the wave's LDS staging slot [slot][lane];
//   * seg_flush(): the (slot, run) pairs are dealt to the lanes, each lane sums its run's staged values in lane
//     order (a fixed order: bitwise reproducible within a wave) and issues ONE atomic for the run.
// That is ~14x fewer atomics and, unlike a shuffle scan, costs about one LDS write + one LDS read per value.
#ifndef KID_EXP_MAXRUN
#define KID_EXP_MAXRUN 16
#endif
#ifndef KID_EXP_CHUNK
#define KID_EXP_CHUNK 12
#endif
constexpr int KID_MAXRUN = KID_EXP_MAXRUN;  // the hot build shares at most this many cell packets per wave
constexpr int KID_CHUNK = KID_EXP_CHUNK;    // staged values per flush: 6 KB of LDS per wave
// row stride of the staging block: 64 would put lane r of every row q on the same LDS bank (measured: half of the LDS
// cycles of the spreading phase were bank conflicts); 66 doubles shifts each row by 4 banks and leaves 2 doubles of
// slack behind a row for the unrolled, predicated reads of seg_flush_impl
constexpr int KID_ROW = 66;
struct Seg {
  lds_double *val;   // [KID_CHUNK][KID_ROW] staging, this wave
  lds_int *head;     // [64] first lane of run r
  lds_int *len;      // [64] length of run r
  lds_int *cell;     // [64] cell index of run r (<0: inactive lanes, nothing to add)
  lds_int *plane;    // [KID_CHUNK] plane id of staged slot
  unsigned long long heads;  // bit l: lane l starts a run
  int R;             // number of runs in this wave
  int npend;         // staged slots (wave-uniform)
  int chunk;         // staged slots per flush (KID_CHUNK; fewer in the builds that trade staging rows for a third wave per SIMD)
};
// LDS a workgroup of `waves` waves needs for its Seg tables
constexpr int seg_lds_doubles(int waves, int chunk = KID_CHUNK) { return waves * chunk * KID_ROW + 4; }
constexpr int seg_lds_ints(int waves, int chunk = KID_CHUNK) { return waves * (3 * 64 + chunk); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
__device__ __forceinline__ Seg make_runs(int key, lds_double *vals, lds_int *ints, int chunk = KID_CHUNK) {
  Seg s;
  const int lane = (int)__lane_id(), wave = (int)(threadIdx.x >> 6);
  s.chunk = chunk;
  s.val = vals + wave * (chunk * KID_ROW);
  lds_int *base = ints + wave * (3 * 64 + chunk);
  s.head = base; s.len = base + 64; s.cell = base + 128; s.plane = base + 192;
  const int prev = __shfl_up(key, 1);
  const bool is_head = (lane == 0) || (prev != key);
  const unsigned long long heads = __ballot(is_head);
  s.heads = heads;
  s.R = __popcll(heads);
  if (is_head) {
    const int r = __popcll(heads & ((1ull << lane) - 1ull));
    const unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));
    const int next = above ? lane + 1 + (int)__ffsll((long long)above) - 1 : 64;
    s.head[r] = lane; s.len[r] = next - lane; s.cell[r] = key;
  }
  s.npend = 0;
  __builtin_amdgcn_wave_barrier();
  return s;
}
// One lane per (staged slot, run) used to sum its run's values one after the other: with the SoA in cell order a wave holds
// one or two runs of 30-60 lanes, so a dozen lanes each walked a chain of dependent LDS reads and adds 60 long while the
// rest of the wave idled -- 14 % of a wave's lifetime in the hot build (tools/profiling/time_segments.py).  Now a run is cut
// into P = 1, 2 or 4 pieces summed by neighbouring lanes (P the largest that still fits the wave's 64 lanes), the pieces are
// added in order ((p0 + p1) + p2) + p3 through two shuffles and the first piece's lane issues the atomic.  Still a fixed
// order of additions for a given arrangement of the lanes.
__device__ __noinline__ void seg_flush_impl(lds_double *val, lds_int *head, lds_int *len, lds_int *cell, lds_int *plane,
                                            int R, int npend, double *acc, size_t ncell) {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const int lane = (int)__lane_id();
  const float rR = __builtin_amdgcn_rcpf((float)R);
  const int nitems = npend * R;
  const int P = (nitems * 4 <= 64) ? 4 : ((nitems * 2 <= 64) ? 2 : 1);   // wave-uniform
  const int lg = (P == 4) ? 2 : ((P == 2) ? 1 : 0);
  for (int base = 0; base < nitems; base += (64 >> lg)) {
    const int item = base + (lane >> lg), part = lane & (P - 1);
    const bool live = item < nitems;
    const int it = live ? item : 0;
    // item / R and item % R without the 25 instructions of an integer division: the float quotient is within one of the truth
    // (item < 1024, R <= 64), one correction either way
    int q = (int)((float)it * rR), r = it - q * R;
    if (r < 0) { q -= 1; r += R; } else if (r >= R) { q += 1; r -= R; }
    const int h = head[r], n = len[r], c = cell[r];
    const int per = (n + P - 1) >> lg;                 // values per piece
    const int t0 = part * per, t1 = (t0 + per < n) ? t0 + per : n;
    double sum = 0.;
    const lds_double *v = val + q * KID_ROW + h;
    int t = t0;
    for (; t + 4 <= t1; t += 4) {  // four reads in flight; the adds of a piece stay in lane order
      const double a0 = v[t], a1 = v[t + 1], a2 = v[t + 2], a3 = v[t + 3];
      sum += a0; sum += a1; sum += a2; sum += a3;
    }
    if (t < t1) {
      const double a0 = v[t], a1 = v[t + 1], a2 = v[t + 2];  // reads past the run stay inside the staging block
      sum += a0;
      if (t + 1 < t1) sum += a1;
      if (t + 2 < t1) sum += a2;
    }
    if (P >= 2) {   // pieces of a run sit in neighbouring lanes: p0 + p1 (and p2 + p3), then (p0 + p1) + ... in order
      const double s1 = __shfl_down(sum, 1);
      if (P == 2) sum = sum + s1;
      else {
        const double s2 = __shfl_down(sum, 2), s3 = __shfl_down(sum, 3);
        sum = ((sum + s1) + s2) + s3;
      }
    }
    if (live && part == 0 && c >= 0 && sum != 0.) unsafeAtomicAdd(acc + (size_t)plane[q] * ncell + (size_t)c, sum);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
__device__ __forceinline__ void seg_flush(Seg &s, double *acc, size_t ncell) {
  if (s.npend > 0) seg_flush_impl(s.val, s.head, s.len, s.cell, s.plane, s.R, s.npend, acc, ncell);
  s.npend = 0;
}
// All lanes of the wave must call this together (every call site is in wave-uniform control flow).
__device__ __forceinline__ void cell_add(double *acc, size_t ncell, int plane, int c, double v, Seg &s, bool active) {
  (void)c;
#ifdef KID_EXP_NO_ATOMICS  // measurement-only build: results are wrong, the arithmetic feeding the adds is kept alive
  if (active && v == 1.2345e-300) acc[(size_t)plane * ncell + c] = v;
  return;
#endif
  s.val[s.npend * KID_ROW + (int)__lane_id()] = active ? v : 0.0;
  s.plane[s.npend] = plane;
  s.npend += 1;
  if (s.npend == s.chunk) seg_flush(s, acc, ncell);
}

// one out-of-line copy of ocml's pow (about 3 KB of code per inlined call site)
__device__ __noinline__ double kid_pow(double x, double y) { return pow(x, y); }
// x**y for the melt laws (x >= 0, y in {0.2, 0.8}): exp(y*log(x)) is within ~2e-15 relative of the correctly rounded pow
// (|y*log x| < 8 here) at half its instruction count; -DKID_EXACT_MATH keeps ocml's pow
#ifdef KID_EXACT_MATH
__device__ __forceinline__ double kid_powr(double x, double y) { return kid_pow(x, y); }
#else
__device__ __noinline__ double kid_powr(double x, double y) { return (x == 0.) ? 0. : exp(y * log(x)); }
#endif
// The melt laws' x**0.8 and a / x**0.2 (IB:2912, 2914, 3046, 3084, 3100).  Both come from r = x**(-1/5): a single-precision seed
// (v_log_f32, v_exp_f32: ~1e-6 relative) and two Newton steps r <- r + r (1 - x r^5) / 5 -- no division, the error goes
// 1e-6 -> 3e-12 -> 3e-23 (checked against 50-digit arithmetic with the seed off by 3e-6: 3e-16 after the second step, the
// rounding of the steps themselves; round 2 ran a third) -- ~20 instructions against ~150 for exp(y log x) + a division;
// x**0.8 = x r.  Within ~1 ulp of pow.
// Outside the seed's range (and for 0, inf, NaN) the general form; -DKID_EXACT_MATH keeps pow.
#ifdef KID_EXACT_MATH
__device__ __forceinline__ double kid_mul_rpow5(double a, double x) { return a / kid_pow(x, 0.2); }
__device__ __forceinline__ double kid_pow08(double x) { return kid_pow(x, 0.8); }
#else
__device__ __noinline__ double kid_rpow5_slow(double x) { return 1. / kid_powr(x, 0.2); }
__device__ __forceinline__ bool kid_rpow5_fast_range(double x) { return x > 1e-30 && x < 1e30; }
__device__ __forceinline__ double kid_rpow5_newton(double x) {
  double r = (double)__builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf((float)x));
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double r2 = r * r, r4 = r2 * r2;
    const double e = __builtin_fma(-x, r4 * r, 1.0);
    r = __builtin_fma(r, 0.2 * e, r);
  }
  return r;
}
__device__ __forceinline__ double kid_mul_rpow5(double a, double x) {
  return a * (__builtin_expect(kid_rpow5_fast_range(x), 1) ? kid_rpow5_newton(x) : kid_rpow5_slow(x));
}
__device__ __forceinline__ double kid_pow08(double x) {
  return __builtin_expect(kid_rpow5_fast_range(x), 1) ? x * kid_rpow5_newton(x) : kid_powr(x, 0.8);
}
#endif

// ---- per-berg thermodynamic state -------------------------------------------------------------------------
struct BergThermo {
  double M, T, W, L, mass_scaling, mass_of_bits, mass_of_fl_bits, mass_of_fl_bergy_bits, fl_k, heat_density;
  double start_mass, start_day, static_berg;
  int start_year, n_bonds;
  bool alive;
};

__device__ __forceinline__ int minloc_abs10(const double *tab, double v) {  // Fortran minloc(abs(tab-v),1) - 1
  int k = 0; double best = fabs(tab[0] - v);
#pragma unroll
  for (int q = 1; q < 10; ++q) { const double d = fabs(tab[q] - v); if (d < best) { best = d; k = q; } }
  return k;
}

// IB:2844-3300 thermodynamics for one berg.  Writes the new state into `b`, scatters into acc planes.
template <int K = 0, class CELL>
__device__ __forceinline__ void spread_mass(const DevGrid &g, const kid_params &p, const CELL &cellv, const BergThermo &b, double uvel, double vvel,
                                            int i, int j, double x, double y, bool active, double *acc, size_t ncell, Seg &seg, bool footprint, double theta);
// TSPREAD (find_melt_using_spread_mass with Iceberg_melt_without_decay, IB:3219-3238): the masses the berg WOULD have after
// the step are spread onto the ocean from inside thermodynamics, before the berg is put back to what it was.  A compile-
// time switch: the spreading code is inlined only into the launches of that namelist combination.
struct TSpreadArgs { double xi, yj, theta; bool footprint; };
template <bool TSPREAD = false, int K = 0, class CELL>
__device__ __forceinline__ void thermodynamics(const DevGrid &g, const kid_params &p, const CELL &cellv, BergThermo &b, const Env &e,
                                               double uvel, double vvel, double lat, int i, int j, bool active,
                                               double *acc, size_t ncell, Seg &seg, double *scal, const TSpreadArgs *ts = nullptr) {
  constexpr double perday = 1. / 86400.;
  const double dt = p.dt;
  const Rcp rdt = kid_rcp(dt);   // the many x/dt below share one reciprocal (exact division with -DKID_EXACT_MATH)
  const int c = g.idx(i, j);
  const double SST = e.sst;
  double SSS = e.sss;
  const double IC = dmin(1., e.cn + p.sicn_shift);
  const double M = b.M, T = b.T, W = b.W, L = b.L;
  const double Vol = T * W * L;
  const Rcp rVol = kid_rcp(Vol);
  double du = uvel - e.uo, dv = vvel - e.vo;
  const double dvo = kid_sqrt_nn(du * du + dv * dv);
  du = e.ua - e.uo; dv = e.va - e.vo;
  const double dva = kid_sqrt_nn(du * du + dv * dv);
#ifdef KID_EXACT_MATH
  const double Ss = 1.5 * kid_pow(dva, 0.5) + 0.1 * dva;
#else
  const double Ss = 1.5 * kid_sqrt_nn(dva) + 0.1 * dva;   // dva**0.5 (IB:2908)
#endif
  const double dvo08 = kid_pow08(dvo);
  double Mv = dmax(7.62e-3 * SST + 1.29e-3 * (SST * SST), 0.) * perday;
  double Mb = dmax(kid_mul_rpow5(0.58 * dvo08 * (SST + 4.0), L), 0.) * perday;
  // cos(pi IC^3): no lane of the wave in sea ice (IC = 0) is the common case, and cos(0) = 1 exactly
  const double cos_ic = (__ballot(IC != 0.) == 0ull) ? 1.0 : cos(p.pi * (IC * IC * IC));
  double Me = dmax(1. / 12. * (SST + 2.) * Ss * (1 + cos_ic), 0.) * perday;
  const bool has_fl = b.mass_of_fl_bits > 0.;
  const double Mv_fl = Mv, Me_fl = Me;  // IB:2924-2926
  const double N_max = Sw<K>::hexagonal_icebergs(p) ? 6.0 : 4.0;
  double N_bonds = 0.;
  if (Sw<K>::use_mixed_melting(p) || Sw<K>::allow_bergs_to_roll(p)) {
    if (Sw<K>::iceberg_bonds_on(p)) N_bonds = (double)b.n_bonds;
    if (b.static_berg == 1) N_bonds = N_max;
  }
  if (Sw<K>::melt_icebergs_as_ice_shelf(p) || Sw<K>::use_mixed_melting(p)) {  // IB:2947-2968
    if (!p.use_mixed_layer_salinity_for_thermo) SSS = 35.0;
    double Ms = dmax(find_basal_melt(g, p, dvo, lat, SSS, SST, p.Use_three_equation_model != 0, T), 0.);
    if ((p.melt_cutoff >= 0.) && p.apply_thickness_cutoff_to_bergs_melt) {
      const double Dn = (p.rho_bergs / RHO_SEAWATER) * T;
      if ((g.ocean_depth[c] - Dn) < p.melt_cutoff) Ms = 0.;
    }
    if (Sw<K>::use_mixed_melting(p)) {
      Me = ((N_max - N_bonds) / N_max) * (Mv + Me);
      Mv = 0.0;
      Mb = (((N_max - N_bonds) / N_max) * (Mb)) + (N_bonds / N_max) * Ms;
    } else { Mv = 0.0; Me = 0.0; Mb = Ms; }
  }
  if (Sw<K>::set_melt_rates_to_zero(p)) { Mv = 0.0; Mb = 0.0; Me = 0.0; }
  double Tn, nVol, Mnew, dMb, dMv, dMe, dM, Ln1 = 0., Wn1 = 0., Ln, Wn;
  if (Sw<K>::use_operator_splitting(p)) {  // IB:2976-2994
    Tn = dmax(T - Mb * dt, 0.);
    nVol = Tn * W * L; const double Mnew1 = (nVol * rVol) * M; dMb = M - Mnew1;
    Ln1 = dmax(L - Mv * dt, 0.); Wn1 = dmax(W - Mv * dt, 0.);
    nVol = Tn * Wn1 * Ln1; const double Mnew2 = (nVol * rVol) * M; dMv = Mnew1 - Mnew2;
    Ln = dmax(Ln1 - Me * dt, 0.); Wn = dmax(Wn1 - Me * dt, 0.);
    nVol = Tn * Wn * Ln; Mnew = (nVol * rVol) * M; dMe = Mnew2 - Mnew;
    dM = M - Mnew;
  } else {
    Ln = dmax(L - (Mv + Me) * (dt), 0.); Wn = dmax(W - (Mv + Me) * (dt), 0.); Tn = dmax(T - Mb * (dt), 0.);
    nVol = Tn * Wn * Ln; Mnew = (nVol * rVol) * M; dM = M - Mnew;
    dMb = (M * rVol) * (W * L) * Mb * dt;
    dMe = (M * rVol) * (T * (W + L)) * Me * dt;
    dMv = (M * rVol) * (T * (W + L)) * Mv * dt;
  }
  double fl_k = b.fl_k;
  if (Sw<K>::footloose(p) && fl_k >= 0) {  // IB:3011-3028
    const double l_c = p.pi / (2. * sqrt(2.));
    const double l_b3 = 3. * l_c * kid_root4(FL_LW_C * p.fl_youngs * FL_B_C1 * kid_cube(Tn));
    if (L > l_b3) {
      const double fb = Tn * (1. - p.rho_bergs / RHO_SEAWATER), kd = Tn - fb;
      if (W > l_b3) fl_k = fl_k + (dMe / fb - dMv / kd) / p.rho_bergs;
      else {
        const double dMv_l = dMv * (Wn1 + W) / (2. * (Ln1 + W));
        const double dMe_l = dMe * (Wn + Wn1) / (2. * (Ln + Wn1));
        fl_k = fl_k + (dMe_l / fb - dMv_l / kd) / p.rho_bergs;
      }
      if (fl_k < 0) fl_k = 0;
    }
  }
  // footloose bits IB:3031-3068
  double Lfl = 0, Wfl = 0, Tfl = 0, Tnfl = 0, Lnfl = 0, Wnfl = 0, Mnew_fl, dMfl = 0., dMb_fl = 0., dMv_fl = 0., dMe_fl = 0.;
  if (has_fl) {
    fl_bits_dimensions_inl<K>(p, T, Lfl, Wfl, Tfl);
    const double Mfl = b.mass_of_fl_bits, Volfl = Lfl * Wfl * Tfl;
    const double Mb_fl = dmax(kid_mul_rpow5(0.58 * dvo08 * (SST + 4.0), Lfl), 0.) * perday;
    Tnfl = dmax(Tfl - Mb_fl * dt, 0.);
    if (Sw<K>::use_operator_splitting(p)) {
      double nVolfl = Tnfl * Wfl * Lfl; const double Mnew1_fl = (nVolfl / Volfl) * Mfl; dMb_fl = Mfl - Mnew1_fl;
      Lnfl = dmax(Lfl - Mv_fl * dt, 0.); Wnfl = dmax(Wfl - Mv_fl * dt, 0.);
      nVolfl = Tnfl * Wnfl * Lnfl; const double Mnew2_fl = (nVolfl / Volfl) * Mfl; dMv_fl = Mnew1_fl - Mnew2_fl;
      Lnfl = dmax(Lnfl - Me_fl * dt, 0.); Wnfl = dmax(Wnfl - Me_fl * dt, 0.);
      nVolfl = Tnfl * Wnfl * Lnfl; Mnew_fl = (nVolfl / Volfl) * Mfl; dMe_fl = Mnew2_fl - Mnew_fl;
    } else {
      Lnfl = dmax(Lfl - (Mv_fl + Me_fl) * dt, 0.); Wnfl = dmax(Wfl - (Mv_fl + Me_fl) * dt, 0.);
      const double nVolfl = Tnfl * Wnfl * Lnfl; Mnew_fl = (nVolfl / Volfl) * Mfl;
      dMb_fl = (Mfl / Volfl) * (Wfl * Lfl) * Mb_fl * dt;
      dMe_fl = (Mfl / Volfl) * (Tfl * (Wfl + Lfl)) * Me_fl * dt;
      dMv_fl = (Mfl / Volfl) * (Tfl * (Wfl + Lfl)) * Mv_fl * dt;
    }
    dMfl = Mfl - Mnew_fl;
  } else Mnew_fl = b.mass_of_fl_bits;
  // bergy bits IB:3071-3111
  double dMbitsE = 0., dMbitsM = 0., nMbits = b.mass_of_bits, dMbitsE_fl = 0., dMbitsM_fl = 0., nMbits_fl = b.mass_of_fl_bergy_bits;
  if (Sw<K>::bergy_bit_erosion_fraction(p) > 0.) {
    const double bbef = Sw<K>::bergy_bit_erosion_fraction(p);
    const double Mbits = b.mass_of_bits;
    dMbitsE = bbef * dMe;
    nMbits = Mbits + dMbitsE;
    const double Lbits = dmin(dmin(dmin(L, W), T), 40.);
    const double Abits = (Mbits / p.rho_bergs) / Lbits;
    double Mbb = dmax(kid_mul_rpow5(0.58 * dvo08 * (SST + 2.0), Lbits), 0.) * perday;
    Mbb = p.rho_bergs * Abits * Mbb;
    dMbitsM = dmin(Mbb * dt, nMbits);
    nMbits = nMbits - dMbitsM;
    if (Mnew == 0.) { dMbitsM = dMbitsM + nMbits; nMbits = 0.; }
    if (has_fl) {
      const double Mbits_fl = b.mass_of_fl_bergy_bits;
      dMbitsE_fl = bbef * dMe_fl;
      nMbits_fl = Mbits_fl + dMbitsE_fl;
      const double Lbits_fl = dmin(dmin(dmin(Lfl, Wfl), Tfl), 40.);
      const double Abits_fl = (Mbits_fl / p.rho_bergs) / Lbits_fl;
      double Mbb_fl = dmax(kid_mul_rpow5(0.58 * dvo08 * (SST + 2.0), Lbits_fl), 0.) * perday;
      Mbb_fl = p.rho_bergs * Abits_fl * Mbb_fl;
      dMbitsM_fl = dmin(Mbb_fl * dt, nMbits_fl);
      nMbits_fl = nMbits_fl - dMbitsM_fl;
      if (Mnew_fl == 0.) { dMbitsM_fl = dMbitsM_fl + nMbits_fl; nMbits_fl = 0.; }
    } else { dMbitsE_fl = 0.; dMbitsM_fl = 0.; nMbits_fl = 0.; }
  }
  // per-cell accumulation IB:3114-3208
  const double area = cellv.area(), ms = b.mass_scaling;
  const Rcp rarea = kid_rcp(area);
  const bool ok = active && (area != 0.);
  const int dm = Sw<K>::diag_mask(p);
#define KID_ACC(F, v) cell_add(acc, ncell, (F), c, (v), seg, ok)
  double melt = (dM - (dMbitsE - dMbitsM) + dMfl - (dMbitsE_fl - dMbitsM_fl)) * rdt;
  KID_ACC(KID_A_FLOATING_MELT, melt * rarea * ms);
  if (dm & KID_DIAG_MELT_BY_CLASS) {
    const int kc = (lat < 0.) ? minloc_abs10(p.initial_mass_s, b.start_mass) : minloc_abs10(p.initial_mass_n, b.start_mass);
    for (int q = 0; q < 10; ++q) KID_ACC(KID_A_MELT_BY_CLASS + q, (kc == q) ? melt * rarea * ms : 0.);
  }
  melt = melt * b.heat_density;
  {
    const double hv = ok ? melt * ms * dt : 0.;
    if (__ballot(hv != 0.) != 0ull) {   // heat_density = 0 everywhere (every BASELINE config): nothing to sum, nothing to stage
      KID_ACC(KID_A_CALVING_HFLX, melt * rarea * ms);
      const double h = wave_sum(hv);
      if (__lane_id() == 0 && h != 0.) unsafeAtomicAdd(scal + KID_S_NET_HEAT_TO_OCEAN, h);
    }
  }
  melt = dM * rdt; KID_ACC(KID_A_BERG_MELT, melt * rarea * ms);
  if (Sw<K>::bergy_bit_erosion_fraction(p) > 0.) {
    melt = (dMbitsE + dMbitsE_fl) * rdt; KID_ACC(KID_A_BERGY_SRC, melt * rarea * ms);
    melt = (dMbitsM + dMbitsM_fl) * rdt; KID_ACC(KID_A_BERGY_MELT, melt * rarea * ms);
  }
  if (__ballot(has_fl && ok) != 0ull) { melt = dMfl * rdt; KID_ACC(KID_A_FL_BITS_MELT, melt * rarea * ms); }
  if (dm & (KID_DIAG_FL_PARENT_MELT | KID_DIAG_FL_CHILD_MELT | KID_DIAG_MELT_BUOY | KID_DIAG_MELT_EROS | KID_DIAG_MELT_CONV |
            KID_DIAG_MELT_BUOY_FL | KID_DIAG_MELT_EROS_FL | KID_DIAG_MELT_CONV_FL)) {
    const bool parent = fl_k >= 0;  // IB:3144; lanes contribute 0 to the planes of the other branch
    if (dm & KID_DIAG_FL_PARENT_MELT) KID_ACC(KID_A_FL_PARENT_MELT, parent ? ((dM - (dMbitsE - dMbitsM)) * rdt) * rarea * ms : 0.);
    if (dm & KID_DIAG_FL_CHILD_MELT)
      KID_ACC(KID_A_FL_CHILD_MELT, (parent ? ((dMfl - (dMbitsE_fl - dMbitsM_fl)) * rdt) : ((dM - (dMbitsE - dMbitsM)) * rdt)) * rarea * ms);
    if (dm & KID_DIAG_MELT_BUOY) KID_ACC(KID_A_MELT_BUOY, parent ? (dMb * rdt) * rarea * ms : 0.);
    if (dm & KID_DIAG_MELT_EROS) KID_ACC(KID_A_MELT_EROS, parent ? (dMe * rdt) * rarea * ms : 0.);
    if (dm & KID_DIAG_MELT_CONV) KID_ACC(KID_A_MELT_CONV, parent ? (dMv * rdt) * rarea * ms : 0.);
    const bool flp = parent && (dMfl > 0);
    if (dm & KID_DIAG_MELT_BUOY_FL) KID_ACC(KID_A_MELT_BUOY_FL, parent ? (flp ? (dMb_fl * rdt) * rarea * ms : 0.) : (dMb * rdt) * rarea * ms);
    if (dm & KID_DIAG_MELT_EROS_FL) KID_ACC(KID_A_MELT_EROS_FL, parent ? (flp ? (dMe_fl * rdt) * rarea * ms : 0.) : (dMe * rdt) * rarea * ms);
    if (dm & KID_DIAG_MELT_CONV_FL) KID_ACC(KID_A_MELT_CONV_FL, parent ? (flp ? (dMv_fl * rdt) * rarea * ms : 0.) : (dMv * rdt) * rarea * ms);
  }
  unsigned nerr = (active && area == 0.) ? 1u : 0u;  // FATAL 'berg appears to have grounded!' IB:3207
  if (Sw<K>::allow_bergs_to_roll(p) && N_bonds == 0.) rolling<K>(p, Tn, Wn, Ln);
  if (Sw<K>::Iceberg_melt_without_decay(p)) {  // IB:3214-3257
    if constexpr (TSPREAD) {  // IB:3219-3238
      BergThermo nb = b;
      nb.mass_of_fl_bits = Mnew_fl; nb.mass_of_fl_bergy_bits = nMbits_fl;
      if (Mnew > 0.) { nb.M = Mnew; nb.mass_of_bits = nMbits; nb.L = Ln; nb.W = Wn; nb.T = Tn; }
      else {  // the parent is gone but footloose bits remain: they stand in for it (addfootloose=.false.)
        const double M_edit = Lnfl * Wnfl * Tnfl * p.rho_bergs;
        nb.M = M_edit; nb.mass_of_bits = nMbits; nb.mass_scaling = Mnew_fl * ms / M_edit; nb.L = Lnfl; nb.W = Wnfl; nb.T = Tnfl;
        nb.mass_of_fl_bits = 0.; nb.mass_of_fl_bergy_bits = 0.;
      }
      spread_mass<K>(g, p, cellv, nb, uvel, vvel, i, j, ts->xi, ts->yj, active && (Mnew > 0. || Mnew_fl > 0.), acc, ncell, seg, ts->footprint, ts->theta);
    }
    Mnew = M;
    b.fl_k = fl_k;
  } else {
    b.M = Mnew; b.mass_of_bits = nMbits; b.mass_of_fl_bits = Mnew_fl; b.mass_of_fl_bergy_bits = nMbits_fl;
    b.T = Tn; b.W = dmin(Wn, Ln); b.L = dmax(Wn, Ln); b.fl_k = fl_k;
  }
  unsigned melted = 0u, calved = 0u;
  double fl_src = 0.;
  if (active && Mnew <= 0.) {  // IB:3271-3296
    if (Mnew_fl > 0) {
      calved = 1u;
      const double mass = Lnfl * Wnfl * Tnfl * p.rho_bergs;
      b.M = mass; b.L = Lnfl; b.W = Wnfl; b.T = Tnfl;
      nMbits_fl = nMbits_fl * ms;
      b.mass_scaling = Mnew_fl * ms / mass;
      b.mass_of_bits = nMbits_fl / b.mass_scaling;
      b.mass_of_fl_bits = 0.; b.mass_of_fl_bergy_bits = 0.; b.fl_k = -1.;
      b.start_year = p.current_year; b.start_day = p.current_yearday;
      if (area != 0.) fl_src = -(mass * b.mass_scaling / (dt * area));
    } else b.alive = false;
    melted = 1u;
  }
  if (__ballot(fl_src != 0.) != 0ull) KID_ACC(KID_A_FL_BITS_SRC, fl_src);
#undef KID_ACC
  const unsigned long long bm = __ballot(melted != 0u), bc = __ballot(calved != 0u), be = __ballot(nerr != 0u);
  if (__lane_id() == 0) {
    if (bm) unsafeAtomicAdd(scal + KID_S_NBERGS_MELTED, (double)__popcll(bm));
    if (bc) unsafeAtomicAdd(scal + KID_S_NBERGS_CALVED_FL, (double)__popcll(bc));
    if (be) unsafeAtomicAdd(scal + KID_S_ERROR_COUNT, (double)__popcll(be));
  }
}

// ---------------------------------------------------------------------------------------------------------
// hexagon footprint, IB:4136-4670 (only with hexagonal_icebergs; kept out of line: ~500 lines of branches)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double tri_area(double Ax, double Ay, double Bx, double By, double Cx, double Cy) {
  return fabs(0.5 * ((Ax * (By - Cy)) + (Bx * (Cy - Ay)) + (Cx * (Ay - By))));
}
__device__ __forceinline__ bool in_interval(double Ax, double Ay, double Bx, double By, double px, double py) {
  return (px <= dmax(Ax, Bx)) && (px >= dmin(Ax, Bx)) && (py <= dmax(Ay, By)) && (py >= dmin(Ay, By));
}
__device__ __forceinline__ bool on_line(double Ax, double Ay, double Bx, double By, double qx, double qy) {
  return fabs((qx - Ax) * (By - Ay) - (qy - Ay) * (Bx - Ax)) <= 0.0;
}
__device__ __forceinline__ bool point_in_triangle(double Ax, double Ay, double Bx, double By, double Cx, double Cy, double qx, double qy) {
  if ((Ax == qx && Ay == qy) || (Bx == qx && By == qy) || (Cx == qx && Cy == qy)) return false;
  if (on_line(Ax, Ay, Bx, By, qx, qy) || on_line(Ax, Ay, Cx, Cy, qx, qy) || on_line(Bx, By, Cx, Cy, qx, qy)) return false;
  const double l0 = (qx - Ax) * (By - Ay) - (qy - Ay) * (Bx - Ax);
  const double l1 = (qx - Bx) * (Cy - By) - (qy - By) * (Cx - Bx);
  const double l2 = (qx - Cx) * (Ay - Cy) - (qy - Cy) * (Ax - Cx);
  const double p0 = (l0 == 0.) ? 0. : sign1(l0), p1 = (l1 == 0.) ? 0. : sign1(l1), p2 = (l2 == 0.) ? 0. : sign1(l2);
  return ((fabs(p0) + fabs(p2)) + (fabs(p1))) == fabs((p0 + p2) + (p1));
}
template <bool XAXIS>
__device__ __forceinline__ void axis_intercept(double Ax, double Ay, double Bx, double By, double &x0, double &y0) {
  x0 = 100000000000.; y0 = 100000000000.;
  if (XAXIS) { if (Ay != By) { x0 = Ax - (((Ax - Bx) / (Ay - By)) * Ay); y0 = 0.; } }
  else { if (Ax != Bx) { x0 = 0.; y0 = -(((Ay - By) / (Ax - Bx)) * Ax) + Ay; } }
}
template <bool XAXIS>
__device__ __forceinline__ void tri_across_axis(double Ax, double Ay, double Bx, double By, double Cx, double Cy, double &Ap, double &An) {
  const double A_tri = tri_area(Ax, Ay, Bx, By, Cx, Cy);
  double pABx, pABy, pACx, pACy;
  axis_intercept<XAXIS>(Ax, Ay, Bx, By, pABx, pABy);
  axis_intercept<XAXIS>(Ax, Ay, Cx, Cy, pACx, pACy);
  const double A0 = XAXIS ? Ay : Ax;
  const double A_half = tri_area(Ax, Ay, pABx, pABy, pACx, pACy);
  if (A0 >= 0.) { Ap = A_half; An = A_tri - A_half; } else { Ap = A_tri - A_half; An = A_half; }
}
template <bool XAXIS>
__device__ __forceinline__ void tri_divide(double Ax, double Ay, double Bx, double By, double Cx, double Cy, double &Ap, double &An) {
  const double A0 = XAXIS ? Ay : Ax, B0 = XAXIS ? By : Bx, C0 = XAXIS ? Cy : Cx;
  const double A_tri = tri_area(Ax, Ay, Bx, By, Cx, Cy);
  Ap = 0.; An = 0.;
  if ((B0 * C0) > 0.) {
    if ((A0 * B0) >= 0.) { if ((A0 > 0.) || ((A0 == 0.) && (B0 > 0.))) Ap = A_tri; else An = A_tri; }
    else tri_across_axis<XAXIS>(Ax, Ay, Bx, By, Cx, Cy, Ap, An);
  } else if ((B0 * C0) < 0.) {
    if ((A0 * B0) >= 0.) tri_across_axis<XAXIS>(Cx, Cy, Bx, By, Ax, Ay, Ap, An);
    else tri_across_axis<XAXIS>(Bx, By, Cx, Cy, Ax, Ay, Ap, An);
  } else {
    if (((A0 == 0.) && (B0 == 0.)) && (C0 == 0.)) { }
    else if ((A0 * B0 < 0.) || (A0 * C0 < 0.)) tri_across_axis<XAXIS>(Ax, Ay, Bx, By, Cx, Cy, Ap, An);
    else if (((A0 * B0 > 0.) || (A0 * C0 > 0.)) || (((fabs(A0) > 0.) && (B0 == 0.)) && (C0 == 0.))) { if (A0 > 0.) Ap = A_tri; else An = A_tri; }
    else if (A0 == 0.) { if ((B0 > 0.) || (C0 > 0.)) Ap = A_tri; else if ((B0 < 0.) || (C0 < 0.)) An = A_tri; }
  }
}
__device__ __noinline__ void triangle_into_quadrants(double Ax, double Ay, double Bx, double By, double Cx, double Cy,
                                                     double &A_tri, double &Q1, double &Q2, double &Q3, double &Q4) {
  double Up, Lo, Ri, Le, px = 0, py = 0, qx = 0, qy = 0, Akey;
  int Key = 4;
  A_tri = tri_area(Ax, Ay, Bx, By, Cx, Cy);
  tri_divide<true>(Ax, Ay, Bx, By, Cx, Cy, Up, Lo);
  tri_divide<false>(Ax, Ay, Bx, By, Cx, Cy, Ri, Le);
  if (point_in_triangle(Ax, Ay, Bx, By, Cx, Cy, 0., 0.)) {
    axis_intercept<true>(Ax, Ay, Bx, By, px, py); axis_intercept<false>(Ax, Ay, Bx, By, qx, qy);
    if (!(in_interval(Ax, Ay, Bx, By, px, py) && in_interval(Ax, Ay, Bx, By, qx, qy))) {
      axis_intercept<true>(Ax, Ay, Cx, Cy, px, py); axis_intercept<false>(Ax, Ay, Cx, Cy, qx, qy);
      if (!(in_interval(Ax, Ay, Cx, Cy, px, py) && in_interval(Ax, Ay, Cx, Cy, qx, qy))) {
        axis_intercept<true>(Bx, By, Cx, Cy, px, py); axis_intercept<false>(Bx, By, Cx, Cy, qx, qy);
      }
    }
    Akey = tri_area(px, py, qx, qy, 0., 0.);
    if ((px >= 0.) && (qy >= 0.)) Key = 1; else if ((px < 0.) && (qy >= 0.)) Key = 2;
    else if ((px < 0.) && (qy < 0.)) Key = 3; else if ((px >= 0.) && (qy < 0.)) Key = 4;
  } else {
    Akey = 0;
    if ((!((((Ax > 0.) && (Ay > 0.)) || ((Bx > 0.) && (By > 0.))) || ((Cx > 0.) && (Cy > 0.)))) && ((Up + Ri) <= A_tri)) Key = 1;
    else if ((!((((Ax < 0.) && (Ay > 0)) || ((Bx < 0.) && (By > 0.))) || ((Cx < 0.) && (Cy > 0.)))) && ((Up + Le) <= A_tri)) Key = 2;
    else if ((!((((Ax < 0.) && (Ay < 0.)) || ((Bx < 0.) && (By < 0.))) || ((Cx < 0.) && (Cy < 0.)))) && ((Lo + Le) <= A_tri)) Key = 3;
    else Key = 4;
  }
  if (Key == 1) { Q1 = Akey; Q2 = Up - Q1; Q4 = Ri - Q1; Q3 = A_tri - (Q1 + Q2 + Q4); }
  else if (Key == 2) { Q2 = Akey; Q1 = Up - Q2; Q4 = Ri - Q1; Q3 = A_tri - (Q1 + Q2 + Q4); }
  else if (Key == 3) { Q3 = Akey; Q2 = Le - Q3; Q1 = Up - Q2; Q4 = A_tri - (Q1 + Q2 + Q3); }
  else { Q4 = Akey; Q1 = Ri - Q4; Q2 = Up - Q1; Q3 = A_tri - (Q1 + Q2 + Q4); }
  Q1 = dmax(Q1, 0.); Q2 = dmax(Q2, 0.); Q3 = dmax(Q3, 0.); Q4 = dmax(Q4, 0.);
}
__device__ __noinline__ void hexagon_into_quadrants(const kid_params &p, double x0, double y0, double H, double theta,
                                                    double &A_hex, double &Q1, double &Q2, double &Q3, double &Q4) {
  const double S = (2 / sqrt(3.)) * H, r3 = H / sqrt(3.);
  double Cx[6] = {S, r3, -r3, -S, -r3, r3}, Cy[6] = {0., H, H, 0., -H, -H};
  const double ct = cos(theta * p.pi / 180), st = sin(theta * p.pi / 180);
#pragma unroll
  for (int k = 0; k < 6; ++k) {  // rotate_and_translate IB:4537-4554
    const double xt = (ct * Cx[k]) + (st * Cy[k]), yt = (-st * Cx[k]) + (ct * Cy[k]);
    Cx[k] = xt + x0; Cy[k] = yt + y0;
  }
  A_hex = 0.; Q1 = 0.; Q2 = 0.; Q3 = 0.; Q4 = 0.;
  for (int k = 0; k < 6; ++k) {
    const int k2 = (k + 1) % 6;
    double a, q1, q2, q3, q4;
    triangle_into_quadrants(x0, y0, Cx[k], Cy[k], Cx[k2], Cy[k2], a, q1, q2, q3, q4);
    A_hex += a; Q1 += q1; Q2 += q2; Q3 += q3; Q4 += q4;  // same left-to-right order as IB:4618-4622
  }
  Q1 = dmax(Q1, 0.); Q2 = dmax(Q2, 0.); Q3 = dmax(Q3, 0.); Q4 = dmax(Q4, 0.);
  const double Error = A_hex - (Q1 + Q2 + Q3 + Q4);
  if (((Q1 >= Q2) && (Q1 >= Q3)) && (Q1 >= Q4)) Q1 = Q1 + Error;
  else if (((Q2 >= Q1) && (Q2 >= Q3)) && (Q2 >= Q4)) Q2 = Q2 + Error;
  else if (((Q3 >= Q1) && (Q3 >= Q2)) && (Q3 >= Q4)) Q3 = Q3 + Error;
  else if (((Q4 >= Q1) && (Q4 >= Q2)) && (Q4 >= Q3)) Q4 = Q4 + Error;
}

// ---------------------------------------------------------------------------------------------------------
// IB:3895-4133 spread_mass_across_ocean_cells + calculate_sum_over_bergs_diagnositcs (IB:5014-5071)
// ---------------------------------------------------------------------------------------------------------
template <int K, class CELL>
__device__ __forceinline__ void spread_mass(const DevGrid &g, const kid_params &p, const CELL &cellv, const BergThermo &b, double uvel, double vvel,
                                            int i, int j, double x, double y, bool active, double *acc, size_t ncell, Seg &seg, bool footprint, double theta) {
  constexpr double rho_sw = 1035.;  // IB:3919 shadows the module's 1025
  const int c = g.idx(i, j);
  const double a_ij = cellv.area();
  const bool ok = active && (a_ij > 0.);  // IB:4994
  const double Area = b.L * b.W, Tn = b.T, scaling = b.mass_scaling;
  double Mass_berg = b.M, Mfl = b.mass_of_fl_bits;
  const double Mbits_fl = b.mass_of_fl_bergy_bits;
  if (Sw<K>::grounding_fraction(p) > 0.) {  // IB:3941-3950
    const double Hocean = Sw<K>::grounding_fraction(p) * (g.ocean_depth[c] + g.ssh[c]);
    double Dn = (p.rho_bergs / rho_sw) * Tn;
    if (Dn > Hocean) Mass_berg = Mass_berg * dmin(1., Hocean / Dn);
    if (Mfl > 0.) {
      double Lfl, Wfl, Tfl; fl_bits_dimensions(p, b.T, Lfl, Wfl, Tfl);
      Dn = (p.rho_bergs / rho_sw) * Tfl;
      if (Dn > Hocean) Mfl = Mfl * dmin(1., Hocean / Dn);
    }
  }
  Mass_berg = Mass_berg + Mfl;
  double Mass = (Mass_berg + b.mass_of_bits + Mbits_fl) * scaling;
  if (Sw<K>::clipping_depth(p) > 0.) Mass = dmin(Mass, Sw<K>::clipping_depth(p) * a_ij * rho_sw);
  double w[9] = {0., 0., 0., 0., 1., 0., 0., 0., 0.};  // yDxL,yDxC,yDxR,yCxL,yCxC,yCxR,yUxL,yUxC,yUxR
  double fraction_used = 1.;
#define KID_M(di, dj) cellv.msk((di), (dj))
  if (!Sw<K>::hexagonal_icebergs(p)) {
    const double L = (a_ij > 0) ? dmin(sqrt(Area / a_ij), 1.0) : 1.;
    double xL, xR, xC, yD, yU, yC;
    if (Sw<K>::use_old_spreading(p)) {
      xL = dmin(0.5, dmax(0., 0.5 - x)); xR = dmin(0.5, dmax(0., x - 0.5)); xC = dmax(0., 1. - (xL + xR));
      yD = dmin(0.5, dmax(0., 0.5 - y)); yU = dmin(0.5, dmax(0., y - 0.5)); yC = dmax(0., 1. - (yD + yU));
    } else {
      xL = dmin(0.5, dmax(0., 0.5 - (x / L))); xR = dmin(0.5, dmax(0., (x / L) + (0.5 - (1 / L)))); xC = dmax(0., 1. - (xL + xR));
      yD = dmin(0.5, dmax(0., 0.5 - (y / L))); yU = dmin(0.5, dmax(0., (y / L) + (0.5 - (1 / L)))); yC = dmax(0., 1. - (yD + yU));
    }
    w[0] = yD * xL * KID_M(-1, -1); w[1] = yD * xC * KID_M(0, -1); w[2] = yD * xR * KID_M(1, -1);
    w[3] = yC * xL * KID_M(-1, 0); w[5] = yC * xR * KID_M(1, 0);
    w[6] = yU * xL * KID_M(-1, 1); w[7] = yU * xC * KID_M(0, 1); w[8] = yU * xR * KID_M(1, 1);
    w[4] = 1. - (((w[0] + w[8]) + (w[2] + w[6])) + ((w[3] + w[5]) + (w[1] + w[7])));
  } else {
    const double H = (a_ij > 0) ? dmin(((sqrt(Area / (2. * sqrt(3.))) / sqrt(a_ij))), 1.) : (sqrt(3.) / 2) * (0.49);
    const double origin_x = (x < 0.5) ? 0. : 1., origin_y = (y < 0.5) ? 0. : 1.;
    double Ah, Q1, Q2, Q3, Q4;
    hexagon_into_quadrants(p, x - origin_x, y - origin_y, H, theta, Ah, Q1, Q2, Q3, Q4);  // theta: IB:4003-4004
    Q1 = Q1 / Ah; Q2 = Q2 / Ah; Q3 = Q3 / Ah; Q4 = Q4 / Ah;
    w[4] = 1.;
    if ((x >= 0.5) && (y >= 0.5)) { w[8] = Q1; w[7] = Q2; w[4] = Q3; w[5] = Q4; }
    else if ((x < 0.5) && (y >= 0.5)) { w[7] = Q1; w[6] = Q2; w[3] = Q3; w[4] = Q4; }
    else if ((x < 0.5) && (y < 0.5)) { w[4] = Q1; w[3] = Q2; w[0] = Q3; w[1] = Q4; }
    else if ((x >= 0.5) && (y < 0.5)) { w[5] = Q1; w[4] = Q2; w[1] = Q3; w[2] = Q4; }
    fraction_used = ((w[0] * KID_M(-1, -1)) + (w[1] * KID_M(0, -1)) + (w[2] * KID_M(1, -1)) + (w[3] * KID_M(-1, 0)) + (w[5] * KID_M(1, 0))
                     + (w[6] * KID_M(-1, 1)) + (w[7] * KID_M(0, 1)) + (w[8] * KID_M(1, 1)) + (kid_pow(w[4], KID_M(0, 0))));  // `**` IB:4081
    if (b.static_berg == 1) fraction_used = 1.;
  }
#undef KID_M
  const double Ifu = 1. / fraction_used;
  const double vars[4] = {Mass, Area * scaling, uvel * Area * scaling, vvel * Area * scaling};
  const int base[4] = {KID_A_MASS_ON_OCEAN, KID_A_AREA_ON_OCEAN, KID_A_UVEL_ON_OCEAN, KID_A_VVEL_ON_OCEAN};
#pragma unroll
  for (int s = 0; s < 9; ++s) {
    // slot unused by the whole wave: skipped (the plain build stages all nine, so that the number of staged values -- and
    // with it the one place where the staging block is flushed -- is known at compile time)
    if (K != 1 && K != 3 && __ballot(ok && w[s] != 0.) == 0ull) continue;
    cell_add(acc, ncell, base[0] + s, c, w[s] * vars[0] * Ifu, seg, ok);
    if (footprint) {  // wave-uniform: area / Uvel / Vvel footprints only when something downstream reads them
#pragma unroll
      for (int v = 1; v < 4; ++v) cell_add(acc, ncell, base[v] + s, c, w[s] * vars[v] * Ifu, seg, ok);
    }
  }
}

template <int K = 0, class CELL>
__device__ __forceinline__ void berg_diagnostics(const DevGrid &g, const kid_params &p, const CELL &cellv, const BergThermo &b, double uvel, double vvel,
                                                 int i, int j, bool active, double *acc, size_t ncell, Seg &seg) {
  const int c = g.idx(i, j);
  const double area = cellv.area(), ms = b.mass_scaling;
  const bool ok = active && (area > 0.);
  const int dm = Sw<K>::diag_mask(p);
#define KID_ACC(F, v) cell_add(acc, ncell, (F), c, (v), seg, ok)
  if (dm & KID_DIAG_VIRTUAL_AREA) {
    double Abits = 0., Abits_fl = 0., Abits_fl_bergy = 0.;
    if (Sw<K>::bergy_bit_erosion_fraction(p) > 0.) Abits = (b.mass_of_bits / p.rho_bergs) / dmin(dmin(dmin(b.L, b.W), b.T), 40.);
    if (p.fl_style == KID_FL_STYLE_FL_BITS) {
      double L_fl, W_fl, T_fl; fl_bits_dimensions(p, b.T, L_fl, W_fl, T_fl);
      Abits_fl = (b.mass_of_fl_bits / p.rho_bergs) / T_fl;
      if (Sw<K>::bergy_bit_erosion_fraction(p) > 0.) Abits_fl_bergy = (b.mass_of_fl_bergy_bits / p.rho_bergs) / dmin(dmin(dmin(L_fl, W_fl), T_fl), 40.);
    }
    KID_ACC(KID_A_VIRTUAL_AREA, (b.W * b.L + Abits + Abits_fl + Abits_fl_bergy) * ms);
  }
  if (dm & (KID_DIAG_MASS | KID_DIAG_U_ICEBERG | KID_DIAG_V_ICEBERG)) KID_ACC(KID_A_MASS, b.M / area * ms);
  if (dm & KID_DIAG_U_ICEBERG) KID_ACC(KID_A_U_ICEBERG, ((b.M / area * ms) * uvel));
  if (dm & KID_DIAG_V_ICEBERG) KID_ACC(KID_A_V_ICEBERG, ((b.M / area * ms) * vvel));
  if ((dm & KID_DIAG_BERGY_MASS) || Sw<K>::add_weight_to_ocean(p)) {
    const double v = (b.mass_of_bits + b.mass_of_fl_bergy_bits) / area * ms;
    if (__ballot(ok && v != 0.) != 0ull) KID_ACC(KID_A_BERGY_MASS, v);
  }
  if ((dm & KID_DIAG_FL_BITS_MASS) || Sw<K>::add_weight_to_ocean(p)) {
    const double v = b.mass_of_fl_bits / area * ms;
    if (__ballot(ok && v != 0.) != 0ull) KID_ACC(KID_A_FL_BITS_MASS, v);
  }
  if ((dm & KID_DIAG_FL_BERGY_BITS_MASS) || Sw<K>::add_weight_to_ocean(p)) {
    const double v = b.mass_of_fl_bergy_bits / area * ms;
    if (__ballot(ok && v != 0.) != 0ull) KID_ACC(KID_A_FL_BERGY_BITS_MASS, v);
  }
#undef KID_ACC
}

}  // namespace kid

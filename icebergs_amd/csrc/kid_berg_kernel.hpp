// kid_berg_kernel.hpp -- the per-berg kernel of the evolve loop (hot and general builds) and the tables it reads.
// Included by kid_hip.hip; a header of its own so that tools/profiling/hot_only.hip can compile one instantiation alone
// (ISA, registers) in seconds.
#pragma once
#include "kid_device.hpp"
#include "kid_thermo.hpp"
#include "kid_footloose.hpp"

namespace {
using namespace kid;

enum : unsigned { PH_INTERP = 1u, PH_EVOLVE = 2u, PH_THERMO = 4u, PH_SPREAD = 8u, PH_FL = 16u, PH_TSPREAD = 32u };   // PH_FL: footloose_calving between evolve and thermodynamics

struct BergPtrs {
  double *f[KID_NB_F64];
  int32_t *i[KID_NB_I32];
  int64_t *id;
  const double *orient;   // per-berg hexagon orientation from the bonds (IB:4004), or null: initial_orientation
};
struct Flags { int has_static, has_fl, store_env, footprint, no_diag; };   // no_diag: calculate_mass_on_ocean(with_diagnostics=.false.)  // footprint: area/Uvel/Vvel_on_ocean are read by somebody
// The SoA arrays are reached through pointers read from a table in memory, which the compiler can only address as
// generic (flat) pointers: flat loads count against the LDS counter as well as the memory counter, so every wait for an
// LDS read would also wait for the loads still in flight.  These go through the global address space explicitly.
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) int32_t gint32;
typedef __attribute__((address_space(1))) char gchar;
// Row k of a field: the field's base is wave-uniform (a scalar register pair read from the table) and the byte offset of the row
// fits 32 bits (kid_create refuses more than 2^29 rows), so one 32-bit offset register serves every field of the row
// (global_load ... v_off, s[base:base+1]) instead of a 64-bit address computed per field and row.
__device__ __forceinline__ double ldg(const double *q, long long k) { return *(const gdouble *)((const gchar *)q + (size_t)((unsigned)k << 3)); }
__device__ __forceinline__ int32_t ldg(const int32_t *q, long long k) { return *(const gint32 *)((const gchar *)q + (size_t)((unsigned)k << 2)); }
// keep(x): an empty use of x.  The per-berg loads at the head of the kernel are written before the early exit of an
// all-dead wave so that they are all in flight at once; values used on the live path only would be sunk below that branch
// by the compiler, behind the wait for `alive` -- a second round trip to HBM per wave.  The dead path "uses" them too.
__device__ __forceinline__ void keep(double x) { asm volatile("" ::"v"(x)); }
__device__ __forceinline__ void keep(int32_t x) { asm volatile("" ::"v"(x)); }
__device__ __forceinline__ void stg(double *q, long long k, double v) { *(gdouble *)((gchar *)q + (size_t)((unsigned)k << 3)) = v; }
__device__ __forceinline__ void stg(int32_t *q, long long k, int32_t v) { *(gint32 *)((gchar *)q + (size_t)((unsigned)k << 2)) = v; }
// the plain builds (K = 1 and, with store_env on, K = 3; kid_device.hpp) are launched only with the other four flags zero; the footloose profile (K = 2) with no static
// bergs, footloose state present, no footprint planes and the diagnostics call on (store_env as the handle says)
template <int K> struct Fl {
#define KID_X(name, k1, k2, k3) static __device__ __forceinline__ int name(const Flags &f) { if constexpr (K == 1) return k1; else if constexpr (K == 2) return k2; else if constexpr (K == 3) return k3; else return f.name; }
  KID_X(has_static, 0, 0, 0) KID_X(has_fl, 0, 1, 0) KID_X(store_env, 0, f.store_env, 1) KID_X(footprint, 0, 0, 0) KID_X(no_diag, 0, 0, 0)
#undef KID_X
};

// -------------------------------------------------------------------------------------------------------
// the per-berg kernel
// -------------------------------------------------------------------------------------------------------
#ifndef KID_WAVES_PER_EU
#define KID_WAVES_PER_EU 2
#endif
// Two builds of the same body share the work of one phase:
//   FAST=true  : every berg, specialised for the overwhelmingly common case (stays in its cell, not at the pole).
//                A berg that meets anything else is left untouched and its index is appended to `redo`.
//   FAST=false : the general code (cell hops, coast bounce, polar cells, tangent plane) over the `redo` list.
// Keeping the rare branches out of the hot build roughly halves its register footprint (2 waves/SIMD, no scratch).
// Threads per workgroup of the hot build.  A workgroup's LDS and registers are released when its LAST wave ends: with four
// waves per workgroup a SIMD slot whose wave finished early (few runs, no bails) idles until the slowest of the four is done
// (measured at 1e7 bergs: 256 threads 1.143 ms per launch, 128: 1.114, 64: 1.091).
#ifndef KID_HOT_WG
#define KID_HOT_WG 64
#endif
#ifdef KID_EXP_NO_PRIO
#define KID_PRIO_ON false
#else
#define KID_PRIO_ON true
#endif
#ifndef KID_GENERAL_WAVES_PER_EU
#define KID_GENERAL_WAVES_PER_EU 2   // <=256 registers: a general-build wave can share a SIMD with a hot-build wave (pipelined mode)
#endif
struct Redo { int *list; int *count; long long k0, klen; int *lane; int step;    // k0, klen: the rows the hot build covers in this launch
              int *fl_cursor; int32_t *fl_counter; long long fl_capacity; int fl_iNg; unsigned fl_step; };   // PH_FL: where footloose children go (FlChildCtx)
// lane/step ("slow lane" schedule, launch_berg_lanes): lane[k] >= step means berg k is owned by general-build launches that
// may still be running on the side stream; the hot build of this step leaves it alone.  A berg the hot build hands over
// at step s gets lane = s + 1: the general build does its steps s and s + 1, the hot build has it back at s + 2.
#ifdef KID_EXP_NUM_VGPR
#define KID_NUM_VGPR_ATTR __attribute__((amdgpu_num_vgpr(KID_EXP_NUM_VGPR)))
#else
#define KID_NUM_VGPR_ATTR
#endif
// The plain hot build of the fused RK4 step (BASELINE configs 1-2 and the headline line) runs THREE waves per SIMD: with the four
// stages unrolled it needs 160 registers (the rolled loop 203: the stage-dependent selects and the loop-carried copies; 168 is
// what a third wave allows); and its LDS fits twelve times into a CU (160 KB in 1280-byte blocks: 12800 bytes per wave) with
// KID_HOT3_SLOTS cell packets and KID_HOT3_CHUNK staging rows.  Measured at 1e7 bergs: 0.98 -> 0.88 ms per launch with 12 + 10;
// 11 + 11 holds the step's eleven staged values (floating_melt, berg_melt, nine mass_on_ocean slots; the heat-flux plane is
// staged only by a wave that has heat) in ONE flush instead of two: 0.805 -> 0.777 ms.
#ifndef KID_FLP_SLOTS
#define KID_FLP_SLOTS KID_MAXRUN
#define KID_FLP_CHUNK 10   // (rows 0-9 park the footloose build's values; it stages at most nine: 14.8 KB per wave = 12 LDS blocks, ten waves per CU instead of nine)
#endif
#ifndef KID_HOT3_SLOTS
#define KID_HOT3_SLOTS 11
#define KID_HOT3_CHUNK 11
#endif
template <bool RK, bool OLD_ORDER, unsigned PH, bool FAST, int K> struct HotCfg {
#if defined(KID_EXACT_MATH) || defined(KID_EXP_NO_HOT3)
  static constexpr bool three = false;
#else
  static constexpr bool three = FAST && RK && OLD_ORDER && (K == 1 || K == 3) && (PH & PH_EVOLVE) != 0 && KID_HOT_WG == 64;
#endif
  static constexpr int slots = three ? KID_HOT3_SLOTS : ((FAST && K == 2) ? KID_FLP_SLOTS : KID_MAXRUN);   // cell packets per wave
  static constexpr int chunk = three ? KID_HOT3_CHUNK : ((FAST && K == 2) ? KID_FLP_CHUNK : KID_CHUNK);    // staging rows per wave (>= 7: the rows the plain build parks M .. heat_density in)
  static constexpr int waves = !FAST ? KID_GENERAL_WAVES_PER_EU : (three ? 3 : KID_WAVES_PER_EU);
};
template <bool RK, bool OLD_ORDER, unsigned PH, bool FAST, int K = 0>
__global__ void KID_NUM_VGPR_ATTR __launch_bounds__(FAST ? KID_HOT_WG : 256, (HotCfg<RK, OLD_ORDER, PH, FAST, K>::waves)) berg_kernel(const DevGrid *__restrict__ gtab, const kid_params *__restrict__ pp, const BergPtrs *__restrict__ bt, const long long n,
                                                   double *__restrict__ acc, const size_t ncell, const Flags fl, const Redo redo) {
  // The parameter block (142 dwords) and the 51 field pointers are read through device-memory tables on demand:
  // as by-value kernel arguments they were all pinned in SGPRs, overflowed the scalar file and came back as
  // thousands of v_readlane spill reloads per wave.
  const kid_params &p = *pp;
  const BergPtrs &b = *bt;
  const DevGrid &g = *gtab;   // like the other two tables: read on demand, not pinned in ~50 SGPRs for the whole kernel
  constexpr bool SCATTER = (PH & (PH_THERMO | PH_SPREAD)) != 0;
  constexpr int WG_WAVES = FAST ? KID_HOT_WG / 64 : 4;   // (the general build is launched one wave per workgroup; its other entry points with up to four)
  using Cfg = HotCfg<RK, OLD_ORDER, PH, FAST, K>;
  constexpr int SLOTS = Cfg::slots, CHUNK = Cfg::chunk;
  __shared__ double lds_vals[SCATTER ? seg_lds_doubles(WG_WAVES, CHUNK) : 1];   // staging of the per-cell sums (kid_thermo.hpp)
  __shared__ int lds_ints[seg_lds_ints(WG_WAVES, CHUNK)];                         // run tables of the workgroup's waves
  __shared__ __attribute__((aligned(16))) double lds_pk[FAST ? WG_WAVES * SLOTS * PK_STRIDE : 2];   // cell packets of the workgroup's waves (hot build)
  // FAST: one pass over all bergs.  General: grid-stride over the (short) redo list.
  const long long total = FAST ? redo.klen : (long long)(*redo.count);
  const long long bdim = FAST ? (long long)KID_HOT_WG : (long long)blockDim.x;   // the general build is launched with one wave per workgroup
  // (the hot build's "loop" visibly runs once: otherwise the compiler hoists the constants of the whole body out of it and
  // holds -- or spills -- them in vector registers)
  long long tid = (long long)blockIdx.x * bdim + threadIdx.x;
  for (bool first = true; FAST ? first : (tid - threadIdx.x < total); first = false, tid += (long long)gridDim.x * bdim) {
  KID_TICK(-1);
  // The wave that is fetching issues first: its loads, the run table and the packet DMA are a few hundred instructions that would
  // otherwise take turns with the other wave's arithmetic, and every cycle they finish earlier is a cycle of memory latency that
  // overlaps with that arithmetic (measured at 1e7 bergs: 1.033 -> 1.016 ms per launch with the flush below)
  if (FAST && KID_PRIO_ON) __builtin_amdgcn_s_setprio(3);
  const bool inrange = tid < total;
  const long long k = inrange ? (FAST ? redo.k0 + tid : (long long)redo.list[tid]) : 0ll;
  const long long kk = inrange ? k : (n - 1);
  // Every load of the step is issued here, back to back, before anything is used: `alive`, the lane stamp, the cell and
  // the fields used to be three dependent round trips to HBM at the head of every wave (~15 % of its lifetime).
  const int32_t alive_v = ldg(b.i[KID_BI_ALIVE], kk);
  // (no branch around the lane-stamp load -- the join would wait for it: without stamps it re-reads `alive`)
  const int32_t lane_v = FAST ? ldg(redo.lane ? (const int32_t *)redo.lane : (const int32_t *)b.i[KID_BI_ALIVE], kk) : 0;
  BergDyn d;
  d.ine = ldg(b.i[KID_BI_INE], kk); d.jne = ldg(b.i[KID_BI_JNE], kk);
  d.xi = ldg(b.f[KID_B_XI], kk); d.yj = ldg(b.f[KID_B_YJ], kk);
  d.lon = ldg(b.f[KID_B_LON], kk); d.lat = ldg(b.f[KID_B_LAT], kk);
  d.uvel = ldg(b.f[KID_B_UVEL], kk); d.vvel = ldg(b.f[KID_B_VVEL], kk);
  d.uvel_prev = 0.; d.vvel_prev = 0.;
  d.axn = 0.; d.ayn = 0.; d.bxn = 0.; d.byn = 0.;
  if (PH & PH_EVOLVE) {
    d.axn = ldg(b.f[KID_B_AXN], kk); d.ayn = ldg(b.f[KID_B_AYN], kk);
    if (!RK) { d.bxn = ldg(b.f[KID_B_BXN], kk); d.byn = ldg(b.f[KID_B_BYN], kk); }
  }
  BergThermo t;
  t.M = ldg(b.f[KID_B_MASS], kk); t.T = ldg(b.f[KID_B_THICKNESS], kk); t.W = ldg(b.f[KID_B_WIDTH], kk); t.L = ldg(b.f[KID_B_LENGTH], kk);
  t.n_bonds = Sw<K>::iceberg_bonds_on(p) ? ldg(b.i[KID_BI_N_BONDS], kk) : 0;
  t.static_berg = Fl<K>::has_static(fl) ? ldg(b.f[KID_B_STATIC_BERG], kk) : 0.;
  const bool halo = Fl<K>::has_static(fl) ? (ldg(b.f[KID_B_HALO_BERG], kk) >= 0.5) : false;
  // The hot build of the fused RK4 step parks what only the thermodynamics needs in its wave's staging rows (idle until
  // the first cell_add) instead of holding the registers -- or re-reading HBM -- across the RK4 loop.
  constexpr bool SCATTER_ = (PH & (PH_THERMO | PH_SPREAD)) != 0;
  constexpr bool PARK = FAST && SCATTER_ && (PH & PH_EVOLVE) != 0;
  // footloose builds: what the footloose phase does not need either (bits, heat density, the bergy bits of the footloose bits)
  // stays in its LDS row until the thermodynamics asks for it
  constexpr bool LATE_PARK = PARK && (PH & PH_FL) != 0;
  double park_ms = 0., park_bits = 0., park_hd = 0., park_flk = 0., park_flbits = 0., park_flbergy = 0.;
  if constexpr (PARK) {
    park_ms = ldg(b.f[KID_B_MASS_SCALING], kk); park_bits = ldg(b.f[KID_B_MASS_OF_BITS], kk);
    park_hd = (PH & PH_THERMO) ? ldg(b.f[KID_B_HEAT_DENSITY], kk) : 0.;
    if (Fl<K>::has_fl(fl)) {
      park_flk = ldg(b.f[KID_B_FL_K], kk); park_flbits = ldg(b.f[KID_B_MASS_OF_FL_BITS], kk); park_flbergy = ldg(b.f[KID_B_MASS_OF_FL_BERGY_BITS], kk);
    }
  }
  bool was_alive = inrange && (alive_v != 0);
  if (FAST && redo.lane) { if (was_alive && lane_v >= redo.step) was_alive = false; }
  KID_TICK(11);   // (the wait for `alive` / the lane stamp: the first round trip)
  if (__ballot(was_alive) == 0ull) {  // wave-uniform; every other lane stays to the end (wave-level sums below)
    keep(d.ine); keep(d.jne); keep(d.xi); keep(d.yj); keep(d.lon); keep(d.lat); keep(d.uvel); keep(d.vvel); keep(d.axn); keep(d.ayn);
    keep(d.bxn); keep(d.byn); keep(t.M); keep(t.T); keep(t.W); keep(t.L); keep(t.n_bonds); keep(t.static_berg);
    keep(park_ms); keep(park_bits); keep(park_hd); keep(park_flk); keep(park_flbits); keep(park_flbergy);
    continue;
  }
  double *scal = acc - KID_NSCALAR;   // the step's scalar increments sit in front of plane 0 (kid_accum_device_ptr)

  // runs of equal cell among the 64 lanes (the SoA is cell-sorted): shared by the packet staging and the scatter
  // (the hot build finds its packet cells from the lanes' own keys and builds the run tables of the scatter after the packet
  // loads are on their way)
  const int mykey = was_alive ? g.idx(d.ine, d.jne) : -1;
  // sin / cos of the cell's reference latitude for the first RK4 stage (lat_terms_cell): fetched with the cell packets
  double latref_s = 0., latref_c = 1.;
  if constexpr (FAST && RK && (PH & PH_EVOLVE) != 0) {
    if (grid_latlon<K>(g) && g.latref) { const long long c2 = 2ll * (mykey < 0 ? 0 : mykey); latref_s = ldg(g.latref, c2); latref_c = ldg(g.latref, c2 + 1); }
  }
  Seg seg;
  if constexpr (!FAST) seg = make_runs(mykey, (lds_double *)lds_vals, (lds_int *)lds_ints, CHUNK);
  const lds_double *pk = nullptr;
  if (FAST) {
    lds_double *wpk = (lds_double *)lds_pk + (threadIdx.x >> 6) * (SLOTS * PK_STRIDE);
    const int lane = (int)__lane_id();
    // One packet slot per DISTINCT cell of the wave's lanes.  As the cell order decays between two re-binnings a wave collects
    // out-of-place bergs, each a run of its own that also splits the run it sits in: a slot per run staged the same packet
    // again and again and ran out of slots two thirds into a 16-step interval (runs ~ 1.5 + 1.3 per step at 139 bergs per
    // cell), although the bergs of a tile only ever spread over the handful of cells around where they were binned.  The sparse
    // population of the footloose profile (5 bergs per cell, 13 cells per wave when freshly binned) pays ~2 % for the search
    // while its order is fresh and gains 25 % over a whole interval: a berg that hops to the next cell in i usually lands in a
    // cell the wave already holds (config 3, 32 steps: 2.02 -> 1.51 ms/step).
    // The packets go from memory straight into LDS (gfx950: global_load_lds_dwordx4, 16 bytes per lane, lane l to LDS base + 16 l):
    // one instruction per cell with the first 34 lanes moves the 544-byte packet, no staging registers, and every cell's load is
    // in flight before the single wait.  The cell of a slot is wave-uniform (a v_readlane): scalar address arithmetic.
    const gchar *gp = (const gchar *)g.pkt;
    static_assert(PK_SIZE * 8 == 34 * 16 && (PK_STRIDE * 8) % 16 == 0 && (PK_GSTRIDE * 8) % 16 == 0, "packet = 34 lanes x 16 bytes, 16-byte aligned slots");
    int myslot;
    {
    // every lane knows its own cell: the wave walks over the distinct cells of its live lanes (first lane not yet served, its
    // cell by v_readlane, a ballot of the lanes that share it) -- no table look-up, no shuffle
    const unsigned loff = (unsigned)lane * 16u;
    unsigned long long rem = __ballot(mykey >= 0);
    int nslot = 0;
    myslot = 0;
    while (rem != 0ull) {  // wave-uniform: one turn per distinct cell
      const int u = (int)__ffsll((long long)rem) - 1;
      const int cu = __builtin_amdgcn_readlane(mykey, u);
      const bool mine = mykey == cu;
      if (mine) myslot = nslot;
      if (nslot < SLOTS && lane < 34)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp + (size_t)cu * (size_t)(PK_GSTRIDE * 8) + loff),
                                         (__attribute__((address_space(3))) void *)(wpk + nslot * PK_STRIDE), 16, 0, 0);
      rem &= ~__ballot(mine);
      ++nslot;
    }
    seg = make_runs(mykey, (lds_double *)lds_vals, (lds_int *)lds_ints, CHUNK);
    // more distinct cells than slots: the lanes of the cells beyond go to the general build one by one (handing over the whole
    // wave made 15 % of the population take the slow path by the end of a 16-step interval)
    if (myslot >= SLOTS) {
      if (was_alive) { const int slot = atomicAdd(redo.count, 1); redo.list[slot] = (int)kk; if (redo.lane) redo.lane[kk] = redo.step + 1; }
      was_alive = false;
    }
    }
    if (KID_PRIO_ON) __builtin_amdgcn_s_setprio(0);
    KID_TICK(12);   // (runs found, packet loads issued)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    KID_TICK(13);   // (everything has arrived)
    if (__ballot(was_alive) == 0ull) continue;
    pk = wpk + (myslot < SLOTS ? myslot : 0) * PK_STRIDE;
  }
  t.alive = was_alive;   // (after the hot build has handed the lanes of its surplus runs to the general build)
  if constexpr (PARK) {
    lds_double *row = seg.val + (int)__lane_id();
    row[0 * KID_ROW] = t.M; row[1 * KID_ROW] = t.T; row[2 * KID_ROW] = t.W; row[3 * KID_ROW] = t.L;
    row[4 * KID_ROW] = park_ms; row[5 * KID_ROW] = park_bits; row[6 * KID_ROW] = park_hd;
    if (Fl<K>::has_fl(fl)) { row[7 * KID_ROW] = park_flk; row[8 * KID_ROW] = park_flbits; row[9 * KID_ROW] = park_flbergy; }
  }
  Env e = {};
  unsigned tickets = 0u;
  int err = 0;
  bool env_dirty = false;
  bool bail = false;     // FAST build: this berg needs the general build
  bool skipped = false;  // ... and has been queued: nothing of it may be written or accumulated here

  if ((PH & PH_INTERP) || (!OLD_ORDER && (PH & (PH_EVOLVE | PH_THERMO)))) {
    if (PH & PH_INTERP) {  // IB:4673-4715
      if (was_alive && !halo) { interp_flds<K>(p, CellOf<FAST>::make(g, pk, d.ine, d.jne), d.xi, d.yj, e); env_dirty = true; }
    } else {  // stored environment (.not.old_interp_flds_order), IB:2039-2040
      e.uo = ldg(b.f[KID_B_UO], kk); e.vo = ldg(b.f[KID_B_VO], kk); e.ui = ldg(b.f[KID_B_UI], kk); e.vi = ldg(b.f[KID_B_VI], kk);
      e.ua = ldg(b.f[KID_B_UA], kk); e.va = ldg(b.f[KID_B_VA], kk); e.ssh_x = ldg(b.f[KID_B_SSH_X], kk); e.ssh_y = ldg(b.f[KID_B_SSH_Y], kk);
      e.sst = ldg(b.f[KID_B_SST], kk); e.sss = ldg(b.f[KID_B_SSS], kk); e.cn = ldg(b.f[KID_B_CN], kk); e.hi = ldg(b.f[KID_B_HI], kk); e.od = ldg(b.f[KID_B_OD], kk);
    }
  }

  if (!RK) KID_TICK(2);   // (Verlet builds: prologue + the interpolation before the evolve)
  if (PH & PH_EVOLVE) {  // IB:7081-7179
    const bool moves = was_alive && (t.static_berg < 0.5);
    if (moves) {
      const BergGeom bg{t.M, t.T, t.W, t.L, t.n_bonds};
      if (RK) {
        if constexpr (!FAST) { if (grid_latlon<K>(g) && g.latref) { const long long c2 = 2ll * g.idx(d.ine, d.jne); latref_s = ldg(g.latref, c2); latref_c = ldg(g.latref, c2 + 1); } }
        rk4_step<OLD_ORDER, FAST, K>(g, p, bg, e, d, tickets, err, bail, pk, latref_s, latref_c);
      }
      else verlet_step<OLD_ORDER, FAST, K>(g, p, bg, e, d, tickets, err, bail, pk);
      if constexpr (PARK) {
        KID_PHASE_FENCE();
        const lds_double *row = seg.val + (int)__lane_id();
        t.M = row[0 * KID_ROW]; t.T = row[1 * KID_ROW]; t.W = row[2 * KID_ROW]; t.L = row[3 * KID_ROW];
        park_ms = row[4 * KID_ROW];
        if constexpr (!LATE_PARK) { park_bits = row[5 * KID_ROW]; park_hd = row[6 * KID_ROW]; }
        if (Fl<K>::has_fl(fl)) { park_flk = row[7 * KID_ROW]; park_flbits = row[8 * KID_ROW]; if constexpr (!LATE_PARK) park_flbergy = row[9 * KID_ROW]; }
      }
      if (FAST && bail) {  // hand this berg to the general build; nothing of it has been written yet
        const int slot = atomicAdd(redo.count, 1);
        redo.list[slot] = (int)kk;
        if (redo.lane) redo.lane[kk] = redo.step + 1;
        skipped = true; tickets = 0u; err = 0;
      } else {
        // a berg whose cell leaves the computational domain is packed-and-deleted by send_bergs_to_other_pes on a
        // PE without that neighbour (FW:3024-3041)
        if (d.ine < g.isc || d.ine > g.iec || d.jne < g.jsc || d.jne > g.jec) {
          bool back = false;
          if constexpr (!FAST) {  // (a berg of the hot build never changes its cell)
            // periodic_reentry: the zonal seam treated as a boundary between two PEs: sent east/west (FW:3024-3041),
            // unpacked on the other side with *_old reset (FW:3573-3577), the cell one period away accepted by the
            // modulo-aware point-in-cell test (check_and_find_cell FW:3628), xi / yj recomputed (FW:3634), lon unchanged
            const int nic = g.iec - g.isc + 1;
            const int i2 = d.ine > g.iec ? d.ine - nic : (d.ine < g.isc ? d.ine + nic : d.ine);
            if (p.periodic_reentry && g.Lx > 0. && d.jne >= g.jsc && d.jne <= g.jec && i2 >= g.isc && i2 <= g.iec) {
              const GlbCell cell2{g, g.idx(i2, d.jne)};
              int perr = 0; bool pbail = false;
              double xi2, yj2;
              if (!pos_within_cell<false>(g, p, cell2, d.lon, d.lat, i2, d.jne, xi2, yj2, perr, pbail)) err = 1;  // not in the cell one period away: 'can not find a cell to place berg in!' FW:3660
              d.ine = i2; d.xi = xi2; d.yj = yj2;
              stg(b.f[KID_B_UVEL_OLD], kk, d.uvel); stg(b.f[KID_B_VVEL_OLD], kk, d.vvel); stg(b.f[KID_B_LON_OLD], kk, d.lon); stg(b.f[KID_B_LAT_OLD], kk, d.lat);
              back = true;
            }
          }
          if (!back) t.alive = false;
        }
        stg(b.f[KID_B_LON], kk, d.lon); stg(b.f[KID_B_LAT], kk, d.lat); stg(b.f[KID_B_UVEL], kk, d.uvel); stg(b.f[KID_B_VVEL], kk, d.vvel);
        stg(b.f[KID_B_AXN], kk, d.axn); stg(b.f[KID_B_AYN], kk, d.ayn); stg(b.f[KID_B_BXN], kk, d.bxn); stg(b.f[KID_B_BYN], kk, d.byn);
        stg(b.f[KID_B_XI], kk, d.xi); stg(b.f[KID_B_YJ], kk, d.yj);
        stg(b.i[KID_BI_INE], kk, d.ine); stg(b.i[KID_BI_JNE], kk, d.jne);
        if (!RK) { stg(b.f[KID_B_UVEL_PREV], kk, d.uvel_prev); stg(b.f[KID_B_VVEL_PREV], kk, d.vvel_prev); }
      }
    }
    const unsigned long long bt = __ballot(tickets != 0u);
    if (bt) {  // rare
      double ts = wave_sum((double)tickets);
      if (__lane_id() == 0) unsafeAtomicAdd(scal + KID_S_NSPEEDING_TICKETS, ts);
    }
  }
  KID_PHASE_FENCE();
  KID_MARK("evolve_done"); KID_TICK(6);

  if (PH & PH_FL) {  // footloose_calving (IB:5453, 2503-2734) on the berg's own rows between its evolve and its thermodynamics:
    // per berg the reference's order is evolve -> footloose -> thermodynamics too, and nothing of another berg is read.
    // Children are appended behind the population and get their thermodynamics + spreading from a second launch over the
    // new rows (kid_step_local).  The hot build of the fused step keeps the berg in registers across the phase (PARK: its
    // mass_scaling, fl_k and bits were fetched with everything else at the top); elsewhere the state goes through the row.
    if (was_alive && !skipped && t.alive) {
      const FlChildCtx cx{redo.fl_cursor, redo.fl_counter, n, redo.fl_capacity, redo.fl_iNg, redo.fl_step};
      if constexpr (PARK) {
        bool touched = false;
        footloose_core(g, p, b, cx, kk, d.ine, d.jne, CellOf<FAST>::make(g, pk, d.ine, d.jne).area(), park_ms, t.static_berg, t.M, t.T, t.W, t.L, park_flk, park_flbits, touched, acc, ncell, scal);
        if (touched) {
          park_flbits = ldg(b.f[KID_B_MASS_OF_FL_BITS], kk); park_flbergy = ldg(b.f[KID_B_MASS_OF_FL_BERGY_BITS], kk);
          if constexpr (LATE_PARK) seg.val[(int)__lane_id() + 9 * KID_ROW] = park_flbergy;
        }
      } else {
        footloose_one(g, p, b, cx, kk, acc, ncell, scal);
        t.M = ldg(b.f[KID_B_MASS], kk); t.T = ldg(b.f[KID_B_THICKNESS], kk); t.W = ldg(b.f[KID_B_WIDTH], kk); t.L = ldg(b.f[KID_B_LENGTH], kk);
      }
    }
  }

  if (PH & (PH_THERMO | PH_SPREAD)) {
    const bool active = t.alive && !skipped;
    if (!FAST) seg = make_runs(active ? g.idx(d.ine, d.jne) : -1, (lds_double *)lds_vals, (lds_int *)lds_ints);  // cells may have changed
    const typename CellOf<FAST>::type cellv = CellOf<FAST>::make(g, pk, d.ine, d.jne);
    // fused hot build of the plain namelist (no static bergs): every berg still active here went through the hot evolve, which
    // bails unless PkCell::hotok (no NaN in the stencils)
    constexpr bool HOTCHECKED = FAST && (PH & PH_EVOLVE) != 0 && K != 0;
    if constexpr (LATE_PARK) {
      const lds_double *row = seg.val + (int)__lane_id();
      park_bits = row[5 * KID_ROW]; park_hd = row[6 * KID_ROW];
      if (Fl<K>::has_fl(fl)) park_flbergy = row[9 * KID_ROW];
    }
    if constexpr (PARK) { t.mass_scaling = park_ms; t.mass_of_bits = park_bits; t.heat_density = park_hd; }
    else {
      t.mass_scaling = ldg(b.f[KID_B_MASS_SCALING], kk);
      t.mass_of_bits = ldg(b.f[KID_B_MASS_OF_BITS], kk);
      t.heat_density = (PH & PH_THERMO) ? ldg(b.f[KID_B_HEAT_DENSITY], kk) : 0.;
    }
    if (PARK && Fl<K>::has_fl(fl)) { t.mass_of_fl_bits = park_flbits; t.mass_of_fl_bergy_bits = park_flbergy; t.fl_k = park_flk; }
    else if (Fl<K>::has_fl(fl)) {
      t.mass_of_fl_bits = ldg(b.f[KID_B_MASS_OF_FL_BITS], kk); t.mass_of_fl_bergy_bits = ldg(b.f[KID_B_MASS_OF_FL_BERGY_BITS], kk);
      t.fl_k = ldg(b.f[KID_B_FL_K], kk);
    } else { t.mass_of_fl_bits = 0.; t.mass_of_fl_bergy_bits = 0.; t.fl_k = 0.; }
    t.start_mass = (Sw<K>::diag_mask(p) & KID_DIAG_MELT_BY_CLASS) ? ldg(b.f[KID_B_START_MASS], kk) : 0.;
    t.start_year = 0; t.start_day = 0.;
    if (PH & PH_THERMO) {
      if (!OLD_ORDER && (PH & PH_EVOLVE) && (PH & PH_INTERP)) {  // fused step: interp_gridded_fields_to_bergs again at the new position, IB:5473
        if (active) { interp_flds<K, HOTCHECKED>(p, cellv, d.xi, d.yj, e); env_dirty = true; }
      }
      if (OLD_ORDER || (!Sw<K>::mts(p) && !Sw<K>::dem(p) && halo)) {  // IB:2890-2894 (od is not passed there)
        const double od_keep = e.od;
        if (active) { interp_flds<K, HOTCHECKED>(p, cellv, d.xi, d.yj, e); env_dirty = true; }
        if (PH & PH_INTERP) e.od = od_keep;
      }
      // the environment the berg carries from here on is final (the thermodynamics only reads it): stored now, not at the end
      // of the kernel -- thirteen values less to hold across the thermodynamics and the spreading
      if (env_dirty && Fl<K>::store_env(fl) && was_alive && !skipped) {
        stg(b.f[KID_B_UO], kk, e.uo); stg(b.f[KID_B_VO], kk, e.vo); stg(b.f[KID_B_UI], kk, e.ui); stg(b.f[KID_B_VI], kk, e.vi);
        stg(b.f[KID_B_UA], kk, e.ua); stg(b.f[KID_B_VA], kk, e.va); stg(b.f[KID_B_SSH_X], kk, e.ssh_x); stg(b.f[KID_B_SSH_Y], kk, e.ssh_y);
        stg(b.f[KID_B_SST], kk, e.sst); stg(b.f[KID_B_SSS], kk, e.sss); stg(b.f[KID_B_CN], kk, e.cn); stg(b.f[KID_B_HI], kk, e.hi);
        if (PH & PH_INTERP) stg(b.f[KID_B_OD], kk, e.od);
        env_dirty = false;
      }
      KID_PHASE_FENCE();
      KID_MARK("thermo_interp_done"); KID_TICK(7);
      const BergThermo before = t;
      if constexpr ((PH & PH_TSPREAD) != 0) {  // thermodynamics spreads the would-be masses itself, IB:3219-3238
        const TSpreadArgs ts{d.xi, d.yj, b.orient ? b.orient[kk] : p.initial_orientation, Fl<K>::footprint(fl) != 0};
        thermodynamics<true, K>(g, p, cellv, t, e, d.uvel, d.vvel, d.lat, d.ine, d.jne, active, acc, ncell, seg, scal, &ts);
      } else thermodynamics<false, K>(g, p, cellv, t, e, d.uvel, d.vvel, d.lat, d.ine, d.jne, active, acc, ncell, seg, scal);
      if (active) {
        stg(b.f[KID_B_MASS], kk, t.M); stg(b.f[KID_B_THICKNESS], kk, t.T); stg(b.f[KID_B_WIDTH], kk, t.W); stg(b.f[KID_B_LENGTH], kk, t.L);
        if (t.mass_of_bits != before.mass_of_bits) stg(b.f[KID_B_MASS_OF_BITS], kk, t.mass_of_bits);
        if (Fl<K>::has_fl(fl)) {
          stg(b.f[KID_B_MASS_OF_FL_BITS], kk, t.mass_of_fl_bits); stg(b.f[KID_B_MASS_OF_FL_BERGY_BITS], kk, t.mass_of_fl_bergy_bits);
          stg(b.f[KID_B_FL_K], kk, t.fl_k);
          if (t.mass_scaling != before.mass_scaling) {  // converted to a footloose child (IB:3272-3289)
            stg(b.f[KID_B_MASS_SCALING], kk, t.mass_scaling);
            stg(b.i[KID_BI_START_YEAR], kk, t.start_year); stg(b.f[KID_B_START_DAY], kk, t.start_day);
          }
        }
      }
      KID_PHASE_FENCE();
      KID_MARK("thermo_done"); KID_TICK(8);
    }
    if (PH & PH_SPREAD) {  // calculate_mass_on_ocean IB:4989-5009 on the post-thermodynamics state
      const bool act2 = t.alive && !skipped;
      if ((Sw<K>::add_weight_to_ocean(p) && !Sw<K>::time_average_weight(p)) || Sw<K>::find_melt_using_spread_mass(p))
        spread_mass<K>(g, p, cellv, t, d.uvel, d.vvel, d.ine, d.jne, d.xi, d.yj, act2, acc, ncell, seg, Fl<K>::footprint(fl) != 0,
                       (K == 0 && b.orient) ? b.orient[kk] : p.initial_orientation);
      if (!Fl<K>::no_diag(fl)) berg_diagnostics<K>(g, p, cellv, t, d.uvel, d.vvel, d.ine, d.jne, act2, acc, ncell, seg);
    }
    if (FAST && KID_PRIO_ON) __builtin_amdgcn_s_setprio(3);   // ... and the wave that is about to retire: its slot goes to a wave that starts fetching
    seg_flush(seg, acc, ncell);
    KID_MARK("spread_done"); KID_TICK(9);
  }

  if (was_alive && !skipped) {
    if (!t.alive) stg(b.i[KID_BI_ALIVE], kk, 0);
    if (env_dirty && Fl<K>::store_env(fl)) {
      stg(b.f[KID_B_UO], kk, e.uo); stg(b.f[KID_B_VO], kk, e.vo); stg(b.f[KID_B_UI], kk, e.ui); stg(b.f[KID_B_VI], kk, e.vi);
      stg(b.f[KID_B_UA], kk, e.ua); stg(b.f[KID_B_VA], kk, e.va); stg(b.f[KID_B_SSH_X], kk, e.ssh_x); stg(b.f[KID_B_SSH_Y], kk, e.ssh_y);
      stg(b.f[KID_B_SST], kk, e.sst); stg(b.f[KID_B_SSS], kk, e.sss); stg(b.f[KID_B_CN], kk, e.cn); stg(b.f[KID_B_HI], kk, e.hi);
      if (PH & PH_INTERP) stg(b.f[KID_B_OD], kk, e.od);
    }
  }
  const unsigned long long be = __ballot(err != 0);
  if (be && __lane_id() == 0) unsafeAtomicAdd(scal + KID_S_ERROR_COUNT, (double)__popcll(be));
  KID_TICK(10); KID_TICK(99);
  }  // grid-stride loop
}

}  // namespace

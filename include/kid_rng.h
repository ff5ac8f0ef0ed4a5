/* kid_rng.h -- the random numbers of the path, as a counter-based generator.
 *
 * The reference draws from FMS's random_numbers_mod (a Mersenne Twister seeded by constructSeed(mpp_pe(), mpp_pe(), time),
 * /root/reference/src/icebergs.F90:2548-2550) one number per footloose calving event, in traversal order (IB:2631, 2664).
 * FMS is not part of the reference tree, a sequential stream has no order on a GPU, and its values depend on the PE
 * layout anyway; what the model needs is "a uniform number in [0,1) per event".  Here that number is a pure function of
 * (seed, parent berg id, footloose step, draw): Philox-4x32 with 10 rounds (Salmon, Moraes, Dror, Shaw: "Parallel random
 * numbers: as easy as 1, 2, 3", SC'11; constants and round function as published), so that the HIP kernel, the CPU oracle
 * and any host that wants to check them produce the same child positions whatever the row order or the decomposition.
 *
 *   counter = (id low word, id high word, step, draw)      key = (seed, 0x4B49445F)
 *   rn      = ((x0 >> 5) * 2^26 + (x1 >> 6)) / 2^53         53 random bits, 0 <= rn < 1
 *   draw    : 0 = the calving block (IB:2631), 1 = the new-berg-from-bits block (IB:2664)
 *   fl_init_child_xy_by_pe (FW:606): one number for the whole run = the generator at id = 0, step = 0, draw = 0.
 *
 * Plain C, included by the CPU oracle and (with KID_RNG_FN = __host__ __device__) by the HIP library.
 */
#ifndef KID_RNG_H
#define KID_RNG_H
#include <stdint.h>

#ifndef KID_RNG_FN
#define KID_RNG_FN static inline
#endif

#define KID_PHILOX_M0 0xD2511F53u
#define KID_PHILOX_M1 0xCD9E8D57u
#define KID_PHILOX_W0 0x9E3779B9u
#define KID_PHILOX_W1 0xBB67AE85u
#define KID_RNG_KEY1 0x4B49445Fu /* "KID_" */

KID_RNG_FN void kid_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)KID_PHILOX_M0 * (uint64_t)c0, p1 = (uint64_t)KID_PHILOX_M1 * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += KID_PHILOX_W0; k1 += KID_PHILOX_W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* uniform in [0,1) for one footloose event */
KID_RNG_FN double kid_fl_uniform(uint32_t seed, int64_t berg_id, uint32_t step, uint32_t draw) {
  const uint64_t id = (uint64_t)berg_id;
  const uint32_t ctr[4] = {(uint32_t)id, (uint32_t)(id >> 32), step, draw}, key[2] = {seed, KID_RNG_KEY1};
  uint32_t x[4];
  kid_philox4x32_10(ctr, key, x);
  return ((double)(x[0] >> 5) * 67108864.0 + (double)(x[1] >> 6)) * (1.0 / 9007199254740992.0);
}

#endif /* KID_RNG_H */

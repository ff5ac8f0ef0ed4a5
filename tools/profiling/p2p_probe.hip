// Does a polled device-coherent load see another workgroup's store on MI355X, and how long does store -> load take?
// Two workgroups (even and odd blockIdx land on different XCDs) play ping-pong on two 8-byte words with bounded spins.
//   variant 0: global_load sc1 / global_store sc1            (what a relaxed agent-scope atomic load/store compiles to)
//   variant 1: global_load sc0 sc1 / global_store sc0 sc1    (system scope)
//   variant 2: sc1 accesses with a buffer_inv sc1 before every poll
//   variant 3: poll with an atomic fetch-or of 0 (read-modify-write executes at the coherence point)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
enum { SPIN_LIMIT = 200000 };
template <int V> __device__ __forceinline__ unsigned long long ld(const unsigned long long *p) {
  unsigned long long v;
  if (V == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (V == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (V == 2) asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else v = __hip_atomic_fetch_or(const_cast<unsigned long long *>(p), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}
template <int V> __device__ __forceinline__ void st(unsigned long long *p, unsigned long long v) {
  if (V == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
// blocks 2i and 2i+1 are partners; words[2*pair] is written by the even block, words[2*pair+1] by the odd one (128 bytes apart)
template <int V> __global__ void pingpong(unsigned long long *words, int rounds, int *fail) {
  if (threadIdx.x != 0) return;
  const int pair = blockIdx.x / 2, odd = blockIdx.x & 1;
  unsigned long long *mine = words + 32 * (2 * pair + odd), *theirs = words + 32 * (2 * pair + (odd ^ 1));
  // first read the partner's word while it is still 0: the stale copy a later poll must not be served from
  (void)ld<V>(theirs);
  for (int r = 1; r <= rounds; ++r) {
    if (!odd) st<V>(mine, (unsigned long long)r);
    int spins = 0;
    while (ld<V>(theirs) < (unsigned long long)r) { if (++spins > SPIN_LIMIT) { atomicAdd(fail, 1); return; } __builtin_amdgcn_s_sleep(1); }
    if (odd) st<V>(mine, (unsigned long long)r);
  }
}
template <int V> static void run(int pairs, unsigned long long *words, int *fail) {
  const int rounds = 2000;
  CK(hipMemset(words, 0, 64 * 1024)); CK(hipMemset(fail, 0, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  void *fn = (void *)pingpong<V>; int r = rounds; void *args[] = {&words, &r, &fail};
  CK(hipEventRecord(e0, 0));
  CK(hipLaunchCooperativeKernel(fn, dim3(2 * pairs), dim3(64), args, 0, 0));
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  int f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
  printf("variant %d pairs %3d: %7.2f us per round trip (two store->load hops)%s\n", V, pairs, 1e3 * ms / rounds, f ? "  FAILED (stale reads)" : "");
}
int main() {
  unsigned long long *words; int *fail;
  CK(hipMalloc(&words, 64 * 1024)); CK(hipMalloc(&fail, 4));
  for (int pairs : {1, 64}) { run<0>(pairs, words, fail); run<1>(pairs, words, fail); run<2>(pairs, words, fail); run<3>(pairs, words, fail); }
  return 0;
}

// How long does one grid barrier cost on MI355X?  A cooperative grid of B workgroups runs N barriers and nothing else;
// variants of the arrival/poll scheme are timed against each other.  Build + run:  tools/profiling/barrier_probe.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
enum { SPIN_LIMIT = 2000000 };

// variant 0: one counter, every workgroup's thread 0 adds and polls it
template <int SLEEP>
__device__ __forceinline__ bool barrier_flat(unsigned *c, unsigned target) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > SPIN_LIMIT) { ok = false; break; }
      __builtin_amdgcn_s_sleep(SLEEP);
    }
  }
  __syncthreads();
  return ok;
}
// variant 1: arrivals are counted in NSUB sub-counters (one cache line each); the last arrival of a sub-counter adds to the
// top counter, which the others poll: NSUB + B/NSUB serialized atomics instead of B, and B pollers on a line nobody adds to
// until the end
template <int SLEEP, int NSUB>
__device__ __forceinline__ bool barrier_tree(unsigned *c, unsigned epoch, unsigned nblocks) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned sub = blockIdx.x % NSUB;
    const unsigned members = nblocks / NSUB + (sub < nblocks % NSUB ? 1u : 0u);
    const unsigned old = __hip_atomic_fetch_add(c + 32 * (1 + sub), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == epoch * members) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    const unsigned nsub_used = nblocks < NSUB ? nblocks : NSUB;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * nsub_used) {
      if (++spins > SPIN_LIMIT) { ok = false; break; }
      __builtin_amdgcn_s_sleep(SLEEP);
    }
  }
  __syncthreads();
  return ok;
}

template <int V>
__global__ void __launch_bounds__(256) probe(unsigned *c, int nbar, double *data, int nload, int *err) {
  double acc = 0.;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  for (int e = 1; e <= nbar; ++e) {
    // optional: nload device-coherent loads + one store per lane per epoch, as the sub-step kernel does
    for (int q = 0; q < nload; ++q) acc += __hip_atomic_load(data + ((t * 7 + q * 1031 + e) & 0xfffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nload) __hip_atomic_store(data + (t & 0xfffff), acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool ok;
    if (V == 0) ok = barrier_flat<1>(c, (unsigned)e * gridDim.x);
    else if (V == 1) ok = barrier_flat<16>(c, (unsigned)e * gridDim.x);
    else if (V == 2) ok = barrier_tree<1, 8>(c, (unsigned)e, gridDim.x);
    else if (V == 3) ok = barrier_tree<4, 32>(c, (unsigned)e, gridDim.x);
    else ok = barrier_tree<16, 32>(c, (unsigned)e, gridDim.x);
    if (__syncthreads_or(ok ? 0 : 1)) { if (threadIdx.x == 0) *err = 1; return; }
  }
  if (acc == 12345.678) data[0] = acc;
}

template <int V>
static void run(int blocks, int nbar, int nload, unsigned *c, double *data, int *err) {
  void *fn = (void *)probe<V>;
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemset(c, 0, 4096 * 4)); CK(hipMemset(err, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    void *args[] = {&c, &nbar, &data, &nload, &err};
    CK(hipEventRecord(e0, 0));
    CK(hipLaunchCooperativeKernel(fn, dim3(blocks), dim3(256), args, 0, 0));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  }
  int herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
  printf("variant %d blocks %4d loads %2d: %8.2f us per barrier%s\n", V, blocks, nload, 1e3 * best / nbar, herr ? "  TIMED OUT" : "");
}

int main() {
  unsigned *c; double *data; int *err;
  CK(hipMalloc(&c, 4096 * 4)); CK(hipMalloc(&data, (1 << 20) * 8)); CK(hipMalloc(&err, 4));
  CK(hipMemset(data, 0, (1 << 20) * 8));
  const int nbar = 200;
  for (int blocks : {64, 256, 512, 784, 1024}) {
    for (int nload : {0, 12}) {
      run<0>(blocks, nbar, nload, c, data, err); run<1>(blocks, nbar, nload, c, data, err); run<2>(blocks, nbar, nload, c, data, err);
      run<3>(blocks, nbar, nload, c, data, err); run<4>(blocks, nbar, nload, c, data, err);
    }
  }
  return 0;
}

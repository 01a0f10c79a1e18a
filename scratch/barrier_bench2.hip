// grid barrier variants: relaxed agent-scope atomics, data exchanged with agent-scope (sc1) stores / loads
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned* xctr, unsigned gen, int* abort_flag) {
  // MODE 0: one counter, relaxed.  MODE 1: per-XCD counter (blockIdx & 7) then one global counter of 8 arrivals
  __builtin_amdgcn_s_waitcnt(0);  // all of this wave's stores acknowledged
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    const unsigned nb = gridDim.x;
    int spins = 0, good = 1;
    if (MODE == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * nb) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { good = 0; *abort_flag = 1; break; }
      }
    } else {
      const unsigned x = blockIdx.x & 7, per = nb / 8;
      const unsigned old = __hip_atomic_fetch_add(&xctr[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == gen * per - 1) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * 8) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { good = 0; *abort_flag = 1; break; }
      }
    }
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned* ctr, unsigned* xctr, int* abort_flag, float* data, int iters) {
  const unsigned nb = gridDim.x;
  float v = 0.f;
  for (int i = 0; i < iters; ++i) {
    if (threadIdx.x < 64)
      __hip_atomic_store(&data[((i & 1) * nb + blockIdx.x) * 64 + threadIdx.x], (float)(i + blockIdx.x + threadIdx.x),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!grid_barrier<MODE>(ctr, xctr, (unsigned)(i + 1), abort_flag)) return;
    if (threadIdx.x < 64) {
      const unsigned ob = (blockIdx.x + 37) % nb;
      const float got = __hip_atomic_load(&data[((i & 1) * nb + ob) * 64 + threadIdx.x], __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_AGENT);
      if (got != (float)(i + ob + threadIdx.x)) atomicAdd(abort_flag + 1, 1);
      v += got;
    }
  }
  if (threadIdx.x == 0 && v == -1.f) data[0] = v;
}

template <int MODE>
void run(int grid, unsigned* ctr, unsigned* xctr, int* ab, float* data) {
  const int iters = 2000;
  CK(hipMemset(ctr, 0, 4)); CK(hipMemset(xctr, 0, 8 * 32 * 4)); CK(hipMemset(ab, 0, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, ctr, xctr, ab, data, iters);
  CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  int h[2]; CK(hipMemcpy(h, ab, 8, hipMemcpyDeviceToHost));
  printf("mode %d grid %d: %.3f us per barrier (abort %d, stale reads %d)\n", MODE, grid, ms * 1e3 / iters, h[0], h[1]);
}

int main() {
  unsigned *ctr, *xctr; int* ab; float* data;
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&xctr, 8 * 32 * 4)); CK(hipMalloc(&ab, 8)); CK(hipMalloc(&data, 2 * 512 * 64 * 4));
  for (int grid : {256, 512}) { run<0>(grid, ctr, xctr, ab, data); run<1>(grid, ctr, xctr, ab, data); }
  return 0;
}

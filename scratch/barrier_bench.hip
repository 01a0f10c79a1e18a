// micro-benchmark: cost of a grid-wide barrier (agent-scope atomics) on MI355X, 1 and 2 workgroups per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, int* abort_flag, bool fence_all) {
  if (fence_all) __threadfence();
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0, good = 1;
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { good = 0; *abort_flag = 1; break; }
    }
    ok = good;
  }
  __syncthreads();
  if (fence_all) __threadfence();
  return ok != 0;
}

__global__ __launch_bounds__(512) void k(unsigned* ctr, int* abort_flag, float* data, int iters, int fence_all) {
  const unsigned nb = gridDim.x;
  float v = 0.f;
  for (int i = 0; i < iters; ++i) {
    // every block writes one value, every block reads its neighbour's value after the barrier
    if (threadIdx.x == 0) data[(i & 1) * nb + blockIdx.x] = (float)(i + blockIdx.x);
    if (!grid_barrier(ctr, (unsigned)(i + 1) * nb, abort_flag, fence_all)) return;
    if (threadIdx.x == 0) {
      const float got = __builtin_nontemporal_load(&data[(i & 1) * nb + (blockIdx.x + 37) % nb]);
      if (got != (float)(i + (blockIdx.x + 37) % nb)) atomicAdd(abort_flag + 1, 1);
      v += got;
    }
  }
  if (threadIdx.x == 0 && v == -1.f) data[0] = v;
}

int main() {
  unsigned* ctr; int* ab; float* data;
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&ab, 8)); CK(hipMalloc(&data, 2 * 4096 * 4));
  for (int fence_all = 0; fence_all < 2; ++fence_all)
    for (int grid : {256, 512}) {
      const int iters = 2000;
      CK(hipMemset(ctr, 0, 4)); CK(hipMemset(ab, 0, 8));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, ctr, ab, data, 10, fence_all);
      CK(hipMemset(ctr, 0, 4));
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, ctr, ab, data, iters, fence_all);
      CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      int h[2]; CK(hipMemcpy(h, ab, 8, hipMemcpyDeviceToHost));
      printf("grid %d fence_all %d: %.3f us per barrier (abort %d, stale reads %d)\n", grid, fence_all, ms * 1e3 / iters, h[0], h[1]);
    }
  return 0;
}

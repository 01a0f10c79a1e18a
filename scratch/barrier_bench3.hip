// persistent-kernel exchange pattern: every block writes a slice of a vector (write-through), grid barrier,
// every block reads the WHOLE vector.  MODE 0: readers use agent-scope (sc1) loads.  MODE 1: acquire fence + plain loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned* xctr, unsigned gen, int* abort_flag) {
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    const unsigned nb = gridDim.x;
    int spins = 0, good = 1;
    const unsigned x = blockIdx.x & 7, per = nb / 8;
    const unsigned old = __hip_atomic_fetch_add(&xctr[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gen * per - 1) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * 8) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { good = 0; *abort_flag = 1; break; }
    }
    if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned* ctr, unsigned* xctr, int* abort_flag, float* data, int NX, int iters) {
  const unsigned nb = gridDim.x;
  float v = 0.f;
  const int per = (NX + nb - 1) / nb;
  for (int i = 0; i < iters; ++i) {
    float* buf = data + (i & 1) * NX;
    for (int j = threadIdx.x; j < per; j += 512) {
      const int idx = blockIdx.x * per + j;
      if (idx < NX) __hip_atomic_store(&buf[idx], (float)((i * 7 + idx) & 1023), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!grid_barrier<MODE>(ctr, xctr, (unsigned)(i + 1), abort_flag)) return;
    int bad = 0;
    for (int j = threadIdx.x; j < NX; j += 512) {
      float got;
      if (MODE == 0) got = __hip_atomic_load(&buf[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else got = buf[j];
      bad += got != (float)((i * 7 + j) & 1023);
      v += got;
    }
    if (bad) atomicAdd(abort_flag + 1, bad);
  }
  if (threadIdx.x == 0 && v == -1.f) data[0] = v;
}

template <int MODE>
void run(int grid, int NX, unsigned* ctr, unsigned* xctr, int* ab, float* data) {
  const int iters = 2000;
  CK(hipMemset(ctr, 0, 4)); CK(hipMemset(xctr, 0, 8 * 32 * 4)); CK(hipMemset(ab, 0, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, ctr, xctr, ab, data, NX, iters);
  CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  int h[2]; CK(hipMemcpy(h, ab, 8, hipMemcpyDeviceToHost));
  printf("mode %d grid %d NX %d: %.3f us per write+barrier+read (abort %d, stale reads %d)\n", MODE, grid, NX, ms * 1e3 / iters, h[0], h[1]);
}

int main() {
  unsigned *ctr, *xctr; int* ab; float* data;
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&xctr, 8 * 32 * 4)); CK(hipMalloc(&ab, 8)); CK(hipMalloc(&data, 2 * 16384 * 4));
  for (int grid : {256, 512})
    for (int NX : {4096, 11008}) { run<0>(grid, NX, ctr, xctr, ab, data); run<1>(grid, NX, ctr, xctr, ab, data); }
  return 0;
}

// Lab: do two DEPENDENT weight-streaming GEMV launches overlap when they alternate between two streams and hand
// their activation vector over through a device-side counter (sc1 stores / sc1 loads, no kernel boundary)?
//
//   chain per "layer": qkv (12288 x 4096) -> o (4096 x 4096) -> gate/up (22016 x 4096) -> down (4096 x 11008),
//   x_{k+1} = first K_{k+1} outputs of launch k (f32), bf16 weights, 32 layers = 128 dependent launches.
//
//   mode 0: one stream, 512 workgroups, x first, plain stores          (what the product does)
//   mode 1: one stream, 256 workgroups                                 (grid of the hand-off form, boundaries kept)
//   mode 2: two streams alternating, 256 workgroups, weight prefetch -> poll the predecessor's counter -> sc1 x loads;
//           sc1 y stores -> vmcnt(0) -> LDS arrival count -> one agent-scope add per workgroup
//   mode 3: mode 2 captured into ONE hipGraph (fork / join) and replayed
//
// Safety: every poll loop is bounded by the wall clock (20 ms) and sets an error word; at most two launches of the chain
// are resident at once (in-stream order) and both fit on the chip together (256 + 256 workgroups, 2 per CU).
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/lab/handoff_lab.hip -o scratch/bin/handoff_lab
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

typedef __attribute__((ext_vector_type(4))) uint32_t uint4v;
typedef __attribute__((ext_vector_type(4))) float float4v;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

struct Args {
  const uint16_t* W;  // [N][ldw] bf16
  const float* x;     // [K]
  float* y;           // [N]
  int N, K, ldw;
  float oscale;
  // hand-off
  const unsigned* wait_cnt;  // predecessor's arrival counter (nullptr: none)
  unsigned wait_val;
  unsigned done_tag;
  unsigned* done_cnt;  // this launch's arrival counter
  unsigned* err;
  unsigned long long* stamp;  // [grid][4]: begin, poll done, x staged, end (nullptr: off)
  int sleep;
  int sentinel;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float4v load16_sc1(const float* p) {
  float4v v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void store4_sc1(float* p, float v) {
  asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

template <int HAND, int XV, int NW = 2>  // HAND: 0 plain, 1 counter hand-off, 2 tagged 8-byte granules {f32, tag}; NW: weight chunk buffers (NW - 1 chunks in flight)  // XV: float4 of x per thread (K <= 2048 * XV)
__global__ __launch_bounds__(512) void gemv_lab(Args a) {
  constexpr int UNR = 4, R = 2, VN = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* xs = reinterpret_cast<uint16_t*>(smem);
  __shared__ unsigned arrive;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K, ldw = a.ldw;
  const int nwaves = gridDim.x * 8, gw = blockIdx.x * 8 + wave;
  const int ngroups = (a.N + R - 1) / R;
  constexpr int CH = 64 * VN * UNR;
  const int nch = (K + CH - 1) / CH;
  const int my_groups = gw < ngroups ? (ngroups - gw + nwaves - 1) / nwaves : 0;
  const int items = my_groups * nch;
  uint4v wb[NW][UNR][R];
  auto& wcur = wb[0];
  auto load_item = [&](int t, uint4v(&w)[UNR][R]) {
    const int g = gw + (t / nch) * nwaves, c = t % nch;
    const int n0 = g * R;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int k = c * CH + u * 64 * VN + lane * VN;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int n = n0 + r < a.N ? n0 + r : a.N - 1;
        w[u][r] = k < K ? __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(a.W + (int64_t)n * ldw + k))
                        : uint4v{0, 0, 0, 0};
      }
    }
  };
  float4v xv[XV];
  if (tid == 0) arrive = 0;
  unsigned long long st0 = 0, st1 = 0, st2 = 0;
  if (a.stamp && tid == 0) st0 = wall_clock64();
  if constexpr (HAND == 0) {
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int k = (tid + i * 512) * 4;
      xv[i] = k < K ? *reinterpret_cast<const float4v*>(a.x + k) : float4v{0.f, 0.f, 0.f, 0.f};
    }
    if (items > 0) load_item(0, wcur);
    if (NW == 3 && items > 1) load_item(1, wb[1]);
  } else if constexpr (HAND == 2) {
    if (items > 0) load_item(0, wcur);
    if (NW == 3 && items > 1) load_item(1, wb[1]);
    // every thread polls its own granules: {value, tag} pairs, 16 bytes = 2 granules per load; no counter, no hot line
    const uint4v* xg = reinterpret_cast<const uint4v*>(a.x);
    const unsigned tag = a.wait_val;
    unsigned long long t_lim = wall_clock64() + 2000000ull;
    if (a.sentinel) {
      // ONE lane per workgroup waits (politely) for a granule pair near the end of the producer's work list; only then
      // does everybody sweep -- 512 threads re-reading 32 - 88 KB per workgroup all through the producer's run is
      // terabytes per second of L2 traffic
      if (tid == 0) {
        const int ks = K - 4 - 4 * (int)(blockIdx.x & 31);
        for (;;) {
          uint4v g0;
          asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g0) : "v"(xg + (ks >> 1)) : "memory");
          if (g0[1] == tag && g0[3] == tag) break;
          if (wall_clock64() > t_lim) break;
          for (int z = 0; z < a.sleep; ++z) __builtin_amdgcn_s_sleep(8);
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int k = (tid + i * 512) * 4;
      xv[i] = float4v{0.f, 0.f, 0.f, 0.f};
      if (k < K) {
        for (;;) {
          uint4v g0, g1;
          asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(g0), "=&v"(g1) : "v"(xg + (k >> 1)) : "memory");
          if (g0[1] == tag && g0[3] == tag && g1[1] == tag && g1[3] == tag) {
            const uint32_t u0 = g0[0], u1 = g0[2], u2 = g1[0], u3 = g1[2];  // (bit_cast of a vector element: compiler hazard 1)
            xv[i] = float4v{__builtin_bit_cast(float, u0), __builtin_bit_cast(float, u1), __builtin_bit_cast(float, u2),
                            __builtin_bit_cast(float, u3)};
            break;
          }
          if (wall_clock64() > t_lim) {
            atomicAdd(a.err, 1u);
            break;
          }
          for (int z = 0; z < (a.sentinel ? 1 : a.sleep); ++z) __builtin_amdgcn_s_sleep(8);
        }
      }
    }
    if (a.stamp && tid == 0) st1 = wall_clock64();
  } else {
    // weights first: they do not depend on the predecessor
    if (items > 0) load_item(0, wcur);
    if (NW == 3 && items > 1) load_item(1, wb[1]);
    if (a.wait_cnt) {
      if (tid == 0) {
        unsigned long long t_lim = wall_clock64() + 2000000ull;  // 100 MHz: 20 ms
        for (;;) {
          const unsigned v = __hip_atomic_load(a.wait_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (v >= a.wait_val) break;
          if (wall_clock64() > t_lim) {
            atomicAdd(a.err, 1u);
            break;
          }
          for (int z = 0; z < a.sleep; ++z) __builtin_amdgcn_s_sleep(16);
        }
      }
      __syncthreads();
    }
    if (a.stamp && tid == 0) st1 = wall_clock64();
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int k = (tid + i * 512) * 4;
      xv[i] = k < K ? load16_sc1(a.x + k) : float4v{0.f, 0.f, 0.f, 0.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the asm loads are invisible to the compiler's counter)
  }
#pragma unroll
  for (int i = 0; i < XV; ++i) {
    const int k = (tid + i * 512) * 4;
    if (k < K) {
      __hip_bfloat16 b[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) b[e] = __float2bfloat16(xv[i][e]);
      uint2 pk;
      pk.x = (uint32_t) reinterpret_cast<uint16_t&>(b[0]) | ((uint32_t) reinterpret_cast<uint16_t&>(b[1]) << 16);
      pk.y = (uint32_t) reinterpret_cast<uint16_t&>(b[2]) | ((uint32_t) reinterpret_cast<uint16_t&>(b[3]) << 16);
      *reinterpret_cast<uint2*>(&xs[k]) = pk;
    }
  }
  __syncthreads();
  if (a.stamp && tid == 0) st2 = wall_clock64();

  float acc[R] = {0.f, 0.f};
  auto compute = [&](int t, uint4v(&w)[UNR][R]) {
    const int c = t % nch;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int k = c * CH + u * 64 * VN + lane * VN;
      if (k < K) {
        const uint4v xq = *reinterpret_cast<const uint4v*>(&xs[k]);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[r] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, (uint32_t)w[u][r][j]),
                                                     __builtin_bit_cast(bf16x2, (uint32_t)xq[j]), acc[r], false);
      }
    }
    if (c == nch - 1) {
      const int n0 = (gw + (t / nch) * nwaves) * R;
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
      if (lane < R && n0 + lane < a.N) {
        const float v = (lane == 0 ? acc[0] : acc[1]) * a.oscale;
        if constexpr (HAND == 2) {
          uint2 g;
          g.x = __builtin_bit_cast(unsigned, v);
          g.y = a.done_tag;
          asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(reinterpret_cast<uint2*>(a.y) + n0 + lane), "v"(g) : "memory");
        } else if constexpr (HAND == 1) store4_sc1(a.y + n0 + lane, v);
        else a.y[n0 + lane] = v;
      }
      acc[0] = acc[1] = 0.f;
    }
  };
  if constexpr (NW == 2) {
    for (int t = 0; t < items; ++t) {
      if (t + 1 < items) load_item(t + 1, wb[1]);
      compute(t, wb[0]);
      if (t + 1 < items) {
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
          for (int r = 0; r < R; ++r) wb[0][u][r] = wb[1][u][r];
      }
    }
  } else {
    for (int t = 0; t < items; t += 3) {
      if (t + 2 < items) load_item(t + 2, wb[2]);
      compute(t, wb[0]);
      if (t + 1 < items) {
        if (t + 3 < items) load_item(t + 3, wb[0]);
        compute(t + 1, wb[1]);
      }
      if (t + 2 < items) {
        if (t + 4 < items) load_item(t + 4, wb[1]);
        compute(t + 2, wb[2]);
      }
    }
  }
  if constexpr (HAND == 1) {
    // every storing wave drains its stores, then counts in; the wave whose count comes back last signals for all
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      if (atomicAdd(&arrive, 1u) == 7u) __hip_atomic_fetch_add(a.done_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (a.stamp) {
    __syncthreads();
    if (tid == 0) {
      unsigned long long* p = a.stamp + (size_t)blockIdx.x * 8;
      p[0] = st0;
      p[1] = st1;
      p[2] = st2;
      p[3] = wall_clock64();
      p[4] = (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
      p[5] = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    }
  }
}

struct Shape {
  int N, K;
};

int main(int argc, char** argv) {
  const int layers = argc > 1 ? atoi(argv[1]) : 32;
  const int reps = argc > 2 ? atoi(argv[2]) : 12;
  const int sleep_n = argc > 3 ? atoi(argv[3]) : 1;
  const int only_mode = argc > 4 ? atoi(argv[4]) : -1;
  const int deep = argc > 7 ? atoi(argv[7]) : 0;   // 1: three weight chunk buffers (two chunks in flight / prefetched)
  const int tail = argc > 8 ? atoi(argv[8]) : 1;   // 1: x_{k+1} = the LAST K outputs of launch k (produced last)
  const int sentinel = argc > 6 ? atoi(argv[6]) : 0;
  const int lds_pad = argc > 5 ? atoi(argv[5]) : 0;  // KB of dynamic LDS asked for in the hand-off modes (72: exactly two workgroups per CU)
  const Shape shp[4] = {{12288, 4096}, {4096, 4096}, {22016, 4096}, {4096, 11008}};
  const int NSETS = 2;  // 2 x 405 MB of weights > the 256 MB Infinity Cache
  CK(hipSetDevice(0));
  uint16_t* W[NSETS][4];
  for (int s = 0; s < NSETS; ++s)
    for (int i = 0; i < 4; ++i) {
      const size_t n = (size_t)shp[i].N * (shp[i].K + 64);
      CK(hipMalloc(&W[s][i], n * 2));
      std::vector<uint16_t> h(n);
      uint32_t st = 1234567u + 977u * (s * 4 + i);
      for (size_t j = 0; j < n; ++j) {
        st = st * 1664525u + 1013904223u;
        // bf16 in roughly +-0.02 (sign + small exponent + random mantissa)
        const float f = ((int)((st >> 8) & 0xffff) - 32768) * (0.02f / 32768.f);
        uint32_t u;
        memcpy(&u, &f, 4);
        h[j] = (uint16_t)(u >> 16);
      }
      CK(hipMemcpy(W[s][i], h.data(), n * 2, hipMemcpyHostToDevice));
    }
  const int L = getenv("LAB_L") ? atoi(getenv("LAB_L")) : layers * 4;
  // activation buffers: a small ring (reused addresses, as in the product: stale-cache hazards must show)
  const int NBUF = 3;
  float* xb[NBUF];
  for (int i = 0; i < NBUF; ++i) CK(hipMalloc(&xb[i], 22016 * 4));
  float* x0;
  CK(hipMalloc(&x0, 4096 * 4));
  {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = sinf(0.37f * i) * 0.7f;
    CK(hipMemcpy(x0, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  }
  float* xg[NBUF];
  for (int i = 0; i < NBUF; ++i) {
    CK(hipMalloc(&xg[i], 22016 * 8));
    CK(hipMemset(xg[i], 0, 22016 * 8));
  }
  float* x0g;
  CK(hipMalloc(&x0g, 4096 * 8));
  {
    std::vector<uint32_t> h(8192);
    for (int i = 0; i < 4096; ++i) {
      const float f = sinf(0.37f * i) * 0.7f;
      memcpy(&h[2 * i], &f, 4);
      h[2 * i + 1] = 0xffffffffu;
    }
    CK(hipMemcpy(x0g, h.data(), 8192 * 4, hipMemcpyHostToDevice));
  }
  unsigned* cnt;
  const size_t NCNT = (size_t)L * 32 * 64;  // one counter per launch and rep, each on a 128-byte line of its own
  CK(hipMalloc(&cnt, NCNT * 4 + 256));
  unsigned* err = cnt + NCNT;
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, (size_t)L * 512 * 8 * 8));
  CK(hipMemset(stamps, 0, (size_t)L * 512 * 8 * 8));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t ef, ej;
  CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));

  auto set_lds = [&](auto kern) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); };
  set_lds(&gemv_lab<0, 2, 2>);
  set_lds(&gemv_lab<0, 6, 2>);
  set_lds(&gemv_lab<1, 2, 2>);
  set_lds(&gemv_lab<1, 6, 2>);
  set_lds(&gemv_lab<2, 2, 2>);
  set_lds(&gemv_lab<2, 6, 2>);
  set_lds(&gemv_lab<0, 2, 3>);
  set_lds(&gemv_lab<0, 6, 3>);
  set_lds(&gemv_lab<1, 2, 3>);
  set_lds(&gemv_lab<1, 6, 3>);
  set_lds(&gemv_lab<2, 2, 3>);
  set_lds(&gemv_lab<2, 6, 3>);

  std::vector<float> ref(4096), out(4096);
  int rep_no = 0;
  bool stamp_on = false;
  auto enqueue = [&](int mode, int rep, hipStream_t s0, hipStream_t s1) {
    for (int l = 0; l < L; ++l) {
      const int i = l & 3, set = (l >> 2) % NSETS;
      Args a{};
      a.W = W[set][i];
      a.N = shp[i].N;
      a.K = shp[i].K;
      a.ldw = shp[i].K + 64;
      const int xoff = (l == 0 || !tail) ? 0 : shp[(l - 1) & 3].N - shp[i].K;
      a.x = l == 0 ? x0 : xb[(l - 1) % NBUF] + xoff;
      a.y = xb[l % NBUF];
      a.oscale = i == 3 ? 0.6f : 1.2f;
      a.err = err;
      a.sleep = sleep_n;
      a.sentinel = sentinel;
      a.stamp = stamp_on ? stamps + (size_t)l * 512 * 8 : nullptr;
      const bool gran = mode >= 6;
      const bool hand = mode >= 2 && mode != 5 && !gran;
      if (hand) {
        unsigned* base = cnt + ((size_t)(rep % 32) * L) * 64;
        a.done_cnt = base + (size_t)l * 64;
        a.wait_cnt = (l > 0 && mode != 4) ? base + (size_t)(l - 1) * 64 : nullptr;
        a.wait_val = 256;
      }
      if (gran) {
        // tags: unique per (rep, launch); buffers hold 8-byte granules
        a.x = l == 0 ? x0g : xg[(l - 1) % NBUF] + 2 * xoff;
        a.y = xg[l % NBUF];
        a.wait_val = l == 0 ? 0xffffffffu : (unsigned)(rep_no * L + l);  // the predecessor's done_tag
        a.done_tag = (unsigned)(rep_no * L + l + 1);
      }
      const int grid = mode == 0 ? 512 : 256;
      hipStream_t s = (mode == 2 || (mode >= 4 && mode != 7)) ? ((l & 1) ? s1 : s0) : s0;
      const size_t lds = std::max((size_t)a.K * 2, mode >= 1 ? (size_t)lds_pad * 1024 : (size_t)0);
      auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a); };
      const int hm = gran ? 2 : hand ? 1 : 0;
      const bool k4 = a.K <= 4096;
      if (!deep) {
        if (hm == 2) k4 ? go(gemv_lab<2, 2, 2>) : go(gemv_lab<2, 6, 2>);
        else if (hm == 1) k4 ? go(gemv_lab<1, 2, 2>) : go(gemv_lab<1, 6, 2>);
        else k4 ? go(gemv_lab<0, 2, 2>) : go(gemv_lab<0, 6, 2>);
      } else {
        if (hm == 2) k4 ? go(gemv_lab<2, 2, 3>) : go(gemv_lab<2, 6, 3>);
        else if (hm == 1) k4 ? go(gemv_lab<1, 2, 3>) : go(gemv_lab<1, 6, 3>);
        else k4 ? go(gemv_lab<0, 2, 3>) : go(gemv_lab<0, 6, 3>);
      }
    }
  };
  double bytes = 0;
  for (int i = 0; i < 4; ++i) bytes += (double)shp[i].N * shp[i].K * 2;
  bytes *= layers;
  const int NOUT = shp[(L - 1) & 3].N >= 4096 ? 4096 : shp[(L - 1) & 3].N;
  (void)NOUT;

  for (int mode = 0; mode < 8; ++mode) {
    if (only_mode >= 0 && mode != only_mode) continue;
    std::vector<double> ts;
    hipGraphExec_t gexec = nullptr;
    int mismatches = 0;
    for (int rep = 0; rep < reps; ++rep, ++rep_no) {
      if (mode >= 2 && rep % 32 == 0) CK(hipMemset(cnt, 0, NCNT * 4 + 256));
      CK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      if (mode < 2) {
        enqueue(mode, rep, sa, sa);
      } else if (mode == 3 || mode == 7) {
        enqueue(mode, rep, sa, sa);
      } else if (mode == 2 || mode >= 4) {
        CK(hipEventRecord(ef, sa));
        CK(hipStreamWaitEvent(sb, ef, 0));
        enqueue(mode, rep, sa, sb);
        CK(hipEventRecord(ej, sb));
        CK(hipStreamWaitEvent(sa, ej, 0));
      } else {
        // graph: the counters of ONE rep slot are baked in, so they are cleared by a memset node in front
        if (!gexec) {
          hipGraph_t g;
          CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
          CK(hipMemsetAsync(cnt, 0, (size_t)L * 64 * 4, sa));
          CK(hipEventRecord(ef, sa));
          CK(hipStreamWaitEvent(sb, ef, 0));
          enqueue(mode, 0, sa, sb);
          CK(hipEventRecord(ej, sb));
          CK(hipStreamWaitEvent(sa, ej, 0));
          CK(hipStreamEndCapture(sa, &g));
          CK(hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0));
          CK(hipGraphDestroy(g));
        }
        CK(hipGraphLaunch(gexec, sa));
      }
      CK(hipDeviceSynchronize());
      const auto t1 = std::chrono::steady_clock::now();
      ts.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
      if (mode >= 6) {
        std::vector<uint32_t> hg(8192);
        CK(hipMemcpy(hg.data(), xg[(L - 1) % NBUF], 8192 * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < 4096; ++i) memcpy(&out[i], &hg[2 * i], 4);
      } else
        CK(hipMemcpy(out.data(), xb[(L - 1) % NBUF], 4096 * 4, hipMemcpyDeviceToHost));
      if (mode == 0 && rep == 0) ref = out;
      else if (memcmp(ref.data(), out.data(), 4096 * 4) != 0) ++mismatches;
    }
    unsigned herr = 0;
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    std::sort(ts.begin(), ts.end());
    const double med = ts[ts.size() / 2], mn = ts[0];
    printf("mode %d: median %.1f us  min %.1f us  per layer %.2f us  %.2f TB/s (min)  mismatching reps %d  poll timeouts %u  y[0..2] %g %g %g\n",
           mode, med, mn, mn / layers, bytes / mn * 1e-6, mismatches, herr, out[0], out[1], out[2]);
    fflush(stdout);
    if (herr) {
      printf("poll timeouts: stopping\n");
      return 2;
    }
    {
      stamp_on = true;
      if (mode >= 2) CK(hipMemset(cnt, 0, NCNT * 4));
      CK(hipDeviceSynchronize());
      if (mode < 2 || mode == 3 || mode == 7) enqueue(mode, 0, sa, sa);
      else {
        CK(hipEventRecord(ef, sa));
        CK(hipStreamWaitEvent(sb, ef, 0));
        enqueue(mode, 0, sa, sb);
        CK(hipEventRecord(ej, sb));
        CK(hipStreamWaitEvent(sa, ej, 0));
      }
      CK(hipDeviceSynchronize());
      stamp_on = false;
      ++rep_no;
      std::vector<unsigned long long> hs((size_t)L * 512 * 8);
      CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
      const int grid = mode == 0 ? 512 : 256;
      unsigned long long prev_end = 0, t00 = 0;
      printf("  launch  shape   first_begin last_begin  first_polled last_polled  last_staged  first_end  last_end   (us, relative to the previous launch's last_end)\n");
      for (int l = 40; l < 52 && l < L; ++l) {
        unsigned long long b0 = ~0ull, b1 = 0, p0 = ~0ull, p1 = 0, s1 = 0, e0 = ~0ull, e1 = 0;
        for (int w = 0; w < grid; ++w) {
          const unsigned long long* q = &hs[((size_t)l * 512 + w) * 8];
          b0 = std::min(b0, q[0]); b1 = std::max(b1, q[0]);
          p0 = std::min(p0, q[1]); p1 = std::max(p1, q[1]);
          s1 = std::max(s1, q[2]);
          e0 = std::min(e0, q[3]); e1 = std::max(e1, q[3]);
        }
        if (l == 40) { prev_end = b0; t00 = b0; }
        auto us = [&](unsigned long long t) { return ((double)t - (double)prev_end) * 0.01; };
        printf("  %4d  %5dx%-5d  %8.2f %8.2f   %8.2f %8.2f   %8.2f   %8.2f %8.2f   | abs end %.2f\n", l, shp[l & 3].N, shp[l & 3].K, us(b0), us(b1),
               mode >= 2 ? us(p0) : 0.0, mode >= 2 ? us(p1) : 0.0, us(s1), us(e0), us(e1), ((double)e1 - (double)t00) * 0.01);
        prev_end = e1;
      }
      for (int l = 42; l < 44 && l < L; ++l) {
        int per_cu[8 * 64] = {0};
        for (int w = 0; w < grid; ++w) {
          const unsigned long long* q = &hs[((size_t)l * 512 + w) * 8];
          const unsigned hw = (unsigned)q[4], xcc = (unsigned)q[5] & 15;
          const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
          per_cu[(xcc & 7) * 64 + ((se & 3) * 16 + sh * 0 + cu) % 64]++;
          if (w < 4) printf("    wg %d: hw_id %08x xcc %u\n", w, hw, xcc);
        }
        int hist[8] = {0};
        for (int i = 0; i < 8 * 64; ++i) hist[std::min(per_cu[i], 7)]++;
        printf("  launch %d: CU slots holding 0/1/2/3/4+ of its workgroups: %d %d %d %d %d\n", l, hist[0], hist[1], hist[2], hist[3], hist[4] + hist[5] + hist[6] + hist[7]);
      }
    }
  }
  return 0;
}

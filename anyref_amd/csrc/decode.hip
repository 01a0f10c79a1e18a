// Persistent greedy-decode kernel: all transformer layers of one decode step in ONE launch.
//
// Why: a decode step of the 7B LLM is 4 weight-streaming GEMVs + one attention per layer.  As separate
// launches every GEMV pays ~4.5 us of fixed cost (drain, dispatch, the activation round trip, ramp)
// on 10-33 us of streaming -- a quarter of the step.  Here the 512 workgroups stay resident, walk the
// layers phase by phase and meet at a grid barrier between phases; the first two weight chunks of the
// NEXT phase are requested between "arrive" and "wait", so HBM keeps streaming through the barrier
// and through the activation re-staging.
//
// Cross-workgroup data (residual stream, qkv, attention output, MLP activation: a few KB each) is
// exchanged with agent-scope relaxed atomics (sc1 loads / write-through stores, decode_attn.h), so
// the barrier needs no L2 write-back / invalidate: measured 2.8-3.4 us per barrier against 10-35 us
// with release/acquire fences (scratch/barrier_bench*.hip).
//
// The work decomposition (rows per wave, K chunking, reduction order) is exactly gemv_kernel's
// (gemm.hip), so a step computed here is bit-identical to the launch-per-op path; tests compare them.
//
// Reference: the loop this replaces is HF greedy search over LlamaDecoderLayer.forward with a KV
// cache (anyref.py:704-716 -> transformers generate); oracle: anyref_oracle.llama_step.
#include <cstdlib>
#include <type_traits>

#include "decode_attn.h"
#include "kernels.h"

namespace anyref {

namespace {

constexpr int DNT = 512;   // threads per workgroup
constexpr int DUNR = 4;    // 16-byte loads per weight row in flight per lane
constexpr int DRW = 2;     // weight rows streamed per pass (two output rows, or gate row + up row)
constexpr unsigned SPIN_LIMIT = 1u << 22;

// sync words (uint32), one 128-byte line each: [set*SET + 0] global counter, [set*SET + 32*(1+x)]
// counter of workgroup group x (0..7), then the abort flag and the exit counter
constexpr int SYNC_SET = 512, SYNC_ABORT = 1024, SYNC_EXIT = 1056, SYNC_WORDS = 2048;
#ifdef DEC_TRACE
// timing experiment: workgroups 0 and gridDim-1 log the 100 MHz wall clock at phase boundaries
constexpr int TRACE_BASE = 4096, TRACE_PER_BLOCK = 2048;  // uint64 slots after the sync words
#define TR(slot)                                                                                       \
  do {                                                                                                 \
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                          \
      reinterpret_cast<unsigned long long*>(a.sync + TRACE_BASE)[(blockIdx.x ? TRACE_PER_BLOCK : 0) +  \
                                                                 l * 32 + (slot)] = wall_clock64();   \
  } while (0)
#else
#define TR(slot)
#endif

struct GridSync {
  unsigned* w;
  unsigned bar;  // barriers this workgroup has arrived at
};

// Split barrier.  arrive(): every store of this workgroup is acknowledged (write-through), then one
// thread counts the workgroup in -- first on one of 8 group counters, the last of a group on the
// global one (256-512 arrivals on a single address cost 5-10 us, the two-level count 3 us).
// Two counter sets alternate because a workgroup may arrive at barrier k+1 before a slower one has
// arrived at k (never at k+2: it cannot pass wait(k+1) before everyone arrived there).
__device__ __forceinline__ void grid_arrive(GridSync& g) {
  __builtin_amdgcn_s_waitcnt(0);
#ifdef DEC_FENCE
  __threadfence();
#endif
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned set = g.bar & 1, gen = (g.bar >> 1) + 1, per = gridDim.x >> 3;
    unsigned* base = g.w + set * SYNC_SET;
    const unsigned old =
        __hip_atomic_fetch_add(base + 32 * (1 + (blockIdx.x & 7)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gen * per - 1) __hip_atomic_fetch_add(base, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// wait(): false = another workgroup never arrived (spin limit) -> every caller leaves the kernel
__device__ __forceinline__ bool grid_wait(GridSync& g, int* ok_lds) {
  if (threadIdx.x == 0) {
    const unsigned set = g.bar & 1, gen = (g.bar >> 1) + 1;
    unsigned* base = g.w + set * SYNC_SET;
    unsigned spins = 0;
    int ok = 1;
    while (__hip_atomic_load(base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * 8) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_LIMIT || __hip_atomic_load(g.w + SYNC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(g.w + SYNC_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    *ok_lds = ok;
  }
  __syncthreads();
#ifdef DEC_FENCE
  __threadfence();
#endif
  g.bar += 1;
  return *ok_lds != 0;
}

struct Phase {  // one weight-streaming phase: y[b, n] = act(sum_k W[n,k] * norm(x)[b,k]) (+ resid)
  const void* W;
  const void* W2;  // != null: y = silu(W x) * (W2 x)
  const float* x;
  const float* gain;  // != null: RMSNorm(x) * gain fused into the staging
  float* y;
  const float* resid;
  int N, K, ldx, ldy;
  float eps;
  int ldw;  // row stride of W / W2 in elements (2K for the interleaved gate / up matrix)
};

template <typename T>
__device__ __forceinline__ void load_item(const Phase& p, int gw, int nwaves, int t, int lane,
                                          uint4v (&w)[DUNR][DRW]) {
  constexpr int VN = Vec16<T>::N, CH = 64 * VN * DUNR;
  const bool dual = p.W2 != nullptr;
  const int nch = cdiv(p.K, CH);
  const int g = gw + (t / nch) * nwaves, c = t % nch;
  const int n0 = g * (dual ? 1 : 2);
  const int r0 = n0 < p.N ? n0 : p.N - 1;
  const int r1 = dual ? n0 : (n0 + 1 < p.N ? n0 + 1 : p.N - 1);
  // gw is wave-uniform (readfirstlane), so both row bases live in SGPRs; the weight pointers come out
  // of a struct in memory, hence the explicit global address space (a flat load would also tick lgkmcnt)
  typedef const __attribute__((address_space(1))) uint4v* gptr;
  const T* a0 = reinterpret_cast<const T*>(p.W) + (int64_t)r0 * p.ldw;
  const T* a1 = reinterpret_cast<const T*>(dual ? p.W2 : p.W) + (int64_t)r1 * p.ldw;
#pragma unroll
  for (int u = 0; u < DUNR; ++u) {
    const int k = c * CH + u * 64 * VN + lane * VN;
    w[u][0] = k < p.K ? __builtin_nontemporal_load((gptr)(a0 + k)) : uint4v{0, 0, 0, 0};
    w[u][1] = k < p.K ? __builtin_nontemporal_load((gptr)(a1 + k)) : uint4v{0, 0, 0, 0};
  }
}

__device__ __forceinline__ int phase_items(const Phase& p, int vn, int gw, int nwaves) {
  const int ngroups = cdiv(p.N, p.W2 ? 1 : 2);
  const int my_groups = gw < ngroups ? (ngroups - gw + nwaves - 1) / nwaves : 0;
  return my_groups * cdiv(p.K, 64 * vn * DUNR);
}

// request items 0 and 1 of a phase (between grid_arrive and grid_wait of the barrier before it)
template <typename T>
__device__ __forceinline__ void prefetch_phase(const Phase& p, int gw, int nwaves, int lane, uint4v (&wcur)[DUNR][DRW],
                                               uint4v (&wnxt)[DUNR][DRW]) {
  const int items = phase_items(p, Vec16<T>::N, gw, nwaves);
  // both buffers are (re)defined on every path: a stale value kept "just in case" would stay live
  // across the attention / barrier code and cost 64 VGPRs there
  if (items > 0) {
    load_item<T>(p, gw, nwaves, 0, lane, wcur);
  } else {
#pragma unroll
    for (int u = 0; u < DUNR; ++u)
#pragma unroll
      for (int r = 0; r < DRW; ++r) wcur[u][r] = uint4v{0, 0, 0, 0};
  }
  if (items > 1) {
    load_item<T>(p, gw, nwaves, 1, lane, wnxt);
  } else {
#pragma unroll
    for (int u = 0; u < DUNR; ++u)
#pragma unroll
      for (int r = 0; r < DRW; ++r) wnxt[u][r] = uint4v{0, 0, 0, 0};
  }
}

// gemv_kernel's body with items 0/1 already in flight.  GAIN phases (fused RMSNorm) hold the row in
// registers: XPT elements per thread, K <= 512*XPT.
template <typename T, int NB, bool DUAL, bool GAIN, int XPT>
__device__ __forceinline__ void gemv_phase(const Phase& a, int nb, T* xs, float (*red)[8], int gw, int nwaves,
                                           uint4v (&wcur)[DUNR][DRW], uint4v (&wnxt)[DUNR][DRW]) {
  constexpr int VN = Vec16<T>::N, R = DUAL ? 1 : 2, CH = 64 * VN * DUNR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K;
  const int nch = cdiv(K, CH);
  const int items = phase_items(a, VN, gw, nwaves);

  if constexpr (GAIN) {
    // RMSNorm fused into the staging (same arithmetic and summation order as gemv_kernel)
    float xr[NB][XPT], gr[XPT];
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int k = tid + i * DNT;
      gr[i] = k < K ? a.gain[k] : 1.f;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float* x = a.x + (int64_t)(b < nb ? b : 0) * a.ldx;
#pragma unroll
      for (int i = 0; i < XPT; ++i) {
        const int k = tid + i * DNT;
        xr[b][i] = (b < nb && k < K) ? ld_x<true>(&x[k]) : 0.f;
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b >= nb) continue;
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < XPT; ++i) ss += xr[b][i] * xr[b][i];
      ss = wave_sum(ss);
      if (lane == 0) red[b][wave] = ss;
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) tot += red[b][w];
      const float scale = rsqrtf(tot / (float)K + a.eps);
#pragma unroll
      for (int i = 0; i < XPT; ++i) {
        const int k = tid + i * DNT;
        if (k < K) xs[b * K + k] = from_f32<T>(xr[b][i] * scale * gr[i]);
      }
    }
  } else {
    // plain copy, eight elements per thread in flight (the weight prefetch holds 64 VGPRs meanwhile)
    for (int b = 0; b < nb; ++b) {
      const float* x = a.x + (int64_t)b * a.ldx;
      for (int k0 = 0; k0 < K; k0 += DNT * 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = k0 + tid + i * DNT;
          v[i] = k < K ? ld_x<true>(&x[k]) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = k0 + tid + i * DNT;
          if (k < K) xs[b * K + k] = from_f32<T>(v[i]);
        }
      }
    }
  }
  __syncthreads();

  float acc[DRW][NB];
#pragma unroll
  for (int r = 0; r < DRW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
  for (int t = 0; t < items; ++t) {
    if (t > 0 && t + 1 < items) load_item<T>(a, gw, nwaves, t + 1, lane, wnxt);  // t == 0: item 1 is in flight
    const int c = t % nch;
#pragma unroll
    for (int u = 0; u < DUNR; ++u) {
      const int k = c * CH + u * 64 * VN + lane * VN;
      if (k < K) {
        float xf[NB][VN];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (b < nb) {
            const uint4v xv = *reinterpret_cast<const uint4v*>(&xs[b * K + k]);
            Vec16<T>::unpack(xv, xf[b]);
          } else {
#pragma unroll
            for (int i = 0; i < VN; ++i) xf[b][i] = 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < DRW; ++r) {
          float wf[VN];
          Vec16<T>::unpack(wcur[u][r], wf);
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < VN; ++i) acc[r][b] = fmaf(wf[i], xf[b][i], acc[r][b]);
        }
      }
    }
    if (c == nch - 1) {
      const int n0 = (gw + (t / nch) * nwaves) * R;
#pragma unroll
      for (int r = 0; r < DRW; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[r][b] = wave_sum(acc[r][b]);
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int n = n0 + r;
          if (n >= a.N) continue;
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            if (b >= nb) continue;
            float v = acc[r][b];
            if (DUAL) v = apply_act(v, ACT_SILU) * acc[DRW - 1][b];
            const int64_t o = (int64_t)b * a.ldy + n;
            if (a.resid) v += ld_x<true>(&a.resid[o]);
            st_x<true>(&a.y[o], v);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < DRW; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
    }
    if (t + 1 < items) {
#pragma unroll
      for (int u = 0; u < DUNR; ++u)
#pragma unroll
        for (int r = 0; r < DRW; ++r) wcur[u][r] = wnxt[u][r];
    }
  }
}

}  // namespace

// No occupancy bound on purpose: capped at 128 VGPRs (two workgroups per CU) hipcc spills ~26 VGPRs to
// scratch, and a kernel with scratch gave WRONG, run-to-run different tokens on this stack whenever it
// ran from a graph replay or next to another stream (correct only eagerly on an idle device).  At
// ~240 VGPRs there is no scratch and one workgroup per CU.
template <typename T, int NB, int HD, int XH>
__global__ __launch_bounds__(DNT) void decode_layers_kernel(DecodeStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[NB][8];
  __shared__ int ok_lds;
  T* xs = reinterpret_cast<T*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = gridDim.x * 8, gw = blockIdx.x * 8 + wave;
  const int H = a.H, F = a.F, nb = a.B;
  GridSync gs{a.sync, 0u};
  uint4v wcur[DUNR][DRW], wnxt[DUNR][DRW];

  auto qkv_phase = [&](const DecodeLayerPtrs& L) {
    return Phase{L.qkv, nullptr, a.x, L.in_gain, a.qkv, nullptr, 3 * H, H, H, 3 * H, a.eps, L.qkv_ld};
  };
  {
    const Phase p = qkv_phase(a.layers[0]);
    prefetch_phase<T>(p, gw, nwaves, lane, wcur, wnxt);
  }
  for (int l = 0; l < a.nl; ++l) {
    const DecodeLayerPtrs& L = a.layers[l];
    // ---- q,k,v = W_qkv * RMSNorm(x) ----
    TR(0);
    gemv_phase<T, NB, false, true, XH>(qkv_phase(L), nb, xs, red, gw, nwaves, wcur, wnxt);
    TR(1);
    grid_arrive(gs);
    TR(2);
    // ---- RoPE + cache append + attention: one workgroup per (sequence, head); the others go
    //      straight on to request the o-projection weights ----
    if ((int)blockIdx.x < nb * a.nh) {
      if (!grid_wait(gs, &ok_lds)) return;
      for (int bh = blockIdx.x; bh < nb * a.nh; bh += gridDim.x)
        decode_attn_body<T, HD, true>(a.qkv, a.pos, a.rope, reinterpret_cast<T*>(L.kc), reinterpret_cast<T*>(L.vc),
                                      a.maxS, a.nh, a.scale, a.att,
                                      l == a.nl - 1 ? reinterpret_cast<T*>(a.q_keep) : nullptr, bh % a.nh, bh / a.nh,
                                      reinterpret_cast<float*>(smem));
    } else {
      gs.bar += 1;
    }
    TR(3);
    grid_arrive(gs);
    // ---- x += W_o * att ----
    {
      const Phase p{L.o, nullptr, a.att, nullptr, a.x, a.x, H, H, H, H, a.eps, L.o_ld};
      prefetch_phase<T>(p, gw, nwaves, lane, wcur, wnxt);
      TR(4);
      if (!grid_wait(gs, &ok_lds)) return;
      TR(5);
      gemv_phase<T, NB, false, false, XH>(p, nb, xs, red, gw, nwaves, wcur, wnxt);
      TR(6);
    }
    grid_arrive(gs);
    // ---- act = silu(W_gate * RMSNorm(x)) * (W_up * RMSNorm(x)) ----
    {
      const Phase p{L.gate, L.up, a.x, L.post_gain, a.act, nullptr, F, H, H, F, a.eps, L.gu_ld};
      prefetch_phase<T>(p, gw, nwaves, lane, wcur, wnxt);
      TR(7);
      if (!grid_wait(gs, &ok_lds)) return;
      TR(8);
      gemv_phase<T, NB, true, true, XH>(p, nb, xs, red, gw, nwaves, wcur, wnxt);
      TR(9);
    }
    grid_arrive(gs);
    // ---- x += W_down * act ----
    {
      const Phase p{L.down, nullptr, a.act, nullptr, a.x, a.x, H, F, F, H, a.eps, L.down_ld};
      prefetch_phase<T>(p, gw, nwaves, lane, wcur, wnxt);
      TR(10);
      if (!grid_wait(gs, &ok_lds)) return;
      TR(11);
      gemv_phase<T, NB, false, false, XH>(p, nb, xs, red, gw, nwaves, wcur, wnxt);
      TR(12);
    }
    if (l + 1 < a.nl) {
      grid_arrive(gs);
      const Phase p = qkv_phase(a.layers[l + 1]);
      prefetch_phase<T>(p, gw, nwaves, lane, wcur, wnxt);
      TR(13);
      if (!grid_wait(gs, &ok_lds)) return;
      TR(14);
    }
  }
  // last workgroup out rearms the counters for the next launch
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(a.sync + SYNC_EXIT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - 1) {
      for (int s = 0; s < 2; ++s)
        for (int i = 0; i < 9; ++i)
          __hip_atomic_store(a.sync + s * SYNC_SET + 32 * i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.sync + SYNC_EXIT, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

#ifdef DEC_TRACE
size_t decode_sync_bytes() { return TRACE_BASE * 4 + 2 * TRACE_PER_BLOCK * 8; }
#else
size_t decode_sync_bytes() { return SYNC_WORDS * sizeof(unsigned); }
#endif

template <typename T>
bool launch_decode_layers(const DecodeStepArgs& a, hipStream_t s) {
  constexpr int VN = Vec16<T>::N;
  const int hd = a.H / a.nh;
  if (a.B < 1 || a.B > 2 || (hd != 128 && hd != 64) || a.H % VN || a.F % VN || a.H > 512 * 16 || a.maxS > 12000)
    return false;
  const size_t lds_gemv = (size_t)a.B * (a.F > a.H ? a.F : a.H) * sizeof(T);
  const size_t lds_attn = hd == 128 ? decode_attn_lds<T, 128>(a.maxS) : decode_attn_lds<T, 64>(a.maxS);
  const size_t lds = lds_gemv > lds_attn ? lds_gemv : lds_attn;
  if (lds > 150 * 1024) return false;

  auto go = [&](auto nb_tag, auto hd_tag, auto xh_tag) -> bool {
    constexpr int NB = decltype(nb_tag)::value, HD = decltype(hd_tag)::value, XH = decltype(xh_tag)::value;
    auto kern = &decode_layers_kernel<T, NB, HD, XH>;
    static int grid = 0;  // per instantiation; 0 = not probed, -1 = cannot be co-resident
    static size_t grid_lds = 0;
    if (grid == 0 || lds > grid_lds) {
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  150 * 1024));
      int dev = 0, cus = 0, per_cu = 0;
      HIP_TRY(hipGetDevice(&dev));
      HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, DNT, lds));
      // every workgroup must be resident at once (grid barrier): never more than the device holds
      int g = cus * (per_cu < 2 ? per_cu : 2);
      g -= g % 8;
      grid = g >= 8 ? g : -1;
      // an instantiation that spilled to scratch is refused (see the note above the kernel)
      hipFuncAttributes fa;
      HIP_TRY(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
      if (fa.localSizeBytes > 0) grid = -1;
      grid_lds = lds;
    }
    if (grid < 0) return false;
    // algorithmic bytes: every layer weight once
    const double wbytes = (double)a.nl * ((double)a.H * a.H * 4 + (double)a.H * a.F * 3) * sizeof(T);
    ProfScope prof(sizeof(T) == 2 ? "decode_layers_bf16" : "decode_layers_f32", 2.0 * a.B * wbytes / sizeof(T),
                   wbytes, s);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(DNT), lds, s, a);
    HIP_TRY(hipGetLastError());
    return true;
  };
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I8 = std::integral_constant<int, 8>;
  using I16 = std::integral_constant<int, 16>;
  using I64 = std::integral_constant<int, 64>;
  using I128 = std::integral_constant<int, 128>;
  const bool wide = a.H > 512 * 8;
  if (hd == 128) {
    if (!wide) return a.B == 1 ? go(I1(), I128(), I8()) : go(I2(), I128(), I8());
    return a.B == 1 ? go(I1(), I128(), I16()) : go(I2(), I128(), I16());
  }
  if (wide) return false;
  return a.B == 1 ? go(I1(), I64(), I8()) : go(I2(), I64(), I8());
}
template bool launch_decode_layers<float>(const DecodeStepArgs&, hipStream_t);
template bool launch_decode_layers<bf16>(const DecodeStepArgs&, hipStream_t);

}  // namespace anyref

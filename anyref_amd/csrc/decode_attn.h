// Decode-step attention body, shared by decode_attn_kernel (attention.hip) and the persistent
// decode-step kernel (decode.hip).
#pragma once
#include "common.h"

namespace anyref {

// Loads / stores of vectors that another workgroup of the SAME launch produced or will consume:
// agent-scope relaxed atomics (global_load/store ... sc1) go past the per-XCD L2, so no cache
// write-back / invalidate is needed around the grid barrier of the persistent kernel.
template <bool COH>
__device__ __forceinline__ float ld_x(const float* p) {
  if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
// The coherent store is an atomic exchange WITH return: a plain write-through store is acknowledged
// by the issuing XCD's L2 before it is visible at the memory side, so under a saturated fabric the
// barrier arrival (another address, another channel) could overtake it and a reader on another XCD
// would see the old value (observed: wrong tokens whenever another stream or a graph replay changed
// the timing).  The returning atomic completes at the coherence point; s_waitcnt vmcnt(0) in
// grid_arrive then really means "visible".
template <bool COH>
__device__ __forceinline__ void st_x(float* p, float v) {
  if (COH) {
    const float old = __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(old));
  } else {
    *p = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Decode-step attention (one query per sequence), fused with RoPE + KV-cache append.  Latency bound
// (a few MB per launch), so the structure minimises dependent memory round trips: one 512-thread
// workgroup per (head, sequence); LPK = hd/VEC lanes share one key (16-byte loads straight from the
// cache), KPI = 512/LPK keys per sweep, UN sweeps per batch.  K AND V of a batch are requested
// together, the first batch before the query is even rotated, the next batch before the current one is
// used; every lane group keeps its own online-softmax state (m, l, partial O) over its keys, so the
// loop has no workgroup barrier, and the KPI partial states are merged once at the end.
// (The previous form -- scores to LDS, block softmax, second sweep for V -- had 7 dependent round
// trips and 6 barriers: 10.1 us per layer at n = 330; this one 4 round trips.)
// ---------------------------------------------------------------------------------------------
// exp of the softmax: the fast hardware form (v_exp_f32) for the 16-bit build, libm's for the f32 parity build
template <typename T>
__device__ __forceinline__ float dexp(float x) {
  if constexpr (sizeof(T) == 2) return __expf(x);
  else return expf(x);
}

template <typename T, int HD, bool COH>
__device__ __forceinline__ void decode_attn_body(const float* __restrict__ qkv, const int* __restrict__ pos,
                                                 const float* __restrict__ cs_tab, T* __restrict__ kc,
                                                 T* __restrict__ vc, int maxS, int H, float scale,
                                                 float* __restrict__ out, T* __restrict__ q_keep, int h, int b,
                                                 float* sm) {
  // sweeps per batch: f32 rows are twice the bytes and LPK = 32 lanes share a key, so a batch of 4 sweeps is 64 keys
  // (bf16: 128): 6 sweeps take a ~330-key context in four batches (round trips) instead of six (8 sweeps spill)
  constexpr int NT = 512, UN = sizeof(T) == 2 ? 4 : 6;
  constexpr int VEC = Vec16<T>::N, LPK = HD / VEC, KPI = NT / LPK, HALF = HD / 2;
  static_assert(LPK <= 64 && (LPK & (LPK - 1)) == 0, "lanes per key must be a power of two within a wave");
  float* q_s = sm;                 // [HD] rotated, T-rounded, pre-scaled query
  float* k_s = q_s + HD;           // [HD] this step's key (as stored in the cache)
  float* v_s = k_s + HD;           // [HD]
  float* part = v_s + HD;          // [KPI][HD] partial O of every lane group
  float* ms = part + KPI * HD;     // [KPI] its running max
  float* ls = ms + KPI;            // [KPI] its running sum
  const int tid = threadIdx.x;
  const int p = pos[b], n = p + 1;
  const float* row = qkv + (int64_t)b * 3 * H * HD;
  const int64_t cbase = ((int64_t)b * maxS * H + h) * HD;   // + j*H*HD for key j
  const int sub = tid % LPK, slice = tid / LPK;

  uint4v kcur[UN], vcur[UN], knxt[UN], vnxt[UN];
  auto load_batch = [&](int j0, uint4v (&kk)[UN], uint4v (&vv)[UN]) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = j0 + u * KPI + slice;
      const int64_t o = cbase + (int64_t)j * H * HD + sub * VEC;
      kk[u] = j < p ? *reinterpret_cast<const uint4v*>(kc + o) : uint4v{0, 0, 0, 0};
      vv[u] = j < p ? *reinterpret_cast<const uint4v*>(vc + o) : uint4v{0, 0, 0, 0};
    }
  };
  // Order in the (in-order) vector-memory queue: this step's q / k / v row (independent of the position) and the RoPE
  // table row FIRST, the first K/V batch behind them -- the rotation then waits for a few hundred bytes, not for the
  // batch's 64 KB (round 2 issued the batch first and the rotation sat behind it).
  float q1 = 0.f, q2 = 0.f, k1 = 0.f, k2 = 0.f, v1 = 0.f, v2 = 0.f, cs = 0.f, sn = 0.f;
  if (tid < HALF) {
    const int d = tid;
    q1 = ld_x<COH>(&row[h * HD + d]); q2 = ld_x<COH>(&row[h * HD + d + HALF]);
    k1 = ld_x<COH>(&row[(H + h) * HD + d]); k2 = ld_x<COH>(&row[(H + h) * HD + d + HALF]);
    v1 = ld_x<COH>(&row[(2 * H + h) * HD + d]); v2 = ld_x<COH>(&row[(2 * H + h) * HD + d + HALF]);
    cs = cs_tab[((int64_t)p * 2) * HALF + d]; sn = cs_tab[((int64_t)p * 2 + 1) * HALF + d];
  }
  load_batch(0, kcur, vcur);  // in flight while the new token is rotated and appended

  if (tid < HALF) {
    const int d = tid;
    const T qa = from_f32<T>(q1 * cs - q2 * sn), qb = from_f32<T>(q2 * cs + q1 * sn);
    const T ka = from_f32<T>(k1 * cs - k2 * sn), kb = from_f32<T>(k2 * cs + k1 * sn);
    const T va = from_f32<T>(v1);
    const T vb = from_f32<T>(v2);
    const int64_t co = cbase + (int64_t)p * H * HD;
    kc[co + d] = ka; kc[co + d + HALF] = kb;
    vc[co + d] = va; vc[co + d + HALF] = vb;
    if (q_keep) { q_keep[co + d] = qa; q_keep[co + d + HALF] = qb; }
    q_s[d] = to_f32<T>(qa) * scale; q_s[d + HALF] = to_f32<T>(qb) * scale;
    k_s[d] = to_f32<T>(ka); k_s[d + HALF] = to_f32<T>(kb);
    v_s[d] = to_f32<T>(va); v_s[d + HALF] = to_f32<T>(vb);
  }
  __syncthreads();
  float qf[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) qf[i] = q_s[sub * VEC + i];

  float m_run = -INFINITY, l_run = 0.f, acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  for (int j0 = 0; j0 < n; j0 += KPI * UN) {
    const bool more = j0 + KPI * UN < n;
    if (more) load_batch(j0 + KPI * UN, knxt, vnxt);
    float sc[UN], vf[UN][VEC];
    float mx = m_run;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = j0 + u * KPI + slice;
      float kf[VEC];
      Vec16<T>::unpack(kcur[u], kf);
      Vec16<T>::unpack(vcur[u], vf[u]);
      if (j == p) {  // the token being decoded: its K/V were written by this workgroup a moment ago
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          kf[i] = k_s[sub * VEC + i];
          vf[u][i] = v_s[sub * VEC + i];
        }
      }
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) dot = fmaf(qf[i], kf[i], dot);
#pragma unroll
      for (int o = 1; o < LPK; o <<= 1) dot += __shfl_xor(dot, o, 64);
      sc[u] = j < n ? dot : -INFINITY;
      mx = fmaxf(mx, sc[u]);
    }
    if (mx > -INFINITY) {  // this lane group has seen a key (uniform within the group)
      const float alpha = dexp<T>(m_run - mx);  // m_run = -inf -> 0
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] *= alpha;
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const float pj = dexp<T>(sc[u] - mx);  // masked key: exp(-inf) = 0
        l_run += pj;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = fmaf(pj, vf[u][i], acc[i]);
      }
      m_run = mx;
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        kcur[u] = knxt[u];
        vcur[u] = vnxt[u];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) part[slice * HD + sub * VEC + i] = acc[i];
  if (sub == 0) {
    ms[slice] = m_run;
    ls[slice] = l_run;
  }
  __syncthreads();
  // merge the KPI partial states (slice 0 always holds key 0, so M is finite).  The weights exp(m_s - M) are the same
  // for every output column: one lane group computes them side by side (the HD merge threads used to evaluate all KPI
  // of them each, one after the other: ~1.5 us of a 10 us kernel)
  float* wts = ls + KPI;  // [KPI]
  if (tid < KPI) {
    float M = -INFINITY;
#pragma unroll 8
    for (int sI = 0; sI < KPI; ++sI) M = fmaxf(M, ms[sI]);
    wts[tid] = dexp<T>(ms[tid] - M);  // empty group: exp(-inf) = 0
  }
  __syncthreads();
  if (tid < HD) {
    float o = 0.f, l = 0.f;
#pragma unroll 8
    for (int sI = 0; sI < KPI; ++sI) {
      const float w = wts[sI];
      o = fmaf(w, part[sI * HD + tid], o);
      l = fmaf(w, ls[sI], l);
    }
    st_x<COH>(&out[((int64_t)b * H + h) * HD + tid], o / l);
  }
  __syncthreads();  // LDS is reused by the caller (persistent decode kernel)
}

// dynamic LDS (bytes) decode_attn_body needs
template <typename T, int HD>
inline size_t decode_attn_lds(int maxS) {
  constexpr int KPI = 512 / (HD / Vec16<T>::N);
  return sizeof(float) * (size_t)(3 * HD + KPI * HD + 16 + maxS + KPI * 4);
}

}  // namespace anyref



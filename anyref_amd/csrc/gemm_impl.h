// Tiled MFMA GEMM for gfx950:  C = act(alpha * A W^T + bias) (+ resid), W in nn.Linear layout [N,K].
//
// Two arithmetic flavours share one source through Mma<T>:
//   T = bf16  : v_mfma_f32_16x16x32_bf16 (perf mode)
//   T = float : v_mfma_f32_16x16x4_f32   (parity mode: exact-f32 fmaf chain, MI355X_MICROARCH
//               "FP32-input MFMA")
// Fragment maps (cdna_hip_programming.md §3): A[row l&15][k = KL*(l>>4)+j], B[k][col l&15],
// C: col = l&15, row = 4*(l>>4)+reg.
//
// Structure: 256 threads = 4 waves as 2x2, block tile BM x BN x BK, register-prefetched global
// loads (issue tile t+1 before computing tile t, write to LDS after the barrier: T14 split).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <type_traits>
#include <utility>
#include <vector>
#include <mutex>

#pragma once
#include "kernels.h"

namespace anyref {

template <typename T>
struct Mma;
template <>
struct Mma<bf16> {
  static constexpr int KS = 32;  // k per MFMA
  static constexpr int VEC = 8;  // elements per 16 B
  using Frag = short8;
  // p points at tile[row][k0]; lane picks its 8 contiguous k
  static __device__ inline Frag load(const bf16* p, int lane) {
    return *reinterpret_cast<const short8*>(p + 8 * (lane >> 4));
  }
  static __device__ inline float4v mma(Frag a, Frag b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Mma<f16> {
  static constexpr int KS = 32;
  static constexpr int VEC = 8;
  using Frag = short8;
  static __device__ inline Frag load(const f16* p, int lane) {
    return *reinterpret_cast<const short8*>(p + 8 * (lane >> 4));
  }
  static __device__ inline float4v mma(Frag a, Frag b, float4v c) { return mfma_16x16x32<f16>(a, b, c); }
};
template <>
struct Mma<sp16> {  // split-pair activations: the LDS-DMA kernel only (bf16 MFMA); the constants are what the launcher reads
  static constexpr int KS = 32;
  static constexpr int VEC = 8;
};
template <>
struct Mma<float> {
  static constexpr int KS = 4;
  static constexpr int VEC = 4;
  using Frag = float;
  static __device__ inline Frag load(const float* p, int lane) { return p[lane >> 4]; }
  static __device__ inline float4v mma(Frag a, Frag b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};

// Shared epilogue: bias, activation, residual, row map, typed store.
// The main loop issues mfma(W-fragment, A-fragment), i.e. it accumulates the TRANSPOSED 16x16 tile:
// lane l then holds C[m = tile_m + (l & 15)][n = tile_n + 4 * (l >> 4) + r], r = 0..3 -- four
// CONSECUTIVE columns of one output row, stored as one 8-byte (bf16) / 16-byte (f32) vector.  With the
// natural orientation a lane holds four rows of one column and the tile leaves as 2-4-byte scalars:
// 4x the store instructions, and the store tail (issue-bound, cf. guide T21) cost ~30 us of an 84 us
// 4096x3840x1280 GEMM.
// compile-time loop: the callable receives std::integral_constant indices
template <int... Is, typename F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

// ACT and VEC are compile-time: the epilogue is fully unrolled over the wave's MI x NI fragments, and with
// a runtime activation switch + both store paths inlined per fragment the 256 x 256 kernel was 45 k lines
// of assembly whose (mostly skipped) epilogue cost ~20 us of a 65 us GEMM in instruction fetch.
template <int ACT, bool FAST>
__device__ __forceinline__ float act_ct(float v) {
  if constexpr (ACT == ACT_NONE) return v;
  else if constexpr (ACT == ACT_GELU && FAST) return gelu_fast(v);
  else return apply_act(v, ACT);
}

template <typename T, int ACT, bool VEC>
__device__ __forceinline__ void epi_store4(const GemmArgs& a, const float* bias, const float* resid, float* Cf, T* Ct,
                                           int dm, int n, float4v v) {
  if constexpr (VEC) {  // N % 4 == 0: the four columns are all valid, rows are 16-byte aligned
    if (a.col_scale) v *= *reinterpret_cast<const float4v*>(a.col_scale + n);  // fp8 weights: per-row scale
    if (a.swiglu_pairs) {  // columns (n, n+1), (n+2, n+3) are (gate, up) pairs -> output columns n/2, n/2 + 1
      const float o0 = apply_act(v[0], ACT_SILU) * v[1], o1 = apply_act(v[2], ACT_SILU) * v[3];
      const int64_t off = (int64_t)dm * a.ldc + (n >> 1);
      if (a.c_f32) {
        Cf[off] = o0;
        Cf[off + 1] = o1;
      } else if constexpr (is_split<T>::value) {
        st2<T>(Ct + (int64_t)dm * a.ldc, n >> 1, o0, o1);
      } else if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<uint32_t*>(Ct + off) = pack2_from_f32<T>(o0, o1);
      } else {
        Ct[off] = from_f32<T>(o0);
        Ct[off + 1] = from_f32<T>(o1);
      }
      return;
    }
    if (bias) v += *reinterpret_cast<const float4v*>(bias + n);
    v = float4v{act_ct<ACT, sizeof(T) == 2 || is_split<T>::value>(v[0]), act_ct<ACT, sizeof(T) == 2 || is_split<T>::value>(v[1]), act_ct<ACT, sizeof(T) == 2 || is_split<T>::value>(v[2]), act_ct<ACT, sizeof(T) == 2 || is_split<T>::value>(v[3])};
    if (resid) v += *reinterpret_cast<const float4v*>(resid + (int64_t)dm * a.ldr + n);
    if (a.c_f32) {
      *reinterpret_cast<float4v*>(Cf + (int64_t)dm * a.ldc + n) = v;
    } else if constexpr (is_split<T>::value) {
      st4<T>(Ct + (int64_t)dm * a.ldc, n, v[0], v[1], v[2], v[3]);
    } else if constexpr (sizeof(T) == 2) {
      const uint32_t lo = pack2_from_f32<T>(v[0], v[1]);
      const uint32_t hi = pack2_from_f32<T>(v[2], v[3]);
      *reinterpret_cast<uint2*>(Ct + (int64_t)dm * a.ldc + n) = make_uint2(lo, hi);
    } else {
      *reinterpret_cast<float4v*>(Ct + (int64_t)dm * a.ldc + n) = v;
    }
  } else {  // scalar tail path (N not a multiple of 4 / unaligned)
    if (a.swiglu_pairs) {
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        if (n + r + 1 < a.N) {
          float g = v[r], u = v[r + 1];
          if (a.col_scale) { g *= a.col_scale[n + r]; u *= a.col_scale[n + r + 1]; }
          const float o = apply_act(g, ACT_SILU) * u;
          const int64_t off = (int64_t)dm * a.ldc + ((n + r) >> 1);
          if (a.c_f32) Cf[off] = o;
          else st1<T>(Ct + (int64_t)dm * a.ldc, (n + r) >> 1, o);
        }
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (n + r < a.N) {
        float x = v[r];
        if (a.col_scale) x *= a.col_scale[n + r];
        if (bias) x += bias[n + r];
        x = act_ct<ACT, sizeof(T) == 2 || is_split<T>::value>(x);
        if (resid) x += resid[(int64_t)dm * a.ldr + n + r];
        if (a.c_f32)
          Cf[(int64_t)dm * a.ldc + n + r] = x;
        else
          st1<T>(Ct + (int64_t)dm * a.ldc, n + r, x);
      }
    }
  }
}

// PAIRED (the LDS-DMA kernel, 16-bit weights): fragments 2p and 2p + 1 of a wave hold INTERLEAVED column groups of
// the 32 columns they share -- lane group g owns columns 8g .. 8g+3 in fragment 2p and 8g+4 .. 8g+7 in fragment 2p + 1
// (the kernel feeds the MFMA the weight rows in that order: a pure LDS address permutation) -- so a lane holds EIGHT
// consecutive output columns of a row and a 16-bit tile leaves as 16-byte stores: half the store instructions, each
// covering 16 rows x 64 contiguous bytes.  The store tail is bound by the number of (instruction, row) write requests,
// not by bytes (a 4096 x 3840 tile set: 9.2 us of a 16.2 us K = 64 launch, the same for f16 and f32 outputs).
template <int NI, bool PAIRED>
__device__ __forceinline__ int frag_col(int j, int lane) {  // first of the 4 consecutive columns fragment j holds, tile-relative
  if (PAIRED && j < (NI & ~1)) return (j >> 1) * 32 + 8 * (lane >> 4) + 4 * (j & 1);
  return j * 16 + 4 * (lane >> 4);
}

template <typename T, int MI, int NI, int ACT, bool VEC, bool PAIRED = false>
__device__ __forceinline__ void gemm_epilogue_ct(const GemmArgs& a, float4v (&acc)[MI][NI], int mrow0, int ncol0, int z,
                                                 int lane) {
  const float* bias = a.bias ? a.bias + (int64_t)z * a.sBias : nullptr;
  const float* resid = a.resid ? a.resid + (int64_t)z * a.sR : nullptr;
  float* Cf = reinterpret_cast<float*>(a.C) + (int64_t)z * a.sC;
  T* Ct = reinterpret_cast<T*>(a.C) + (int64_t)z * a.sC;
  // NB: every acc index must stay a compile-time constant (full unroll, no `continue`): a runtime-indexed
  // accumulator array is demoted to scratch memory for the WHOLE kernel (guide rule 20; measured 3x slower).
  int dms[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = mrow0 + i * 16 + (lane & 15);
    dms[i] = m < a.M ? (a.row_map ? a.row_map[m] : m) : -1;
  }
  if constexpr (VEC) {
    // LEAN: the 256 x 320 tile holds 160 accumulator registers -- no column-scale vectors (that case takes the
    // per-fragment path below) and the residual rows are not loaded a row ahead (with them the epilogue spilled 124 bytes)
    constexpr bool LEAN = MI * NI * 4 > 128;
    if (!a.swiglu_pairs && !(LEAN && a.col_scale)) {
      // The common epilogue in PHASES.  Written per fragment (epi_store4) the compiler emitted, for each of the MI x NI
      // fragments, a bias load -> s_waitcnt vmcnt(0) -> (residual load -> s_waitcnt vmcnt(0)) -> store chain behind run-time
      // flag branches: 32-40 dependent round trips per wave (bias alone: SAM qkv 46.5 -> 49.3 us, fc1 61.8 -> 66.7 us).
      // Here a lane's NI bias / column-scale vectors are loaded ONCE (they do not depend on the row), and each fragment
      // row issues its NI residual loads together, one row ahead of the row being converted and stored.
      int ns[NI];
      float4v bv[NI], sv[LEAN ? 1 : NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        ns[j] = ncol0 + frag_col<NI, PAIRED>(j, lane);
        const bool in = ns[j] < a.N;
        bv[j] = bias && in ? *reinterpret_cast<const float4v*>(bias + ns[j]) : float4v{0.f, 0.f, 0.f, 0.f};
        if constexpr (!LEAN) {
          sv[j] = a.col_scale && in ? *reinterpret_cast<const float4v*>(a.col_scale + ns[j]) : float4v{1.f, 1.f, 1.f, 1.f};
          sv[j] *= a.alpha;
        }
      }
      // 16-byte stores of a fragment pair: rows 16-byte aligned and the pair never split by the N edge
      const bool pair16 = PAIRED && a.N % 8 == 0 && a.ldc % 8 == 0 && a.sC % 8 == 0 && !((uintptr_t)a.C & 15);
      float4v rv[LEAN ? 1 : 2][NI];
      auto load_resid = [&](auto i_c, float4v (&r)[NI]) {
        constexpr int i = decltype(i_c)::value;
#pragma unroll
        for (int j = 0; j < NI; ++j)
          r[j] = resid && dms[i] >= 0 && ns[j] < a.N ? *reinterpret_cast<const float4v*>(resid + (int64_t)dms[i] * a.ldr + ns[j])
                                                      : float4v{0.f, 0.f, 0.f, 0.f};
      };
      if constexpr (!LEAN) load_resid(std::integral_constant<int, 0>(), rv[0]);
      static_for(std::make_integer_sequence<int, MI>{}, [&](auto i_c) {
        constexpr int i = decltype(i_c)::value, cur = LEAN ? 0 : (i & 1);
        if constexpr (LEAN) load_resid(i_c, rv[0]);
        else if constexpr (i + 1 < MI) load_resid(std::integral_constant<int, i + 1>(), rv[(i + 1) & 1]);
        float4v vv[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          float4v v;
          if constexpr (LEAN) v = acc[i][j] * a.alpha + bv[j];
          else v = acc[i][j] * sv[j] + bv[j];
          v = float4v{act_ct<ACT, sizeof(T) == 2 || is_split<T>::value>(v[0]), act_ct<ACT, sizeof(T) == 2 || is_split<T>::value>(v[1]), act_ct<ACT, sizeof(T) == 2 || is_split<T>::value>(v[2]),
                      act_ct<ACT, sizeof(T) == 2 || is_split<T>::value>(v[3])};
          vv[j] = v + rv[cur][j];
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if (dms[i] >= 0 && ns[j] < a.N) {
            const int64_t off = (int64_t)dms[i] * a.ldc + ns[j];
            if (a.c_f32) {
              *reinterpret_cast<float4v*>(Cf + off) = vv[j];
            } else if constexpr (is_split<T>::value) {
              // both terms of the pair as 8-byte stores (16-byte for a fragment pair: 8 consecutive columns, one 64-block)
              uint16_t* p = reinterpret_cast<uint16_t*>(Ct + (int64_t)dms[i] * a.ldc) + sp_col(ns[j]);
              if constexpr (PAIRED) {
                if (j < (NI & ~1) && pair16) {
                  if ((j & 1) == 0) {
                    uint16_t h[8], l[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                      sp_split(vv[j][e], h[e], l[e]);
                      sp_split(vv[j | 1][e], h[4 + e], l[4 + e]);
                    }
                    *reinterpret_cast<uint4v*>(p) = uint4v{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16),
                                                           (uint32_t)h[4] | ((uint32_t)h[5] << 16), (uint32_t)h[6] | ((uint32_t)h[7] << 16)};
                    *reinterpret_cast<uint4v*>(p + 64) = uint4v{(uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16),
                                                                (uint32_t)l[4] | ((uint32_t)l[5] << 16), (uint32_t)l[6] | ((uint32_t)l[7] << 16)};
                  }
                  continue;
                }
              }
              st4<T>(Ct + (int64_t)dms[i] * a.ldc, ns[j], vv[j][0], vv[j][1], vv[j][2], vv[j][3]);
            } else if constexpr (sizeof(T) == 2) {
              if constexpr (PAIRED) {
                if (j < (NI & ~1) && pair16) {  // (N % 8 == 0: the pair is in range together)
                  if ((j & 1) == 0)
                    *reinterpret_cast<uint4v*>(Ct + off) =
                        uint4v{pack2_from_f32<T>(vv[j][0], vv[j][1]), pack2_from_f32<T>(vv[j][2], vv[j][3]),
                               pack2_from_f32<T>(vv[j | 1][0], vv[j | 1][1]), pack2_from_f32<T>(vv[j | 1][2], vv[j | 1][3])};
                  continue;
                }
              }
              *reinterpret_cast<uint2*>(Ct + off) = make_uint2(pack2_from_f32<T>(vv[j][0], vv[j][1]), pack2_from_f32<T>(vv[j][2], vv[j][3]));
            } else {
              *reinterpret_cast<float4v*>(Ct + off) = vv[j];
            }
          }
        }
      });
      return;
    }
  }
  static_for(std::make_integer_sequence<int, MI * NI>{}, [&](auto ij) {
    constexpr int i = decltype(ij)::value / NI, j = decltype(ij)::value % NI;
    const int n = ncol0 + frag_col<NI, PAIRED>(j, lane);
    if (dms[i] >= 0 && n < a.N) epi_store4<T, ACT, VEC>(a, bias, resid, Cf, Ct, dms[i], n, acc[i][j] * a.alpha);
  });
}

// mrow0 / ncol0: first output row / column of this WAVE's sub-tile (MI x NI fragments of 16 x 16)
template <typename T, int MI, int NI, bool PAIRED = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, float4v (&acc)[MI][NI], int mrow0, int ncol0, int z,
                                              int lane) {
  if (a.vec_ok) {
    switch (a.act) {
      case ACT_NONE: gemm_epilogue_ct<T, MI, NI, ACT_NONE, true, PAIRED>(a, acc, mrow0, ncol0, z, lane); break;
      case ACT_RELU: gemm_epilogue_ct<T, MI, NI, ACT_RELU, true, PAIRED>(a, acc, mrow0, ncol0, z, lane); break;
      case ACT_GELU: gemm_epilogue_ct<T, MI, NI, ACT_GELU, true, PAIRED>(a, acc, mrow0, ncol0, z, lane); break;
      case ACT_QUICK_GELU: gemm_epilogue_ct<T, MI, NI, ACT_QUICK_GELU, true, PAIRED>(a, acc, mrow0, ncol0, z, lane); break;
      default: gemm_epilogue_ct<T, MI, NI, ACT_SILU, true, PAIRED>(a, acc, mrow0, ncol0, z, lane); break;
    }
  } else {
    switch (a.act) {
      case ACT_NONE: gemm_epilogue_ct<T, MI, NI, ACT_NONE, false, PAIRED>(a, acc, mrow0, ncol0, z, lane); break;
      case ACT_RELU: gemm_epilogue_ct<T, MI, NI, ACT_RELU, false, PAIRED>(a, acc, mrow0, ncol0, z, lane); break;
      case ACT_GELU: gemm_epilogue_ct<T, MI, NI, ACT_GELU, false, PAIRED>(a, acc, mrow0, ncol0, z, lane); break;
      case ACT_QUICK_GELU: gemm_epilogue_ct<T, MI, NI, ACT_QUICK_GELU, false, PAIRED>(a, acc, mrow0, ncol0, z, lane); break;
      default: gemm_epilogue_ct<T, MI, NI, ACT_SILU, false, PAIRED>(a, acc, mrow0, ncol0, z, lane); break;
    }
  }
}

template <typename T, int BM, int BN, int BK>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs a) {  // 2 waves/SIMD: 256-register budget
  using M_ = Mma<T>;
  constexpr int VEC = M_::VEC, KS = M_::KS;
  constexpr int LD = BK + VEC;  // +16 B row pad
  constexpr int MI = BM / 32, NI = BN / 32;
  constexpr int KV = BK / VEC;             // vectors per tile row
  constexpr int AV = BM * KV / 256, WV = BN * KV / 256;
  static_assert(BM * KV % 256 == 0 && BN * KV % 256 == 0, "tile/thread mismatch");
  __shared__ __attribute__((aligned(16))) T As[BM * LD];
  __shared__ __attribute__((aligned(16))) T Ws[BN * LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // 1-D grid, XCD-aware bijective remap (T1), then M-fastest tile order: the workgroups that share
  // a weight panel are neighbours on one XCD's L2
  const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN), nwg = tiles_m * tiles_n;
  int id = blockIdx.x;
  if (a.order & 1) {
    const int q = nwg / 8, r = nwg % 8, xcd = id % 8;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8;
  }
  const int m0 = ((a.order & 2) ? id % tiles_m : id / tiles_n) * BM;
  const int n0 = ((a.order & 2) ? id / tiles_m : id % tiles_n) * BN;
  const int z = blockIdx.z;
  const T* __restrict__ A = reinterpret_cast<const T*>(a.A) + (int64_t)z * a.sA;
  const T* __restrict__ W = reinterpret_cast<const T*>(a.W) + (int64_t)z * a.sW;

  float4v acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

  uint4v ra[AV], rw[WV];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int v = tid + i * 256, row = v / KV, kv = v % KV;
      const int gm = m0 + row, gk = k0 + kv * VEC;
      ra[i] = (gm < a.M && gk < a.K)
                  ? *reinterpret_cast<const uint4v*>(A + (int64_t)gm * a.lda + gk)
                  : uint4v{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = tid + i * 256, row = v / KV, kv = v % KV;
      const int gn = n0 + row, gk = k0 + kv * VEC;
      rw[i] = (gn < a.N && gk < a.K)
                  ? *reinterpret_cast<const uint4v*>(W + (int64_t)gn * a.ldw + gk)
                  : uint4v{0, 0, 0, 0};
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int v = tid + i * 256, row = v / KV, kv = v % KV;
      *reinterpret_cast<uint4v*>(&As[row * LD + kv * VEC]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = tid + i * 256, row = v / KV, kv = v % KV;
      *reinterpret_cast<uint4v*>(&Ws[row * LD + kv * VEC]) = rw[i];
    }
  };

  gload(0);
  for (int k0 = 0; k0 < a.K; k0 += BK) {
    sstore();
    __syncthreads();
    if (k0 + BK < a.K) gload(k0 + BK);
#pragma unroll
    for (int ks = 0; ks < BK / KS; ++ks) {
      typename M_::Frag af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        af[i] = M_::load(&As[(wr * (BM / 2) + i * 16 + (lane & 15)) * LD + ks * KS], lane);
#pragma unroll
      for (int j = 0; j < NI; ++j)
        bf[j] = M_::load(&Ws[(wc * (BN / 2) + j * 16 + (lane & 15)) * LD + ks * KS], lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = M_::mma(bf[j], af[i], acc[i][j]);  // C^T tile: see gemm_epilogue
    }
    __syncthreads();
  }

  gemm_epilogue<T, MI, NI>(a, acc, m0 + wr * (BM / 2), n0 + wc * (BN / 2), z, lane);
}

// ---------------------------------------------------------------------------------------------
// bf16 GEMM, LDS-DMA staging (perf mode).  BM x BN x 64 tile, WM x WN waves; both operand tiles go
// global -> LDS with global_load_lds (16 bytes per lane, no VGPR round trip) into TWO stage buffers:
// tile t+1 is requested before tile t is multiplied and is only waited for after, so one raw barrier
// per K tile is the whole synchronisation.  Against the register-staged kernel above (2 barriers per
// tile, one tile of lead) this is 1.2-1.6x on the SAM / prefill / CLIP shapes (scratch/lab/gemm_lab.hip);
// the 256 x 256 tile doubles the FLOPs per L2 byte, which is what bounded the 128^2 kernel (~25 % of
// the MFMA peak = ~9 TB/s of L2 reads).
//
// LDS image: rows of 64 bf16 = 128 bytes, 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7).
// An LDS-DMA instruction writes lane-linearly (base + 16 * lane = 8 rows), so the permutation is applied
// to the SOURCE address each lane fetches and again on the fragment reads (cdna_hip_programming.md
// rule 21); with it the 16 rows a quarter-wave reads for one MFMA operand hit 16 different 16-byte
// slots of the 256-byte bank row.
// ---------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NS stage buffers: tile t+NS-1 is requested right after the barrier of tile t and a counted vmcnt leaves
// NS-2 tiles in flight across it.  NS = 2 (two workgroups per CU for the 128^2 tile) when there are more
// tiles than CUs; NS = 3 when every workgroup has a CU to itself anyway (skinny-M prefill / CLIP shapes
// whose W tiles come from HBM: prefill qkv 71 -> 51 us, CLIP fc1 18 -> 12 us with cold weights).
// W8: the weight operand is fp8 e4m3 bytes (ANYREF_MODE_PERF_FP8W): its tile is DMA'd as bytes (64 B per row,
// 16 rows per wave instruction, 16-byte chunk c of row r at chunk c ^ ((r >> 2) & 3): conflict-free 8-byte
// fragment reads), each fragment is widened to bf16 in registers (exact: e4m3 is a subset of bf16) right
// before the bf16 MFMA, and the per-row scale of the weight multiplies the accumulator column in the epilogue.
__device__ __forceinline__ short8 fp8x8_to_bf16x8(uint2v r) {
  short8 o;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float2v lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[h], false);
    const float2v hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[h], true);
    o[4 * h] = (short)f2bf(lo[0]).x; o[4 * h + 1] = (short)f2bf(lo[1]).x;
    o[4 * h + 2] = (short)f2bf(hi[0]).x; o[4 * h + 3] = (short)f2bf(hi[1]).x;
  }
  return o;
}

// PERSIST: the grid is CAPPED (GemmArgs::max_wg workgroups, a multiple of 8) and every workgroup walks the tiles
// id, id + gridDim.x, ... one after the other -- the SAM encoder's launches on the side stream, which must leave
// CUs to the decode GEMVs of the main stream (DESIGN.md "CU share of the side stream").  A separate instantiation:
// the one-tile-per-workgroup kernels are untouched.
template <int BM, int BN, int WM, int WN, int NS, bool W8 = false, bool PERSIST = false, typename TT = bf16>
__global__ __launch_bounds__(WM * WN * 64) void gemm_glds_kernel(GemmArgs a) {
  using T = TT;  // bf16 or f16 (same tile, DMA and fragment code; the MFMA and the epilogue rounding differ)
  // sp16 (split-pair activations, common.h): the A rows are 2K/64 bf16 tiles [hi | lo] per 64 logical columns, walked
  // against K/64 tiles of the bf16 weight (W tile = A tile >> 1, re-fetched from L2 for the lo pass); the epilogue
  // writes f32 or a split pair again.  ET: element type of both operand tiles and of the MFMA.
  constexpr bool SPLIT = is_split<TT>::value;
  using ET = std::conditional_t<SPLIT, bf16, TT>;
  constexpr int AX = SPLIT ? 2 : 1;  // 16-bit elements per logical A element
  static_assert(!W8 || !(is_half16<T>::value || SPLIT), "fp8 weights are widened to bf16");
  constexpr int BK = 64, NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
  constexpr int ROWB = BK * 2;             // bytes per A tile row
  constexpr int WROWB = W8 ? BK : BK * 2;  // bytes per W tile row
  constexpr int WRPI = W8 ? 16 : 8;        // W rows per wave DMA instruction
  constexpr int TILEB = BM * ROWB + BN * WROWB;  // one stage
  // LDS-DMA rounds: RA / RWF full rounds of all NW waves; a BN that is not a multiple of NW * WRPI rows adds a
  // last round that only waves 0 .. PW-1 take part in (128 x 160: 256 equal tiles for SAM fc2, 4096 x 1280)
  constexpr int RA = BM / (NW * 8), RWF = BN / (NW * WRPI), PW = (BN % (NW * WRPI)) / WRPI, RW = RWF + (PW > 0);
  constexpr int LPT = RA + RWF;                          // DMA instructions per lane per tile (+1 on waves < PW)
  // 16-bit weights: the wave's W fragments come in PAIRS with interleaved rows (see frag_col above: lane i = 4 g' + r' of
  // fragment 2p + q reads tile row 32 p + 8 g' + 4 q + r'), so the epilogue holds 8 consecutive columns per lane.  The W
  // tile's chunk swizzle is keyed on the row bits that differ among those 16 rows (bit 1 and bits 3-4; rows 2k, 2k + 1
  // already sit in different halves of the 256-byte bank row): conflict-free like (row >> 1) & 7 is for 16 consecutive
  // rows (the A tile, and an odd last fragment -- 80-column waves -- which reads consecutive rows at 2-way conflicts).
#ifdef ANYREF_GEMM_NO_PAIRW  // lab build: the unpaired fragment order
  constexpr bool PAIRW = false;
#else
  // (not the 128 x 160 tile -- SAM fc2, f32 output: nothing to gain from 16-byte stores and its odd fifth fragment's
  //  2-way conflicts cost 1.3 us of 61.6; on one box, unpaired -> paired: qkv 45.3 -> 41.6 us, fc1 62.6 -> 59.8, proj 20.8 -> 20.0)
  constexpr bool PAIRW = !W8 && !(NI % 2 == 1 && BM < 256);
#endif
  constexpr int NIP = PAIRW ? (NI & ~1) : 0;  // fragments that come in pairs
  static_assert(BM % (NW * 8) == 0 && BN % WRPI == 0 && (PW == 0 || !W8), "tile rows must split over the waves");
  static_assert(NS >= 2 && NS <= 4 && (NS - 2) * (LPT + 1) <= 63, "stage count / vmcnt range");
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // the ONLY LDS object (rule: one array)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  auto wswz = [](int row) { return PAIRW ? ((row >> 1) & 1) | (((row >> 3) & 3) << 1) : (row >> 1) & 7; };
  // W tile row lane (l & 15) reads for the wave's fragment f
  auto wrow = [&](int f) {
    const int i = lane & 15;
    return wc * TN + (f < NIP ? (f >> 1) * 32 + 8 * (i >> 2) + 4 * (f & 1) + (i & 3) : f * 16 + i);
  };
  const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN), nwg = tiles_m * tiles_n;
  for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {  // (one pass unless PERSIST: see the end of the body)
  int id = vb;
  if (a.order & 1) {
    const int q = nwg / 8, r = nwg % 8, xcd = id % 8;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8;
  }
  int tm, tn;
  if (a.group_m > 0) {
    // grouped order: GM tile rows at a time, column by column inside the group, so the contiguous chunk
    // of tile ids one XCD receives is a GM x (chunk / GM) RECTANGLE of the output (its A rows stay in
    // that XCD's L2, every W panel is fetched once per XCD) and the ~64 workgroups an XCD runs at once
    // form a near-square block.  With plain M-fastest order an XCD walks whole columns and re-fetches
    // all of A for each: 1.5-2x the L2-miss (fabric) traffic.
    const int per = a.group_m * tiles_n, g = id / per, first = g * a.group_m;
    const int gsz = tiles_m - first < a.group_m ? tiles_m - first : a.group_m;
    tm = first + (id % per) % gsz;
    tn = (id % per) / gsz;
  } else {
    tm = (a.order & 2) ? id % tiles_m : id / tiles_n;
    tn = (a.order & 2) ? id / tiles_m : id % tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z;
  const ET* __restrict__ A = reinterpret_cast<const ET*>(a.A) + (int64_t)z * a.sA * AX;
  using WT = std::conditional_t<W8, uint8_t, ET>;
  const WT* __restrict__ W = reinterpret_cast<const WT*>(a.W) + (int64_t)z * a.sW;

  float4v acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

  // rows past M / N fetch the last valid row (never stored); K is a multiple of 64 (launcher)
  const int srow = lane >> 3, sp = lane & 7;
  const ET* asrc[RA];
  const WT* wsrc[RW];
#pragma unroll
  for (int r = 0; r < RA; ++r) {
    const int row = (r * NW + wave) * 8 + srow;
    int gm = m0 + row;
    gm = gm < a.M ? gm : a.M - 1;
    if (a.a_row_map) gm = a.a_row_map[gm];
    asrc[r] = A + (int64_t)gm * a.lda * AX + ((sp ^ ((row >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    if constexpr (W8) {
      const int row = (r * NW + wave) * 16 + (lane >> 2);
      int gn = n0 + row;
      gn = gn < a.N ? gn : a.N - 1;
      wsrc[r] = W + (int64_t)gn * a.ldw + (((lane & 3) ^ ((row >> 2) & 3)) << 4);
    } else {
      const int row = (r * NW + wave) * 8 + srow;
      int gn = n0 + row;
      gn = gn < a.N ? gn : a.N - 1;
      wsrc[r] = W + (int64_t)gn * a.ldw + ((sp ^ wswz(row)) << 3);
    }
  }
  // The stage index is a compile-time constant (loop unrolled by two below): with a runtime index hipcc
  // cannot tell the DMA destination from the buffer being read and puts s_waitcnt vmcnt(0) in front of the
  // first ds_read of every tile, which serialises the prefetch (seen in the .s; -40 % on 256^2).
  auto stage = [&](auto buf_c, int t) {
    constexpr int buf = decltype(buf_c)::value;
    char* base = smem + buf * TILEB;
#pragma unroll
    for (int r = 0; r < RA; ++r)
      __builtin_amdgcn_global_load_lds((gas_ptr)(asrc[r] + t * BK), (las_ptr)(base + (r * NW + wave) * 8 * ROWB), 16, 0,
                                       0);
#pragma unroll
    for (int r = 0; r < RW; ++r)
      if (r < RWF || wave < PW)  // (wave-uniform)
        __builtin_amdgcn_global_load_lds((gas_ptr)(wsrc[r] + (SPLIT ? t >> 1 : t) * BK),
                                         (las_ptr)(base + BM * ROWB + (r * NW + wave) * WRPI * WROWB), 16, 0, 0);
  };
  // wait until at most N tiles' worth of this wave's own DMAs are still in flight
  auto wait_tiles = [&](auto n_c) {
    constexpr int N = decltype(n_c)::value;
    if constexpr (PW == 0 || N == 0) {
      wait_vmcnt<N * LPT>();
    } else {
      if (wave < PW) wait_vmcnt<N * (LPT + 1)>();
      else wait_vmcnt<N * LPT>();
    }
  };
  // 256-row tiles (8 waves, 128 x 64 / 128 x 80 per wave): the K tile is worked off in four QUADRANT phases of the
  // wave's output (rows i in {0,1} x columns j in {0,1}, order 00 01 11 10 so that one operand's fragments carry
  // over), the fragments of phase p + 1 requested before the MFMAs of phase p, s_setprio 1 around every MFMA cluster
  // (guide T5: keeps hipcc from drifting the MFMAs in among the LDS reads).  Lab (scratch/lab/gemm8_lab.hip,
  // interleaved rounds): SAM qkv 44.4 -> 41.5 us, fc1-shaped 77 -> 73, 4096^3 +4 %, 8192^3 +3 %.
  constexpr bool QUAD = BM == 256 && WM == 2 && !W8 && MI == 8 && NI == 4;  // (256 x 320: 104 fragment + 160 accumulator registers would spill)
  auto compute_quad = [&](auto buf_c) {
    constexpr int buf = decltype(buf_c)::value;
    constexpr int MH = MI / 2, NH0 = NI / 2, NH1 = NI - NH0;
    const char* Ab = smem + buf * TILEB;
    const char* Wb = Ab + BM * ROWB;
    short8 a0[2][MH], a1[2][MH], b0[2][NH0], b1[2][NH1];
    auto lda = [&](auto i_c, short8 (&af)[2][MH]) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < MH; ++f) {
          const int row = wr * TM + (decltype(i_c)::value * MH + f) * 16 + (lane & 15), c = ks * 4 + (lane >> 4);
          af[ks][f] = *reinterpret_cast<const short8*>(Ab + row * ROWB + ((c ^ ((row >> 1) & 7)) << 4));
        }
    };
    auto ldw0 = [&]() {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < NH0; ++f) {
          const int row = wrow(f), c = ks * 4 + (lane >> 4);
          b0[ks][f] = *reinterpret_cast<const short8*>(Wb + row * ROWB + ((c ^ wswz(row)) << 4));
        }
    };
    auto ldw1 = [&]() {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < NH1; ++f) {
          const int row = wrow(NH0 + f), c = ks * 4 + (lane >> 4);
          b1[ks][f] = *reinterpret_cast<const short8*>(Wb + row * ROWB + ((c ^ wswz(row)) << 4));
        }
    };
    auto mma = [&](auto i_c, auto j_c, const short8 (&af)[2][MH], const auto& bf) {
      constexpr int i = decltype(i_c)::value, j = decltype(j_c)::value, NH = j == 0 ? NH0 : NH1;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < MH; ++f)
#pragma unroll
          for (int h = 0; h < NH; ++h)
            acc[i * MH + f][j * NH0 + h] =
                mfma_16x16x32<ET>(bf[ks][h], af[ks][f], acc[i * MH + f][j * NH0 + h]);  // C^T tile
      __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    ldw0();
    lda(I0(), a0);
    ldw1();
    mma(I0(), I0(), a0, b0);
    lda(I1(), a1);
    mma(I0(), I1(), a0, b1);
    mma(I1(), I1(), a1, b1);
    mma(I1(), I0(), a1, b0);
  };
  auto compute = [&](auto buf_c) {
    if constexpr (QUAD) {
      compute_quad(buf_c);
      return;
    }
    constexpr int buf = decltype(buf_c)::value;
    const char* Ab = smem + buf * TILEB;
    const char* Wb = Ab + BM * ROWB;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      short8 af[MI], bfr[NI];
      const int c = ks * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wr * TM + i * 16 + (lane & 15);
        af[i] = *reinterpret_cast<const short8*>(Ab + row * ROWB + ((c ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = wrow(j);
        if constexpr (W8) {
          // k = 32 ks + 8 g .. + 7: the 8 bytes at chunk (2 ks + g / 2), half (g & 1) of the 64-byte row
          const int g = lane >> 4, c8 = ks * 2 + (g >> 1);
          bfr[j] = fp8x8_to_bf16x8(
              *reinterpret_cast<const uint2v*>(Wb + row * WROWB + ((c8 ^ ((row >> 2) & 3)) << 4) + ((g & 1) << 3)));
        } else {
          bfr[j] = *reinterpret_cast<const short8*>(Wb + row * ROWB + ((c ^ wswz(row)) << 4));
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = mfma_16x16x32<ET>(bfr[j], af[i], acc[i][j]);  // C^T tile
    }
  };
  const int nt = a.K / BK * AX;
  static_for(std::make_integer_sequence<int, NS - 1>{}, [&](auto b) {  // prologue: tiles 0 .. NS-2
    if (decltype(b)::value < nt) stage(b, decltype(b)::value);
  });
  for (int t0 = 0; t0 < nt; t0 += NS) {
    static_for(std::make_integer_sequence<int, NS>{}, [&](auto b) {
      constexpr int B = decltype(b)::value;
      const int t = t0 + B;
      if (t < nt) {
        // own DMAs of tile t have landed once at most min(NS-2, nt-1-t) younger tiles are still in flight
        const int behind = nt - 1 - t;
        if (behind >= NS - 2) wait_tiles(std::integral_constant<int, NS - 2>());
        else if (NS > 3 && behind == 1) wait_tiles(std::integral_constant<int, 1>());
        else wait_tiles(std::integral_constant<int, 0>());
        __builtin_amdgcn_s_barrier();  // ... and so have everyone else's; all waves are done with tile t-1
        if (t + NS - 1 < nt) stage(std::integral_constant<int, (B + NS - 1) % NS>(), t + NS - 1);  // into t-1's buffer
        compute(b);
      }
    });
  }
  gemm_epilogue<T, MI, NI, PAIRW>(a, acc, m0 + wr * TM, n0 + wc * TN, z, lane);
  if constexpr (!PERSIST) break;
  __syncthreads();  // every wave is done with this tile's LDS stages before the next tile's first DMA lands
  }
}

// ---------------------------------------------------------------------------------------------
// Split-K for skinny, deep GEMMs (LLM prefill o_proj / down_proj at M = 320, CLIP fc2): with one
// 64x64 workgroup per CU and a single tile of prefetch they are HBM-LATENCY bound (0.58 us per
// k-step measured, 10x the MFMA time).  K is cut into `splits` slices run as the batch dimension
// into an f32 slab buffer; a second tiny kernel sums the slabs in a fixed order (deterministic, no
// float atomics) and applies bias / activation / residual.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, int splits, int64_t slab_stride, GemmArgs a) {
  const int64_t total = (int64_t)a.M * a.N / 4;
  float* Cf = reinterpret_cast<float*>(a.C);
  T* Ct = reinterpret_cast<T*>(a.C);
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < total; v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = v * 4;
    const int m = (int)(e / a.N), n = (int)(e % a.N);
    float4v acc = *reinterpret_cast<const float4v*>(slabs + e);
    for (int z = 1; z < splits; ++z) acc += *reinterpret_cast<const float4v*>(slabs + z * slab_stride + e);
    acc *= a.alpha;
    if (a.col_scale) acc *= *reinterpret_cast<const float4v*>(a.col_scale + n);
    if (a.swiglu_pairs) {
      const float o0 = apply_act(acc[0], ACT_SILU) * acc[1], o1 = apply_act(acc[2], ACT_SILU) * acc[3];
      const int64_t off = (int64_t)m * a.ldc + (n >> 1);
      if (a.c_f32) { Cf[off] = o0; Cf[off + 1] = o1; }
      else if constexpr (is_split<T>::value) st2<T>(Ct + (int64_t)m * a.ldc, n >> 1, o0, o1);
      else { Ct[off] = from_f32<T>(o0); Ct[off + 1] = from_f32<T>(o1); }
      continue;
    }
    if (a.bias) acc += *reinterpret_cast<const float4v*>(a.bias + n);
    acc = float4v{apply_act(acc[0], a.act), apply_act(acc[1], a.act), apply_act(acc[2], a.act), apply_act(acc[3], a.act)};
    if (a.resid) acc += *reinterpret_cast<const float4v*>(a.resid + (int64_t)m * a.ldr + n);
    if (a.c_f32) {
      *reinterpret_cast<float4v*>(Cf + (int64_t)m * a.ldc + n) = acc;
    } else if constexpr (is_split<T>::value) {
      st4<T>(Ct + (int64_t)m * a.ldc, n, acc[0], acc[1], acc[2], acc[3]);
    } else {
      T* o = Ct + (int64_t)m * a.ldc + n;
      o[0] = from_f32<T>(acc[0]); o[1] = from_f32<T>(acc[1]); o[2] = from_f32<T>(acc[2]); o[3] = from_f32<T>(acc[3]);
    }
  }
}

// Split-K reduction of one output ROW per workgroup with the RMSNorm that follows fused in: sums the slabs
// in a fixed order, applies bias / activation / residual, writes C, then y = C * rsqrt(mean C^2 + eps) * gain
// as T.  Saves a launch and a pass over x per prefill o_proj / down_proj (N <= 8192).
// NT threads per row, SP slabs (0: run-time count).  With 1024 threads a 4096-wide row is one float4 column per
// thread and the SP slab loads + the residual load of a thread are straight-line code, all in flight together; the
// 256-thread / run-time-count form walked 4 columns x 4 slabs as ~16 dependent round trips (11.6 us per launch).
// MAXV: float4 columns per thread, N <= 4 * NT * MAXV (instantiated tight: the unrolled column loop is code the
// instruction fetch pays for whether or not a column is live)
template <typename T, int NT, int SP, int MAXV>
__global__ __launch_bounds__(NT) void splitk_reduce_norm_kernel(const float* __restrict__ slabs, int splits,
                                                                int64_t slab_stride, GemmArgs a) {
  const int m = blockIdx.x, tid = threadIdx.x, nv = a.N / 4;
  float* Cf = reinterpret_cast<float*>(a.C);
  T* Ct = reinterpret_cast<T*>(a.C);
  float4v v[MAXV];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = tid + i * NT;
    if (c < nv) {
      const int n = c * 4;
      const int64_t e = (int64_t)m * a.N + n;
      float4v acc;
      if constexpr (SP > 0) {
        float4v t[SP];
#pragma unroll
        for (int z = 0; z < SP; ++z) t[z] = *reinterpret_cast<const float4v*>(slabs + z * slab_stride + e);
        acc = t[0];
#pragma unroll
        for (int z = 1; z < SP; ++z) acc += t[z];  // same order as the run-time loop: bit-identical
      } else {
        acc = *reinterpret_cast<const float4v*>(slabs + e);
        for (int z = 1; z < splits; ++z) acc += *reinterpret_cast<const float4v*>(slabs + z * slab_stride + e);
      }
      acc *= a.alpha;
      if (a.col_scale) acc *= *reinterpret_cast<const float4v*>(a.col_scale + n);
      if (a.bias) acc += *reinterpret_cast<const float4v*>(a.bias + n);
      acc = float4v{apply_act(acc[0], a.act), apply_act(acc[1], a.act), apply_act(acc[2], a.act), apply_act(acc[3], a.act)};
      if (a.resid) acc += *reinterpret_cast<const float4v*>(a.resid + (int64_t)m * a.ldr + n);
      if (a.c_f32) {
        *reinterpret_cast<float4v*>(Cf + (int64_t)m * a.ldc + n) = acc;
      } else if constexpr (is_split<T>::value) {
        st4<T>(Ct + (int64_t)m * a.ldc, n, acc[0], acc[1], acc[2], acc[3]);
      } else {
        T* o = Ct + (int64_t)m * a.ldc + n;
        o[0] = from_f32<T>(acc[0]); o[1] = from_f32<T>(acc[1]); o[2] = from_f32<T>(acc[2]); o[3] = from_f32<T>(acc[3]);
      }
      v[i] = acc;
      ss += acc[0] * acc[0] + acc[1] * acc[1] + acc[2] * acc[2] + acc[3] * acc[3];
    } else {
      v[i] = float4v{0.f, 0.f, 0.f, 0.f};
    }
  }
  __shared__ float red[NT / 64];
  float mean = 0.f;
  if (a.norm_bias) {  // LayerNorm (CLIP / audio blocks): two passes over the registers, as norm_kernel does
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) s1 += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    s1 = wave_sum(s1);
    if ((tid & 63) == 0) red[tid >> 6] = s1;
    __syncthreads();
    float tot1 = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) tot1 += red[w];
    mean = tot1 / (float)a.N;
    __syncthreads();
    ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (tid + i * NT < nv) {
        v[i] -= mean;
        ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
      }
    }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) tot += red[w];
  const float scale = rsqrtf(tot / (float)a.N + a.norm_eps);
  T* y = reinterpret_cast<T*>(a.norm_out) + (int64_t)m * a.norm_ld;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = tid + i * NT;
    if (c < nv) {
      const int n = c * 4;
      const float4v g = *reinterpret_cast<const float4v*>(a.norm_gain + n);
      float4v o = v[i] * scale * g;
      if (a.norm_bias) o += *reinterpret_cast<const float4v*>(a.norm_bias + n);
      st4<T>(y, n, o[0], o[1], o[2], o[3]);
    }
  }
}

// per-stream split-K slab workspace: defined once, in gemm.hip
float* splitk_workspace(hipStream_t s, size_t bytes);

// Measurement knobs (read once, scratch/bench_gemm*.py A/B runs; never set in production):
//   ANYREF_GEMM_NO_GLDS    bf16: use the register-staged kernel instead of the LDS-DMA one
//   ANYREF_GEMM_NO_SPLITK  never split K
//   ANYREF_GEMM_GM=n       grouped tile order with n tile rows per group (0: plain M-fastest)
//   ANYREF_GEMM_TILE=0..3  register-staged kernel: force 64x64 / 64x128 / 128x64 / 128x128
//   ANYREF_GEMM_M320=-1    no whole-M (320-row) tiles for prefill gate/up and the split-K slabs
struct GemmKnobs {
  bool no_glds = getenv("ANYREF_GEMM_NO_GLDS") != nullptr;
  bool no_splitk = getenv("ANYREF_GEMM_NO_SPLITK") != nullptr;
  int gm = getenv("ANYREF_GEMM_GM") ? atoi(getenv("ANYREF_GEMM_GM")) : -1;
  int tile = getenv("ANYREF_GEMM_TILE") ? atoi(getenv("ANYREF_GEMM_TILE")) : -1;
  int m320 = getenv("ANYREF_GEMM_M320") ? atoi(getenv("ANYREF_GEMM_M320")) : 0;  // -1: no 320-row tiles
  int ns320 = getenv("ANYREF_GEMM_NS320") ? atoi(getenv("ANYREF_GEMM_NS320")) : 3;  // stages of the 320 x 96 tile (2: round 2)
  int force128 = getenv("ANYREF_GEMM_FORCE128") ? atoi(getenv("ANYREF_GEMM_FORCE128")) : 0;  // probe: 3 = 128^2 NS3, 2 = NS2
};
static const GemmKnobs& knobs() {
  static const GemmKnobs k;
  return k;
}

template <typename T>
void launch_gemm(const GemmArgs& a_in, hipStream_t s) {
  GemmArgs a = a_in;
  if (a.norm_done) *a.norm_done = false;
  if (a.swiglu_pairs && (a.N % 4 || a.bias || a.resid || a.row_map || a.act != ACT_NONE || a.batch != 1))
    throw std::runtime_error("gemm: the SwiGLU epilogue takes interleaved gate/up rows, N % 4 == 0, and nothing else");
  constexpr int VEC = Mma<T>::VEC;
  constexpr bool SP = is_split<T>::value;         // split-pair A (and C unless c_f32), bf16 W
  constexpr bool IS16 = sizeof(T) == 2 || SP;     // the 16-bit MFMA paths
  if constexpr (SP) {
    if (a.K % 64 || a.lda % 64 || a.sA % 64 || a.w_fp8 || (!a.c_f32 && !a.slabs_out && (a.ldc % 64 || a.sC % 64)) ||
        (a.norm_out && a.norm_ld % 64))
      throw std::runtime_error("gemm<sp16>: K, lda, the A batch stride (and ldc of a split-pair output) must be multiples of 64");
  }
  if (a.slabs_out && a.slabs > 1) {  // raw split-K: the consumer sums the slices
    if (!IS16 || a.batch != 1 || a.row_map || a.bias || a.resid || a.act != ACT_NONE || a.swiglu_pairs || a.w_fp8 ||
        a.K % (64 * a.slabs))
      throw std::runtime_error("gemm: raw split-K slabs take a plain bf16 product with K % (64 * slabs) == 0");
    GemmArgs g = a;
    g.K = a.K / a.slabs;
    g.batch = a.slabs;
    g.sA = g.K;
    g.sW = g.K;
    g.C = a.slabs_out; g.ldc = a.N; g.sC = (int64_t)a.M * a.N; g.c_f32 = 1;
    g.slabs_out = nullptr; g.slabs = 0; g.norm_out = nullptr;
    launch_gemm<T>(g, s);
    return;
  }
  // ---- split-K decision: few 64x64 tiles, deep K, plain row-major output ----
  // (K >= 1024 when a norm rides on the reduction: CLIP out_proj, 24 tiles of 16 K steps + a LayerNorm launch otherwise)
  if (!knobs().no_splitk && IS16 && a.batch == 1 && !a.row_map && a.M <= 512 &&
      (a.K >= 2048 || (a.K >= 1024 && a.norm_out && a.norm_bias)) && a.N % 4 == 0 && a.ldc % 4 == 0 &&
      (!a.resid || a.ldr % 4 == 0)) {
    // 128 x 128 workgroups (two per CU): split until there are ~256 of them, slices of >= 512, multiples of 64
    const int64_t tiles = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 128);
    int splits = 1;
    while (splits < 8 && tiles * splits < 256 && (a.K / (splits * 2)) % 64 == 0 && a.K / (splits * 2) >= 512) splits *= 2;
    if (splits > 1) {
      const int64_t slab = (int64_t)a.M * a.N;
      float* ws = splitk_workspace(s, (size_t)splits * slab * sizeof(float));
      GemmArgs g = a;
      g.K = a.K / splits;
      g.batch = splits;
      g.sA = g.K;   // column offset inside the same rows
      g.sW = g.K;
      g.C = ws; g.ldc = a.N; g.sC = slab; g.c_f32 = 1;
      g.bias = nullptr; g.resid = nullptr; g.act = ACT_NONE; g.alpha = 1.f; g.col_scale = nullptr; g.swiglu_pairs = 0;
      launch_gemm<T>(g, s);
      if (a.norm_out && a.norm_gain && a.N <= 8192 && a.norm_ld % 4 == 0 && !a.swiglu_pairs) {
        auto rn = [&](auto nt_t, auto sp_t, auto mv_t) {
          constexpr int NT = decltype(nt_t)::value, SP = decltype(sp_t)::value, MV = decltype(mv_t)::value;
          hipLaunchKernelGGL((splitk_reduce_norm_kernel<T, NT, SP, MV>), dim3(a.M), dim3(NT), 0, s, ws, splits, slab, a);
        };
        using C0 = std::integral_constant<int, 0>;
        using C1 = std::integral_constant<int, 1>;
        using C2 = std::integral_constant<int, 2>;
        using C4 = std::integral_constant<int, 4>;
        using C8 = std::integral_constant<int, 8>;
        using C256 = std::integral_constant<int, 256>;
        using C1024 = std::integral_constant<int, 1024>;
        if (a.N <= 1024) {  // CLIP (N = 1024): one column per thread of a 256-thread row
          if (splits == 8) rn(C256(), C8(), C1());
          else if (splits == 2) rn(C256(), C2(), C1());
          else rn(C256(), C0(), C1());
        } else if (a.N <= 4096) {  // LLM 7B (N = 4096)
          if (splits == 4) rn(C1024(), C4(), C1());
          else rn(C1024(), C0(), C1());
        } else {
          if (splits == 4) rn(C1024(), C4(), C2());
          else rn(C1024(), C0(), C2());
        }
        if (a.norm_done) *a.norm_done = true;
        return;
      }
      const int64_t total = slab / 4;
      const int grid = (int)(cdiv64(total, 256) < 2048 ? cdiv64(total, 256) : 2048);
      hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(grid), dim3(256), 0, s, ws, splits, slab, a);
      return;
    }
  }
  constexpr int BK = sizeof(T) == 2 ? 64 : 16;
  if (a.M <= 0 || a.N <= 0) return;
  if (a.a_row_map && (!IS16 || a.K % 64 || knobs().no_glds || knobs().tile >= 0 || a.batch != 1))
    throw std::runtime_error("gemm: a_row_map is taken by the 16-bit LDS-DMA kernel only (K % 64 == 0, batch 1)");
  if (a.K % VEC || a.lda % VEC || a.ldw % VEC || ((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15) ||
      (a.sA % VEC) || (a.sW % VEC))
    throw std::runtime_error("gemm: K/lda/ldw must be multiples of 16 bytes and operands 16-byte aligned");
  // Tile choice, from measurements on MI355X (scratch/bench_gemm.py, all four variants per shape):
  // 128x128 once there are >= 2 tiles per CU; 64x128 for mid-size N (more, smaller tiles fill the
  // chip) and for skinny-M / wide-N weight-streaming shapes; 64x64 when even that leaves CUs idle.
  const int64_t t128 = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 128) * a.batch;
  const int64_t t64x128 = (int64_t)cdiv(a.M, 64) * cdiv(a.N, 128) * a.batch;
  bool bm128;
  int bn;
  if (a.M <= 512) {
    bm128 = false;
    bn = t64x128 >= 384 ? 128 : 64;
  } else if (t128 >= 512) {
    bm128 = true;
    bn = 128;
  } else {
    bm128 = false;
    bn = t64x128 >= 256 ? 128 : 64;
  }
  if (knobs().tile >= 0) {
    const int v = knobs().tile;
    bm128 = v & 2;
    bn = (v & 1) ? 128 : 64;
  }
  dim3 block(256);
  a.order = 3;
  {
    const int ov = a.c_f32 ? 4 : (int)sizeof(T);  // output element bytes
    const bool al = a.N % 4 == 0 && a.ldc % 4 == 0 && a.sC % 4 == 0 && !((uintptr_t)a.C & 15) &&
                    (!a.resid || (a.ldr % 4 == 0 && a.sR % 4 == 0 && !((uintptr_t)a.resid & 15))) &&
                    (!a.bias || (a.sBias % 4 == 0 && !((uintptr_t)a.bias & 15)));
    a.vec_ok = al && (ov == 4 || ov == 2) ? 1 : 0;
    if (a.swiglu_pairs) a.vec_ok = (a.N % 4 == 0 && a.ldc % 2 == 0 && !((uintptr_t)a.C & 3) && (!a.col_scale || !((uintptr_t)a.col_scale & 15))) ? 1 : 0;
  }
  const double flops = 2.0 * a.M * a.N * (double)a.K * a.batch;
  // algorithmic bytes: A (pairs: 4 bytes per element), W as stored (fp8: 1 byte, pairs mode: bf16), C
  const double wbytes = a.w_fp8 ? 1.0 : (SP ? 2.0 : (double)sizeof(T));
  const double bytes = ((double)a.M * a.K * sizeof(T) + (double)a.N * a.K * wbytes) * a.batch +
                       (double)a.M * a.N * (a.c_f32 ? 4 : sizeof(T)) * a.batch;
  if (a.w_fp8 && (sizeof(T) != 2 || a.K % 64 || knobs().no_glds || knobs().tile >= 0))
    throw std::runtime_error("gemm: an fp8 weight operand needs the bf16 LDS-DMA kernel (K % 64 == 0)");
  if constexpr (IS16) {
    if (SP || (!knobs().no_glds && a.K % 64 == 0 && knobs().tile < 0)) {
      // Tile choice from scratch/lab/gemm_lab.hip on MI355X: 256^2 when its tiles fill whole rounds of the
      // 256 CUs (SAM qkv: 240 tiles, square 8192^3), 64 x 256 for skinny-M / very wide N (prefill gate/up),
      // otherwise 128^2 with 8 waves (two workgroups per CU).
      auto go = [&](auto bm_t, auto bn_t, auto wm_t, auto wn_t, auto ns_t, const char* tag) {
        constexpr int BM = decltype(bm_t)::value, BN = decltype(bn_t)::value, WM = decltype(wm_t)::value,
                      WN = decltype(wn_t)::value, NS = decltype(ns_t)::value;
        constexpr size_t lds = NS * (size_t)(BM + BN) * 128;
        if constexpr (BN % (WM * WN * 16) == 0 && !is_half16<T>::value && !SP) {
          if (a.w_fp8) {  // fp8 weight operand (LDS of the full-width kernel is an upper bound)
            auto kern8 = &gemm_glds_kernel<BM, BN, WM, WN, NS, true>;
            static KernelAttrOnce once8;
            ensure_dyn_lds(once8, reinterpret_cast<const void*>(kern8), (int)lds);
            const int tiles_m8 = cdiv(a.M, BM), nwg8 = tiles_m8 * cdiv(a.N, BN);
            int gm8 = (int)lround(sqrt((double)(nwg8 > 8 ? nwg8 / 8 : 1)));
            a.group_m = gm8 < 1 ? 1 : (gm8 > tiles_m8 ? tiles_m8 : gm8);
            char tag8[56];
            snprintf(tag8, sizeof(tag8), "%s_fp8w%s", tag, a.M <= 16 ? "_dec" : "");  // _dec: decode rows (weight streaming)
            ProfScope prof(tag8, flops, bytes, s);
            hipLaunchKernelGGL(kern8, dim3(nwg8, 1, a.batch), dim3(WM * WN * 64), NS * ((size_t)BM * 128 + (size_t)BN * 64),
                               s, a);
            return;
          }
        }
        auto kern = &gemm_glds_kernel<BM, BN, WM, WN, NS, false, false, T>;
        static KernelAttrOnce once;  // (per instantiation of this generic lambda, per device)
        ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), (int)lds);
        {  // rows per group ~ sqrt(tiles one XCD gets), so its chunk is a near-square rectangle
          const int tiles_m = cdiv(a.M, BM), nwg = tiles_m * cdiv(a.N, BN);
          int gm = (int)lround(sqrt((double)(nwg > 8 ? nwg / 8 : 1)));
          gm = gm < 1 ? 1 : (gm > tiles_m ? tiles_m : gm);
          a.group_m = knobs().gm >= 0 ? knobs().gm : gm;
        }
        // gemm_bf16_... -> gemm_f16_... / gemm_sp16_...; "_dec": at most 16 rows (the MFMA decode path: HBM-bound weight streaming,
        // booked apart from the MFMA-bound prefill launches of the same tile)
        char tagt[56];
        snprintf(tagt, sizeof(tagt), "%s%s%s", SP ? "gemm_sp16_" : is_half16<T>::value ? "gemm_f16_" : "gemm_bf16_", tag + 10,
                 a.M <= 16 ? "_dec" : "");
        ProfScope prof(tagt, flops, bytes, s);
        dim3 grid(cdiv(a.N, BN) * cdiv(a.M, BM), 1, a.batch);
        if constexpr (BM <= 256 && BM >= 128) {  // the SAM encoder's tiles
          // (tiles small enough for two workgroups per CU -- 128 x 128 on two stages: 64 KB -- take twice the cap: the
          // share is meant in CUs)
          const int cap = a.max_wg / 8 * 8 * (lds * 2 <= 160 * 1024 ? 2 : 1);
          if (cap >= 8 && a.batch == 1 && (int)grid.x > cap) {
            if constexpr (BM == 256) {
              // the 256-row tiles sit at the register limit (256 x 320: 104 fragment + 160 accumulator registers): the
              // walking loop spills, so the cap is kept by launching row blocks of at most `cap` tiles one after the other
              const int tn = cdiv(a.N, BN), rows = std::max(1, cap / tn) * BM;
              for (int m0 = 0; m0 < a.M; m0 += rows) {
                GemmArgs c = a;
                c.M = std::min(rows, a.M - m0);
                if (a.a_row_map) c.a_row_map = a.a_row_map + m0;  // (logical rows m0 ..: A itself stays)
                else c.A = reinterpret_cast<const T*>(a.A) + (int64_t)m0 * a.lda;
                if (a.row_map) {
                  c.row_map = a.row_map + m0;  // C / resid are indexed by the mapped (destination) row
                } else {
                  c.C = a.c_f32 ? (void*)(reinterpret_cast<float*>(a.C) + (int64_t)m0 * a.ldc)
                                : (void*)(reinterpret_cast<T*>(a.C) + (int64_t)m0 * a.ldc);
                  if (a.resid) c.resid = a.resid + (int64_t)m0 * a.ldr;
                }
                const int tiles_m = cdiv(c.M, BM), nwg = tiles_m * tn;
                int gm = (int)lround(sqrt((double)(nwg > 8 ? nwg / 8 : 1)));
                c.group_m = gm < 1 ? 1 : (gm > tiles_m ? tiles_m : gm);
                hipLaunchKernelGGL(kern, dim3(nwg, 1, 1), dim3(WM * WN * 64), lds, s, c);
              }
              return;
            } else {
              auto kp = &gemm_glds_kernel<BM, BN, WM, WN, NS, false, true, T>;
              static KernelAttrOnce oncep;
              ensure_dyn_lds(oncep, reinterpret_cast<const void*>(kp), (int)lds);
              hipLaunchKernelGGL(kp, dim3(cap), dim3(WM * WN * 64), lds, s, a);
              return;
            }
          }
        }
        hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, s, a);
      };
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      using I3 = std::integral_constant<int, 3>;
      using I4 = std::integral_constant<int, 4>;
      using I64 = std::integral_constant<int, 64>;
      using I128 = std::integral_constant<int, 128>;
      using I256 = std::integral_constant<int, 256>;
      const int cus = device_cus();
      const int64_t t256 = (int64_t)cdiv(a.M, 256) * cdiv(a.N, 256) * a.batch;
      const double fill256 = (double)t256 / (double)(cdiv64(t256, cus) * cus);
      const int64_t t64w = (int64_t)cdiv(a.M, 64) * cdiv(a.N, 256) * a.batch;
      const int64_t t128 = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 128) * a.batch;
      using I320 = std::integral_constant<int, 320>;
      using I160 = std::integral_constant<int, 160>;
      const int64_t t160 = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 160) * a.batch;
      const double fill160 = (double)t160 / (double)(cdiv64(t160, cus) * cus);
      const double fill128 = (double)t128 / (double)(cdiv64(t128, cus) * cus);
      const int64_t t320 = (int64_t)cdiv(a.M, 256) * cdiv(a.N, 320) * a.batch;
      const double fill320 = (double)t320 / (double)(cdiv64(t320, cus) * cus);
      if (a.M >= 1024 && knobs().force128 == 3) {
        go(I128(), I128(), I2(), I4(), I3(), "gemm_bf16_128x128s3");
        return;
      }
      if (a.M >= 1024 && knobs().force128 == 2) {
        go(I128(), I128(), I2(), I4(), I2(), "gemm_bf16_128x128g");
        return;
      }
      // 256^2 from 80 % fill of whole rounds of the chip (lab knob ANYREF_GEMM_FILL256; it was 0.85: prefill gate/up at four
      // sequences -- 1280 x 22016 x 4096, 430 tiles = 0.84 -- went to 1720 tiles of 128^2: 267 us against 230 us on 256^2)
      static const double fill256_min = getenv("ANYREF_GEMM_FILL256") ? atof(getenv("ANYREF_GEMM_FILL256")) : 0.80;
      if (a.M >= 1024 && fill256 >= fill256_min) {
        go(I256(), I256(), I2(), I4(), I2(), "gemm_bf16_256x256");
        return;
      }
      if (!a.w_fp8 && a.M >= 1024 && a.N % 320 == 0 && fill320 >= 0.95) {
        go(I256(), I320(), I2(), I4(), I2(), "gemm_bf16_256x320");  // SAM fc1: 16 x 16 tiles = one per CU
        return;
      }
      if (!a.w_fp8 && a.M >= 1024 && a.N % 160 == 0 && fill160 >= 0.95 && fill128 < 0.7) {
        go(I128(), I160(), I4(), I2(), I3(), "gemm_bf16_128x160s3");  // SAM fc2: 32 x 8 tiles = one per CU
        return;
      }
      if constexpr (!is_half16<T>::value) {  // the LLaMA prefill / CLIP shapes (bf16 only: not compiled for the f16 towers)
        if (!a.w_fp8 && knobs().m320 >= 0 && a.M > 192 && a.M <= 320 && a.N >= 16384 && a.batch == 1 && cdiv(a.N, 96) <= cus) {
          // prefill gate/up (320 x 22016 x 4096): every workgroup owns a weight panel outright (all of M in one
          // tile, 230 panels on 256 CUs) instead of five 64-row workgroups sharing one: 7 % faster from cold weights
          if (knobs().ns320 == 2) go(I320(), std::integral_constant<int, 96>(), I4(), I2(), I2(), "gemm_bf16_320x96");
          else go(I320(), std::integral_constant<int, 96>(), I4(), I2(), I3(), "gemm_bf16_320x96s3");
          return;
        }
        if (!a.w_fp8 && knobs().m320 >= 0 && a.M > 192 && a.M <= 320 && a.batch == 2 && a.N >= 8192 &&
            (int64_t)cdiv(a.N, 96) * 2 <= cus) {
          // prefill qkv as two K slices (128 panels x 2 = 256 workgroups, a panel per workgroup); the RoPE kernel adds them
          if (knobs().ns320 == 2) go(I320(), std::integral_constant<int, 96>(), I4(), I2(), I2(), "gemm_bf16_320x96");
          else go(I320(), std::integral_constant<int, 96>(), I4(), I2(), I3(), "gemm_bf16_320x96s3");
          return;
        }
        if (!a.w_fp8 && knobs().m320 >= 0 && a.M > 192 && a.M <= 320 && a.batch > 1 && (int64_t)cdiv(a.N, 64) * a.batch <= cus &&
            (int64_t)cdiv(a.N, 64) * a.batch * 4 >= cus * 3) {  // (CLIP fc2's 128 slabs stay on 128^2 tiles: 192 workgroups)
          // split-K slabs of prefill o_proj / down_proj: 64 column panels x 4 K slices = 256 workgroups, one round, all
          // of M per tile (o_proj 30.2 -> 26.9 us, down_proj 55.3 -> 49.1 us with the reduction; qkv on 320 x 48 /
          // 320 x 64 tiles measured 5 % slower than 64 x 256 and stays there)
          go(I320(), I64(), I4(), I2(), I3(), "gemm_bf16_320x64");
          return;
        }
        {
          // prefill o_proj / down_proj at FOUR sequences (1280 x 4096 x 4096 / 11008: C3's per-GPU shape): 320 tiles of 128^2 are 1.25
          // workgroups per CU -- 64 CUs carry two and set the launch's time (2 x 64 K tiles x 32 KB through one CU's L2 -> LDS
          // path: 63 / 158 us) -- while 4 x 64 tiles of 320 x 64 are exactly one per CU (64 K tiles x 49 KB)
          static const bool m320x4 = !(getenv("ANYREF_GEMM_M320X4") && atoi(getenv("ANYREF_GEMM_M320X4")) == 0);
          const int64_t t320x64 = (int64_t)cdiv(a.M, 320) * cdiv(a.N, 64);
          if (m320x4 && !a.w_fp8 && a.batch == 1 && a.M > 512 && a.M <= 2560 && a.K >= 2048 && t320x64 <= cus && t320x64 * 10 >= cus * 9 &&
              fill128 < 0.7) {
            go(I320(), I64(), I4(), I2(), I3(), "gemm_bf16_320x64");
            return;
          }
        }
        if (a.M <= 512 && a.N >= 8192) {
          if (t64w <= cus) go(I64(), I256(), I1(), I4(), I3(), "gemm_bf16_64x256s3");
          else go(I64(), I256(), I1(), I4(), I2(), "gemm_bf16_64x256");
          return;
        }
      }
      if constexpr (!is_half16<T>::value) {
        // CLIP qkv / fc1 (257 x 3072 / 4096 x 1024): 72 / 96 workgroups of 128^2 stream 32 KB per K tile each through L2 -> LDS
        // (~0.5 us x 16 K tiles + ~4 us of launch, first tile and epilogue); 64 x 128 tiles are 120 / 160 workgroups at 24 KB
        static const bool clip64 = !(getenv("ANYREF_GEMM_CLIP64") && atoi(getenv("ANYREF_GEMM_CLIP64")) == 0);
        if (clip64 && !a.w_fp8 && a.batch == 1 && a.M > 128 && a.M <= 320 && a.N >= 2048 && a.N < 8192 && a.K <= 2048 && t128 <= cus) {
          go(I64(), I128(), I1(), I4(), I3(), "gemm_bf16_64x128s3");
          return;
        }
      }
      if (t128 <= cus)
        go(I128(), I128(), I2(), I4(), I3(), "gemm_bf16_128x128s3");
      else
        go(I128(), I128(), I2(), I4(), I2(), "gemm_bf16_128x128g");
      return;
    }
  }
  if constexpr (SP) {
    throw std::runtime_error("gemm<sp16>: no register-staged fallback (unreachable)");
  } else {
  const char* tag = is_half16<T>::value ? (bm128 ? (bn == 128 ? "gemm_f16_128x128" : "gemm_f16_128x64")
                                                 : (bn == 128 ? "gemm_f16_64x128" : "gemm_f16_64x64"))
                    : sizeof(T) == 2 ? (bm128 ? (bn == 128 ? "gemm_bf16_128x128" : "gemm_bf16_128x64")
                                            : (bn == 128 ? "gemm_bf16_64x128" : "gemm_bf16_64x64"))
                                   : (bm128 ? (bn == 128 ? "gemm_f32_128x128" : "gemm_f32_128x64")
                                            : (bn == 128 ? "gemm_f32_64x128" : "gemm_f32_64x64"));
  ProfScope prof(tag, flops, bytes, s);
  if (bm128 && bn == 128) {
    dim3 grid(cdiv(a.N, 128) * cdiv(a.M, 128), 1, a.batch);
    hipLaunchKernelGGL((gemm_kernel<T, 128, 128, BK>), grid, block, 0, s, a);
  } else if (bm128) {
    dim3 grid(cdiv(a.N, 64) * cdiv(a.M, 128), 1, a.batch);
    hipLaunchKernelGGL((gemm_kernel<T, 128, 64, BK>), grid, block, 0, s, a);
  } else if (bn == 128) {
    dim3 grid(cdiv(a.N, 128) * cdiv(a.M, 64), 1, a.batch);
    hipLaunchKernelGGL((gemm_kernel<T, 64, 128, BK>), grid, block, 0, s, a);
  } else {
    dim3 grid(cdiv(a.N, 64) * cdiv(a.M, 64), 1, a.batch);
    hipLaunchKernelGGL((gemm_kernel<T, 64, 64, BK>), grid, block, 0, s, a);
  }
  }
}

}  // namespace anyref

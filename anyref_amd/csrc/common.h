// Shared device/host helpers for the gfx950 kernels.  Wavefront = 64 everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <stdio.h>
#include <stdexcept>

namespace anyref {

// ---- storage types -------------------------------------------------------------------------
// bf16 is carried as raw 16-bit words; conversion by the hardware cvt (keeps NaN a NaN,
// MI355X_MICROARCH "Correctness boundaries").
struct bf16 {
  uint16_t x;
};

__host__ __device__ inline float bf2f(bf16 h) {
  uint32_t u = ((uint32_t)h.x) << 16;
  return __builtin_bit_cast(float, u);
}
__device__ inline bf16 f2bf(float f) {
  __bf16 h = (__bf16)f;
  bf16 r;
  r.x = __builtin_bit_cast(uint16_t, h);
  return r;
}
inline bf16 f2bf_host(float f) {  // RNE, host side (weight packing of small tables)
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) {
    bf16 r;
    r.x = (uint16_t)((u >> 16) | 0x40);
    return r;
  }
  u += 0x7fffu + ((u >> 16) & 1u);
  bf16 r;
  r.x = (uint16_t)(u >> 16);
  return r;
}

// IEEE half as raw 16-bit words: the storage type of the SAM image encoder in the perf build (3 more mantissa bits
// than bf16 at the same MFMA rate and bytes; the reference itself runs the tower in fp16, eval_referseg.py:70-72).
struct f16 {
  uint16_t x;
};
__device__ inline float h2f(f16 h) { return (float)__builtin_bit_cast(_Float16, h.x); }
__device__ inline f16 f2h(float f) {  // v_cvt_f16_f32, round to nearest even
  f16 r;
  r.x = __builtin_bit_cast(uint16_t, (_Float16)f);
  return r;
}
// 16-bit storage types at a glance (what the MFMA wrappers and the traits below key on)
template <typename T>
struct is_half16 {
  static constexpr bool value = false;
};
template <>
struct is_half16<f16> {
  static constexpr bool value = true;
};

template <typename T>
__device__ inline float to_f32(T v);
template <>
__device__ inline float to_f32<float>(float v) {
  return v;
}
template <>
__device__ inline float to_f32<bf16>(bf16 v) {
  return bf2f(v);
}
template <>
__device__ inline float to_f32<f16>(f16 v) {
  return h2f(v);
}
template <typename T>
__device__ inline T from_f32(float v);
template <>
__device__ inline f16 from_f32<f16>(float v) {
  return f2h(v);
}
template <>
__device__ inline float from_f32<float>(float v) {
  return v;
}
template <>
__device__ inline bf16 from_f32<bf16>(float v) {
  return f2bf(v);
}
// four consecutive outputs as ONE store (p: 8-byte aligned for bf16, 16-byte for f32); same rounding as from_f32
template <typename T>
__device__ inline void store4_from_f32(T* p, float a, float b, float c, float d);
template <>
__device__ inline void store4_from_f32<float>(float* p, float a, float b, float c, float d) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  *reinterpret_cast<f4*>(p) = f4{a, b, c, d};
}
template <>
__device__ inline void store4_from_f32<bf16>(bf16* p, float a, float b, float c, float d) {
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  *reinterpret_cast<u2*>(p) = u2{(uint32_t)f2bf(a).x | ((uint32_t)f2bf(b).x << 16), (uint32_t)f2bf(c).x | ((uint32_t)f2bf(d).x << 16)};
}

template <>
__device__ inline void store4_from_f32<f16>(f16* p, float a, float b, float c, float d) {
  typedef uint32_t u2 __attribute__((ext_vector_type(2)));
  *reinterpret_cast<u2*>(p) = u2{(uint32_t)f2h(a).x | ((uint32_t)f2h(b).x << 16), (uint32_t)f2h(c).x | ((uint32_t)f2h(d).x << 16)};
}
// two floats -> one packed pair of T (16-bit T)
template <typename T>
__device__ inline uint32_t pack2_from_f32(float lo, float hi) {
  return (uint32_t)from_f32<T>(lo).x | ((uint32_t)from_f32<T>(hi).x << 16);
}

// ---- split-pair activations (ANYREF_MODE_PARITY16) ---------------------------------------------
// An f32 activation a is carried as TWO bf16 terms, hi = bf16(a) and lo = bf16(a - hi): |a - hi - lo| <= 2^-18 |a|
// (bf16 alone: 2^-9).  A bf16 weight times each term is exact in the MFMA's f32 accumulator, so a GEMM over the pair
// is the f32 product of the activation with the (exactly stored) 16-bit weight to ~4e-6 relative -- at two 16-bit MFMA
// passes instead of the f32 MFMA's 1/16 rate, and with the weights never widened in HBM.
// Layout of a [rows, K] sp16 matrix (K % 64 == 0, 4 bytes per element like f32): every row is K / 64 blocks of
// [64 x hi | 64 x lo] bf16 -- exactly the 64-wide K tiles of the LDS-DMA GEMM, which walks 2K/64 tiles of A against
// K/64 tiles of W (W tile index = A tile index >> 1): no kernel-side conversion, no second weight copy.
struct sp16 {
  uint32_t x;
};
template <typename T>
struct is_split {
  static constexpr bool value = false;
};
template <>
struct is_split<sp16> {
  static constexpr bool value = true;
};
// bf16 index (inside the row) of the hi term of logical column c; the lo term sits 64 elements further
__host__ __device__ inline int sp_col(int c) { return ((c >> 6) << 7) + (c & 63); }
__device__ inline void sp_split(float v, uint16_t& hi, uint16_t& lo) {
  const bf16 h = f2bf(v);
  hi = h.x;
  lo = f2bf(v - bf2f(h)).x;
}
__device__ inline float sp_load(const sp16* row, int c) {
  const bf16* p = reinterpret_cast<const bf16*>(row) + sp_col(c);
  return bf2f(p[0]) + bf2f(p[64]);
}

// Typed row stores: `row` points at the first element of a matrix row, c is the logical column.
template <typename T>
__device__ inline void st1(T* row, int c, float v) {
  if constexpr (is_split<T>::value) {
    uint16_t* p = reinterpret_cast<uint16_t*>(row) + sp_col(c);
    uint16_t h, l;
    sp_split(v, h, l);
    p[0] = h;
    p[64] = l;
  } else {
    row[c] = from_f32<T>(v);
  }
}
template <typename T>
__device__ inline void st2(T* row, int c, float a, float b) {  // c % 2 == 0
  if constexpr (is_split<T>::value) {
    uint16_t* p = reinterpret_cast<uint16_t*>(row) + sp_col(c);
    uint16_t h0, l0, h1, l1;
    sp_split(a, h0, l0);
    sp_split(b, h1, l1);
    *reinterpret_cast<uint32_t*>(p) = (uint32_t)h0 | ((uint32_t)h1 << 16);
    *reinterpret_cast<uint32_t*>(p + 64) = (uint32_t)l0 | ((uint32_t)l1 << 16);
  } else if constexpr (sizeof(T) == 2) {
    *reinterpret_cast<uint32_t*>(row + c) = pack2_from_f32<T>(a, b);
  } else {
    row[c] = from_f32<T>(a);
    row[c + 1] = from_f32<T>(b);
  }
}
template <typename T>
__device__ inline void st4(T* row, int c, float a, float b, float cc, float d) {  // c % 4 == 0, aligned rows
  if constexpr (is_split<T>::value) {
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    uint16_t* p = reinterpret_cast<uint16_t*>(row) + sp_col(c);
    uint16_t h[4], l[4];
    sp_split(a, h[0], l[0]);
    sp_split(b, h[1], l[1]);
    sp_split(cc, h[2], l[2]);
    sp_split(d, h[3], l[3]);
    *reinterpret_cast<u2*>(p) = u2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
    *reinterpret_cast<u2*>(p + 64) = u2{(uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16)};
  } else {
    store4_from_f32<T>(row + c, a, b, cc, d);
  }
}

// ---- vector types --------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) short short8;   // 8 x bf16 = one MFMA A/B fragment
typedef __attribute__((ext_vector_type(4))) float float4v;  // MFMA 16x16 accumulator
typedef __attribute__((ext_vector_type(4))) uint32_t uint4v;
typedef __attribute__((ext_vector_type(2))) uint32_t uint2v;

// 16-byte vector of T -> floats.  NB: __builtin_bit_cast applied directly to an ext_vector element
// (v[i]) reads element 0 for every i on hipcc/ROCm 7.2; always go through a scalar temporary.
template <typename T>
struct Vec16;
template <>
struct Vec16<bf16> {
  static constexpr int N = 8;
  static __device__ inline void unpack(const uint4v& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t u = v[i];
      f[2 * i] = __builtin_bit_cast(float, u << 16);
      f[2 * i + 1] = __builtin_bit_cast(float, u & 0xffff0000u);
    }
  }
};
template <>
struct Vec16<f16> {
  static constexpr int N = 8;
  static __device__ inline void unpack(const uint4v& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t u = v[i];
      f[2 * i] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xffffu));
      f[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16));
    }
  }
};
template <>
struct Vec16<float> {
  static constexpr int N = 4;
  static __device__ inline void unpack(const uint4v& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t u = v[i];
      f[i] = __builtin_bit_cast(float, u);
    }
  }
};

// ---- 16-bit MFMA by storage type (same rate, same fragment / accumulator maps for bf16 and f16) ----
typedef __attribute__((ext_vector_type(4))) short short4v_;
typedef __attribute__((ext_vector_type(8))) _Float16 half8v;
typedef __attribute__((ext_vector_type(4))) _Float16 half4v;
template <typename T>
__device__ __forceinline__ float4v mfma_16x16x32(short8 a, short8 b, float4v c) {
  if constexpr (is_half16<T>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8v, a), __builtin_bit_cast(half8v, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <typename T>
__device__ __forceinline__ float4v mfma_16x16x16(short4v_ a, short4v_ b, float4v c) {
  if constexpr (is_half16<T>::value)
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4v, a), __builtin_bit_cast(half4v, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

// ---- fp8 (OCP e4m3fn on gfx950) weight-only quantisation helpers ------------------------------
typedef float float2v __attribute__((ext_vector_type(2)));
// 16 bytes = 16 e4m3 values -> 16 floats (v_cvt_pk_f32_fp8: two per instruction, exact)
__device__ inline void unpack_fp8x16(const uint4v& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int w = (int)v[i];
    const float2v lo = __builtin_amdgcn_cvt_pk_f32_fp8(w, false);
    const float2v hi = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
    f[4 * i] = lo[0]; f[4 * i + 1] = lo[1]; f[4 * i + 2] = hi[0]; f[4 * i + 3] = hi[1];
  }
}
constexpr float FP8_E4M3_MAX = 448.f;

// ---- activations ---------------------------------------------------------------------------
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_QUICK_GELU = 3, ACT_SILU = 4 };

__device__ inline float apply_act(float v, int act) {
  switch (act) {
    case ACT_RELU:
      return v > 0.f ? v : 0.f;
    case ACT_GELU:  // exact erf form (torch.nn.GELU default; common.py:13-25, mask_decoder.py:53-63)
      return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    case ACT_QUICK_GELU:  // HF CLIP quick_gelu
      return v / (1.f + __expf(-1.702f * v));
    case ACT_SILU:
      return v / (1.f + __expf(-v));
    default:
      return v;
  }
}

// GELU for the bf16 (perf) path: erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7, far below the
// bf16 rounding of the result) in ~14 instructions; ocml's erff is ~3x that and cost ~20 us of the
// 85 us SAM fc1 GEMM epilogue (21 M activations per layer).  The f32 parity path keeps erff.
__device__ inline float gelu_fast(float v) {
  const float x = fabsf(v) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 1.f - p * t * __expf(-x * x);  // erf(|v| / sqrt 2)
  return 0.5f * v * (1.f + copysignf(e, v));
}

// ---- wave reductions -----------------------------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- host-side error plumbing ---------------------------------------------------------------
#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      char _b[512];                                                                     \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
               __LINE__);                                                               \
      throw std::runtime_error(_b);                                                     \
    }                                                                                   \
  } while (0)

// ---- per-device launcher state ---------------------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE, and launchers are entered from any thread: a host that
// keeps one handle per GPU in ONE process (or two handles on one GPU from two threads) must not find the attribute "already
// set" because another device's first launch set a process-wide flag.  One bit per device ordinal per call site; two
// threads racing on the first launch both set the (idempotent) attribute.
struct KernelAttrOnce {
  std::atomic<uint64_t> done[4] = {};  // 256 device ordinals
};
inline void ensure_dyn_lds(KernelAttrOnce& st, const void* kernel, int bytes) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::atomic<uint64_t>& word = st.done[(dev >> 6) & 3];
  const uint64_t bit = 1ull << (dev & 63);
  if (word.load(std::memory_order_acquire) & bit) return;
  HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  word.fetch_or(bit, std::memory_order_release);
}
// CU count of the calling thread's current device (cached per device ordinal)
inline int device_cus() {
  static std::atomic<int> cache[256] = {};
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::atomic<int>& c = cache[dev & 255];
  int v = c.load(std::memory_order_relaxed);
  if (!v) {
    HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    c.store(v, std::memory_order_relaxed);
  }
  return v;
}

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return cdiv(a, b) * b; }

}  // namespace anyref

// Fused multi-head attention for gfx950 (flash-style, online softmax, scores never leave the CU).
//
// One source serves every attention on the path through strides + template head dim:
//   CLIP (257 tok, hd 64), LLaMA causal prefill + KV-cache decode (hd 128), SAM windowed /
//   global with decomposed rel-pos bias (hd 80/64), the mask decoder's 6x4096 / 4096x6 / 6x6
//   cross/self attention (hd 16/32).
// T = bf16 -> v_mfma_f32_16x16x32_bf16, T = float -> v_mfma_f32_16x16x4_f32 (parity mode).
//
// Workgroup = 4 waves = 64 query rows (16 per wave, Q fragments live in registers); K/V tiles of
// BKV keys go through LDS (V transposed for the bf16 B-operand), S = QK^T and O += PV on MFMA,
// softmax statistics per row in registers (rows sit on 16-lane groups of the C layout).
#include <cstdlib>
#include <type_traits>

#include "decode_attn.h"
#include <mutex>

#include "kernels.h"

#ifndef ANYREF_ATTN_PAIRED_V
#define ANYREF_ATTN_PAIRED_V 1
#endif

namespace anyref {

template <typename T>
struct AMma;
template <>
struct AMma<bf16> {
  static constexpr int KS = 32, VEC = 8;
  using Frag = short8;
  static __device__ inline Frag lds(const bf16* p, int lane) {
    return *reinterpret_cast<const short8*>(p + 8 * (lane >> 4));
  }
  // fragment straight from a global row (guarded): elements [k0 + 8*(lane>>4), +8)
  static __device__ inline Frag glb(const bf16* row, int k0, int lane, bool ok, int HD) {
    const int d = k0 + 8 * (lane >> 4);
    if (ok && d < HD) return *reinterpret_cast<const short8*>(row + d);
    return short8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  static __device__ inline float4v mma(Frag a, Frag b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ inline float fexp(float x) { return __expf(x); }
};
template <>
struct AMma<f16> {
  static constexpr int KS = 32, VEC = 8;
  using Frag = short8;
  static __device__ inline Frag lds(const f16* p, int lane) {
    return *reinterpret_cast<const short8*>(p + 8 * (lane >> 4));
  }
  static __device__ inline Frag glb(const f16* row, int k0, int lane, bool ok, int HD) {
    const int d = k0 + 8 * (lane >> 4);
    if (ok && d < HD) return *reinterpret_cast<const short8*>(row + d);
    return short8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  static __device__ inline float4v mma(Frag a, Frag b, float4v c) { return mfma_16x16x32<f16>(a, b, c); }
  static __device__ inline float fexp(float x) { return __expf(x); }
};
template <>
struct AMma<float> {
  static constexpr int KS = 4, VEC = 4;
  using Frag = float;
  static __device__ inline Frag lds(const float* p, int lane) { return p[lane >> 4]; }
  static __device__ inline Frag glb(const float* row, int k0, int lane, bool ok, int HD) {
    const int d = k0 + (lane >> 4);
    return (ok && d < HD) ? row[d] : 0.f;
  }
  static __device__ inline float4v mma(Frag a, Frag b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ inline float fexp(float x) { return expf(x); }
};

template <typename T>
struct AttnTile {
  static constexpr int BKV = sizeof(T) == 2 ? 64 : 32;
};

// ---------------------------------------------------------------------------------------------
// Transposed formulation.  Every wave owns 16 queries and computes, per K/V tile,
//     S^T = K Q^T          (16x16 tiles: key on the accumulator ROW, query on the lane COLUMN)
//     P^T = softmax columns (each lane owns ONE query: max / sum are lane-local over the tile's
//                            keys + two xor-shuffles across the 4 lane groups)
//     O^T += V^T P^T       (bf16: v_mfma_f32_16x16x16_bf16, whose B operand layout B[k = 4g+j][n]
//                            IS the accumulator layout of S^T, so P never leaves registers;
//                            f32: v_mfma_f32_16x16x4_f32 with k-slot g <- key 4g+r)
// so there is no P round trip through LDS, no per-row replication of the softmax statistics, and
// the output leaves as 4 consecutive head-dim values per lane.
// V^T fragments (A operand, A[d][key 4g+j]) come from the ROW-major V tile through
// ds_read_b64_tr_b16 (guide T10): per 16-lane group a 4-row x 16-column block is transposed by the
// LDS read itself; lane 4q+p supplies the address of row q, columns 4p..4p+3 and lane i receives
// column i of the 4 rows.  K and V are staged with plain 16-byte row stores.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short short4v;

// DB transposed 4x16 blocks (consecutive 16-column blocks of the same 4 rows) in ONE asm statement
// with their wait (guide §5.7 form (i)); EXEC is all ones (only wave-uniform control flow above).
template <int DB>
__device__ inline void lds_tr_blocks(uint32_t addr, uint2v (&f)[DB]);
template <>
__device__ inline void lds_tr_blocks<1>(uint32_t addr, uint2v (&f)[1]) {
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(f[0]) : "v"(addr) : "memory");
}
template <>
__device__ inline void lds_tr_blocks<2>(uint32_t addr, uint2v (&f)[2]) {
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:32\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(f[0]), "=&v"(f[1]) : "v"(addr) : "memory");
}
template <>
__device__ inline void lds_tr_blocks<4>(uint32_t addr, uint2v (&f)[4]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:32\n\t"
      "ds_read_b64_tr_b16 %2, %4 offset:64\n\tds_read_b64_tr_b16 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]) : "v"(addr) : "memory");
}
template <>
__device__ inline void lds_tr_blocks<5>(uint32_t addr, uint2v (&f)[5]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %5\n\tds_read_b64_tr_b16 %1, %5 offset:32\n\t"
      "ds_read_b64_tr_b16 %2, %5 offset:64\n\tds_read_b64_tr_b16 %3, %5 offset:96\n\t"
      "ds_read_b64_tr_b16 %4, %5 offset:128\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]) : "v"(addr) : "memory");
}
template <>
__device__ inline void lds_tr_blocks<8>(uint32_t addr, uint2v (&f)[8]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:32\n\t"
      "ds_read_b64_tr_b16 %2, %8 offset:64\n\tds_read_b64_tr_b16 %3, %8 offset:96\n\t"
      "ds_read_b64_tr_b16 %4, %8 offset:128\n\tds_read_b64_tr_b16 %5, %8 offset:160\n\t"
      "ds_read_b64_tr_b16 %6, %8 offset:192\n\tds_read_b64_tr_b16 %7, %8 offset:224\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]), "=&v"(f[7])
      : "v"(addr) : "memory");
}

// Two key blocks' worth of transposed V reads (2 x DB) behind ONE wait: halves the exposed LDS round trips of the P V
// phase (each wave used to serialise NB x (DB reads -> wait -> DB MFMAs) per tile)
template <int DB>
__device__ inline void lds_tr_blocks2(uint32_t a0, uint32_t a1, uint2v (&f0)[DB], uint2v (&f1)[DB]);
template <>
__device__ inline void lds_tr_blocks2<4>(uint32_t a0, uint32_t a1, uint2v (&f0)[4], uint2v (&f1)[4]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:32\n\t"
      "ds_read_b64_tr_b16 %2, %8 offset:64\n\tds_read_b64_tr_b16 %3, %8 offset:96\n\t"
      "ds_read_b64_tr_b16 %4, %9\n\tds_read_b64_tr_b16 %5, %9 offset:32\n\t"
      "ds_read_b64_tr_b16 %6, %9 offset:64\n\tds_read_b64_tr_b16 %7, %9 offset:96\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(f0[0]), "=&v"(f0[1]), "=&v"(f0[2]), "=&v"(f0[3]), "=&v"(f1[0]), "=&v"(f1[1]), "=&v"(f1[2]), "=&v"(f1[3])
      : "v"(a0), "v"(a1) : "memory");
}
template <>
__device__ inline void lds_tr_blocks2<5>(uint32_t a0, uint32_t a1, uint2v (&f0)[5], uint2v (&f1)[5]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %10\n\tds_read_b64_tr_b16 %1, %10 offset:32\n\t"
      "ds_read_b64_tr_b16 %2, %10 offset:64\n\tds_read_b64_tr_b16 %3, %10 offset:96\n\t"
      "ds_read_b64_tr_b16 %4, %10 offset:128\n\t"
      "ds_read_b64_tr_b16 %5, %11\n\tds_read_b64_tr_b16 %6, %11 offset:32\n\t"
      "ds_read_b64_tr_b16 %7, %11 offset:64\n\tds_read_b64_tr_b16 %8, %11 offset:96\n\t"
      "ds_read_b64_tr_b16 %9, %11 offset:128\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(f0[0]), "=&v"(f0[1]), "=&v"(f0[2]), "=&v"(f0[3]), "=&v"(f0[4]), "=&v"(f1[0]), "=&v"(f1[1]), "=&v"(f1[2]),
        "=&v"(f1[3]), "=&v"(f1[4])
      : "v"(a0), "v"(a1) : "memory");
}
template <>
__device__ inline void lds_tr_blocks2<8>(uint32_t a0, uint32_t a1, uint2v (&f0)[8], uint2v (&f1)[8]) {
  asm volatile(
      "ds_read_b64_tr_b16 %0, %16\n\tds_read_b64_tr_b16 %1, %16 offset:32\n\t"
      "ds_read_b64_tr_b16 %2, %16 offset:64\n\tds_read_b64_tr_b16 %3, %16 offset:96\n\t"
      "ds_read_b64_tr_b16 %4, %16 offset:128\n\tds_read_b64_tr_b16 %5, %16 offset:160\n\t"
      "ds_read_b64_tr_b16 %6, %16 offset:192\n\tds_read_b64_tr_b16 %7, %16 offset:224\n\t"
      "ds_read_b64_tr_b16 %8, %17\n\tds_read_b64_tr_b16 %9, %17 offset:32\n\t"
      "ds_read_b64_tr_b16 %10, %17 offset:64\n\tds_read_b64_tr_b16 %11, %17 offset:96\n\t"
      "ds_read_b64_tr_b16 %12, %17 offset:128\n\tds_read_b64_tr_b16 %13, %17 offset:160\n\t"
      "ds_read_b64_tr_b16 %14, %17 offset:192\n\tds_read_b64_tr_b16 %15, %17 offset:224\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(f0[0]), "=&v"(f0[1]), "=&v"(f0[2]), "=&v"(f0[3]), "=&v"(f0[4]), "=&v"(f0[5]), "=&v"(f0[6]), "=&v"(f0[7]),
        "=&v"(f1[0]), "=&v"(f1[1]), "=&v"(f1[2]), "=&v"(f1[3]), "=&v"(f1[4]), "=&v"(f1[5]), "=&v"(f1[6]), "=&v"(f1[7])
      : "v"(a0), "v"(a1) : "memory");
}

__device__ inline uint32_t pack_bf16x2(float lo, float hi) { return (uint32_t)f2bf(lo).x | ((uint32_t)f2bf(hi).x << 16); }

// NWV waves of 16 queries each (BQ = 16 NWV queries per workgroup) and BKV keys per tile.  Default 4 x 64;
// 7 x 80 for the 14 x 14 SAM windows (196 tokens: 2 query blocks x 3 key tiles per window-head where 4 x 64
// needs 4 x 4 with the last of each nearly empty): 80 -> 73 us per window layer.  Measured alternatives:
// 5 x 80 87 us, 4 x 80 123 us, 8 x 80 75 us, 7 x 64 78 us, 8 x 64 80 us (register-limited occupancy decides).
// NRES > 0: ALL keys stay resident in LDS as NRES tiles of BKV (Sk <= NRES * BKV, one query block covers Sq): the
// SAM windows again -- one workgroup per (window, head) loads K and V exactly once, a single barrier, then every
// wave walks the resident tiles on its own (the streaming form ran two query blocks per window-head, each
// re-loading K/V through three barrier-separated tiles at one workgroup per CU: 73 us per window layer).
template <typename T, int HD, int NWV, int BKV_, int NRES>
__device__ __forceinline__ void attn_body(const AttnArgs& a, const int bx, const int by, const int bz, char* smem) {
  constexpr int NT = NWV * 64;
  using M_ = AMma<T>;
  constexpr int KS = M_::KS, VEC = M_::VEC;
  constexpr bool BF = sizeof(T) == 2;
  constexpr int BKV = BKV_ > 0 ? BKV_ : AttnTile<T>::BKV;
  constexpr int BQ = 16 * NWV;
  constexpr int HDK = (HD + KS - 1) / KS * KS;  // QK^T contraction length (zero padded)
  constexpr int LDK = HDK + VEC;                // K tile row stride
  // V tile row stride (row-major [key][d]).  bf16: the transposed reads (ds_read_b64_tr_b16) fetch, per
  // 16-lane group, 4 rows x 32 B; a stride of 32 B modulo the 256-B bank row lets 8 consecutive rows tile the
  // banks (HD + 8 elements left them overlapping: 2x more conflict cycles than LDS-active cycles measured)
  constexpr int LDV = BF ? ((HD * 2 + 255) / 256 * 256 + 32) / 2 : HD + VEC;
  constexpr int NB = BKV / 16;                  // key blocks of one tile
  constexpr int DB = HD / 16;                   // output d blocks
  static_assert(HD % 16 == 0, "head dim must be a multiple of 16");

  constexpr int NTILE = NRES > 0 ? NRES : 1;  // tiles held in LDS
  T* Ks = reinterpret_cast<T*>(smem);
  T* Vs = Ks + NTILE * BKV * LDK;
  float* relh_s = reinterpret_cast<float*>(Vs + NTILE * BKV * LDV);
  float* relw_s = relh_s + BQ * a.kh;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = lane & 15, g = lane >> 4;  // this lane's query (column) and lane group
  // kv_splits > 1 (few queries, many keys: the mask decoder's token -> image attention is 7 x 4096 on 8 heads =
  // 8 workgroups walking 128 tiles each): blockIdx.x = q-block * splits + split, every split writes an
  // un-normalised partial (O, m, l) and attn_combine_kernel merges them
  const int split = bx % a.kv_splits;
  const int b = bz, h = by, q0 = (bx / a.kv_splits) * BQ;
  const int kv_len = a.kv_len ? a.kv_len[b] : a.Sk;
  const int q_len = a.q_len ? a.q_len[b] : a.Sq;
  if (q0 >= q_len) return;  // uniform per workgroup
  const int pos0 = a.q_pos0 ? a.q_pos0[b] : 0;
  int kv_end = kv_len;
  if (a.causal) {
    const int imax = (q0 + BQ < q_len ? q0 + BQ : q_len) - 1;
    if (pos0 + imax + 1 < kv_end) kv_end = pos0 + imax + 1;
  }

  const T* Qb = reinterpret_cast<const T*>(a.Q) + (int64_t)b * a.q_bs + (int64_t)h * a.q_hs;
  const T* Kb = reinterpret_cast<const T*>(a.K) + (int64_t)b * a.k_bs + (int64_t)h * a.k_hs;
  const T* Vb = reinterpret_cast<const T*>(a.V) + (int64_t)b * a.v_bs + (int64_t)h * a.v_hs;

  // Q fragments, used as the B operand of S^T = K Q^T: B[k = d][n = query lane&15]
  const int il = wave * 16 + qi, iq = q0 + il;  // local / global query of this lane
  const bool q_ok = iq < q_len;
  typename M_::Frag qf[HDK / KS];
  {
    const T* qrow = Qb + (int64_t)(q_ok ? iq : 0) * a.q_rs;
#pragma unroll
    for (int kk = 0; kk < HDK / KS; ++kk) qf[kk] = M_::glb(qrow, kk * KS, lane, q_ok, HD);
  }
  // bf16 path: softmax in the log2 domain (one v_exp_f32 per score): scores and biases carry log2(e)
  constexpr float LOG2E = 1.4426950408889634f;
  const float bsc = BF ? LOG2E : 1.f;
  bool rel_tab = false;  // tables given: bias computed here (resident form, bf16)
  if constexpr (NRES > 0 && BF) rel_tab = a.rel_tab_h != nullptr;
  const bool has_rel = a.rel_h != nullptr || a.rel_p != nullptr || rel_tab;
  // Row-padded key order (resident form with rel-pos tables, kw <= 16: the SAM windows): LDS key slot 16 r + c holds
  // key (r, c) of the kh x kw window (c >= kw: a zero row), so a 16-key block is exactly one bias row -- the kh term
  // is ONE value per block, the kw term four registers per lane for the whole kernel, and the padding masks itself
  // (bias -inf).  The general path below costs two LDS reads + a multiply-shift divide per score: with 13 waves on
  // 4 SIMDs that VALU work, not the MFMAs, set the kernel's time (47.5 us per window layer).
  bool rowpad = false;
  if constexpr (NRES > 0 && BF && BKV % 16 == 0) rowpad = rel_tab && a.kw <= 16 && a.kh * 16 <= NRES * BKV;
  if (rowpad) kv_end = a.kh * 16;
  T* Rs = reinterpret_cast<T*>(relw_s + BQ * a.kw);  // [2][32][LDK] zero-padded tables (rel_tab only)
  if constexpr (NRES > 0 && BF) {
    if (rel_tab) {
      for (int i = tid; i < BQ * (a.kh + a.kw); i += NT) relh_s[i] = 0.f;  // relw_s follows relh_s
      constexpr int RV = LDK / VEC;
      for (int i = tid; i < 2 * 32 * RV; i += NT) {
        const int t = i / (32 * RV), e = (i / RV) % 32, d = (i % RV) * VEC;
        const int ne = 2 * (t == 0 ? a.kh : a.kw) - 1;
        const T* tab = reinterpret_cast<const T*>(t == 0 ? a.rel_tab_h : a.rel_tab_w);
        const uint4v v = (e < ne && d < HD) ? *reinterpret_cast<const uint4v*>(tab + (int64_t)e * a.rel_tab_ld + d)
                                            : uint4v{0, 0, 0, 0};
        *reinterpret_cast<uint4v*>(&Rs[(t * 32 + e) * LDK + d]) = v;
      }
    }
  }
  if (rel_tab) {
    // filled after the K/V barrier below
  } else if (a.rel_p) {
    const float* P = a.rel_p + (int64_t)h * a.rel_hs + ((int64_t)b * a.Sq + q0) * a.rel_ld;
    const int np = a.rel_ld / 2;
    for (int i = tid; i < BQ * a.kh; i += NT) {
      const int r = i / a.kh, c = i % a.kh;
      const int y = (q0 + r) / a.kw;
      relh_s[i] = q0 + r < q_len ? P[(int64_t)r * a.rel_ld + (y - c + a.kh - 1)] * bsc : 0.f;
    }
    for (int i = tid; i < BQ * a.kw; i += NT) {
      const int r = i / a.kw, c = i % a.kw;
      const int x = (q0 + r) % a.kw;
      relw_s[i] = q0 + r < q_len ? P[(int64_t)r * a.rel_ld + np + (x - c + a.kw - 1)] * bsc : 0.f;
    }
  } else if (has_rel) {
    const float* rh = a.rel_h + ((int64_t)b * a.H + h) * a.Sq * a.kh;
    const float* rw = a.rel_w + ((int64_t)b * a.H + h) * a.Sq * a.kw;
    for (int i = tid; i < BQ * a.kh; i += NT) {
      const int r = i / a.kh, c = i % a.kh;
      relh_s[i] = q0 + r < q_len ? rh[(int64_t)(q0 + r) * a.kh + c] * bsc : 0.f;
    }
    for (int i = tid; i < BQ * a.kw; i += NT) {
      const int r = i / a.kw, c = i % a.kw;
      relw_s[i] = q0 + r < q_len ? rw[(int64_t)(q0 + r) * a.kw + c] * bsc : 0.f;
    }
  }
  // global-attention fast path: the key tile is exactly one bias row (kw == BKV, tiles aligned), so
  // the kw-term of this lane's 16 keys never changes and the kh-term is one value per tile
  const bool rel_fast = NRES == 0 && has_rel && a.kw == BKV;  // (compile-time off for the resident form: registers)
  // general path: j / kw by multiply-shift, exact for j < 4096 and kw <= 64 (error j / 2^20 < 1 / kw)
  const unsigned kw_magic = has_rel ? (1u << 20) / (unsigned)a.kw + 1u : 0u;
  float relw_reg[NB][4];
  if (rel_fast) {
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) relw_reg[nb][r] = relw_s[il * a.kw + nb * 16 + 4 * g + r];
  }

  float m_run = -INFINITY, l_run = 0.f;  // running max / (partial) sum of THIS lane's query
  float4v ot[DB];                        // O^T tiles: row = d (4g+r), col = query
#pragma unroll
  for (int d = 0; d < DB; ++d) ot[d] = float4v{0.f, 0.f, 0.f, 0.f};

  // K [BKV][HDK] and V [BKV][HD] tiles, both row-major, 16-byte vectors.  Register-staged with the loads of
  // tile t+1 issued BEFORE tile t is multiplied and written to LDS after the barrier that frees it (guide
  // T14): staged synchronously, the waves spent 65 % of their cycles waiting on these loads (SQ_WAIT_ANY /
  // SQ_WAVE_CYCLES on the SAM global-attention launch).
  constexpr int KVEC = HDK / VEC, VVEC = HD / VEC;
  constexpr int KPT = (BKV * KVEC + NT - 1) / NT, VPT = (BKV * VVEC + NT - 1) / NT;  // vectors per thread
  uint4v kreg[KPT], vreg[VPT];
  auto gload_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int v = tid + i * NT, row = v / KVEC, d = (v % KVEC) * VEC;
      int j = kt + row;
      bool ok = j < kv_end;
      if (rowpad) {  // slot -> key (slot / 16, slot % 16) of the window
        const int jr = j >> 4, jc = j & 15;
        ok = jc < a.kw && jr < a.kh;
        j = jr * a.kw + jc;
      }
      kreg[i] = (v < BKV * KVEC && ok && d < HD) ? *reinterpret_cast<const uint4v*>(Kb + (int64_t)j * a.k_rs + d)
                                                 : uint4v{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int v = tid + i * NT, row = v / VVEC, d = (v % VVEC) * VEC;
      int j = kt + row;
      bool ok = j < kv_end;
      if (rowpad) {
        const int jr = j >> 4, jc = j & 15;
        ok = jc < a.kw && jr < a.kh;
        j = jr * a.kw + jc;
      }
      vreg[i] = (v < BKV * VVEC && ok) ? *reinterpret_cast<const uint4v*>(Vb + (int64_t)j * a.v_rs + d)
                                       : uint4v{0, 0, 0, 0};
    }
  };
  auto sstore_tile = [&](int slot = 0) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int v = tid + i * NT, row = slot * BKV + v / KVEC, d = (v % KVEC) * VEC;
      if (v < BKV * KVEC) *reinterpret_cast<uint4v*>(&Ks[row * LDK + d]) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int v = tid + i * NT, row = slot * BKV + v / VVEC, d = (v % VVEC) * VEC;
      if (v < BKV * VVEC) *reinterpret_cast<uint4v*>(&Vs[row * LDV + d]) = vreg[i];
    }
  };
  int kt_begin = 0;
  if (a.kv_splits > 1) {
    const int chunk = cdiv(cdiv(kv_end, a.kv_splits), BKV) * BKV;
    kt_begin = split * chunk;
    kv_end = kv_end < kt_begin + chunk ? kv_end : kt_begin + chunk;
  }
#ifndef ANYREF_ATTN_RES80_ONESHOT
#define ANYREF_ATTN_RES80_ONESHOT 0  // lab build flag: SAM windows with the CLIP form's one-shot request (36.9 vs 36.4 us: off)
#endif
  if constexpr (NRES > 0 && (HD == 64 || (HD == 80 && ANYREF_ATTN_RES80_ONESHOT))) {
    // short rows (CLIP): every resident tile requested before the first LDS store -- one round trip, not NRES
    uint4v kall[NRES][KPT], vall[NRES][VPT];
#pragma unroll
    for (int t = 0; t < NRES; ++t) {
      gload_tile(t * BKV);  // (tiles past kv_end load zeros)
#pragma unroll
      for (int i = 0; i < KPT; ++i) kall[t][i] = kreg[i];
#pragma unroll
      for (int i = 0; i < VPT; ++i) vall[t][i] = vreg[i];
    }
#pragma unroll
    for (int t = 0; t < NRES; ++t) {
#pragma unroll
      for (int i = 0; i < KPT; ++i) kreg[i] = kall[t][i];
#pragma unroll
      for (int i = 0; i < VPT; ++i) vreg[i] = vall[t][i];
      sstore_tile(t);
    }
    __syncthreads();
  } else if constexpr (NRES > 0) {
#pragma unroll
    for (int t = 0; t < NRES; ++t)
      if (t * BKV < kv_end) {
        gload_tile(t * BKV);
        sstore_tile(t);
      }
    __syncthreads();
  } else {
    if (kv_end > kt_begin) gload_tile(kt_begin);
  }
  if constexpr (NRES > 0) {
    if constexpr (BF) {
      if (rel_tab) {
        // P^T[entry][query] = R q^T for both tables: entry = 16 blk + 4 g + r, query = this lane's column; the
        // reference's shifted gather (get_rel_pos) becomes a scatter: entry e of the h table is the bias of key
        // row y + kh - 1 - e, where y is the query's own row
        float4v ph[2], pw[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          ph[blk] = float4v{0.f, 0.f, 0.f, 0.f};
          pw[blk] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < HDK / KS; ++kk) {
            ph[blk] = M_::mma(M_::lds(&Rs[(blk * 16 + qi) * LDK + kk * KS], lane), qf[kk], ph[blk]);
            pw[blk] = M_::mma(M_::lds(&Rs[(32 + blk * 16 + qi) * LDK + kk * KS], lane), qf[kk], pw[blk]);
          }
        }
        if (q_ok) {
          const int y = iq / a.kw, x = iq - y * a.kw;
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int e = blk * 16 + 4 * g + r;
              const int ch = y + a.kh - 1 - e, cw = x + a.kw - 1 - e;
              if (ch >= 0 && ch < a.kh) relh_s[il * a.kh + ch] = ph[blk][r] * bsc;
              if (cw >= 0 && cw < a.kw) relw_s[il * a.kw + cw] = pw[blk][r] * bsc;
            }
        }
        __syncthreads();
      }
    }
  }
  float relw_pad[4] = {0.f, 0.f, 0.f, 0.f};  // rowpad: the kw term of this lane's four key columns (-inf: padding)
  if (rowpad) {
#pragma unroll
    for (int r = 0; r < 4; ++r) relw_pad[r] = 4 * g + r < a.kw ? relw_s[il * a.kw + 4 * g + r] : -INFINITY;
  }
  for (int kt = kt_begin; kt < kv_end; kt += BKV) {
    const int slot = NRES > 0 ? kt / BKV : 0;  // resident tile of this step
    const T* Kt = Ks + slot * BKV * LDK;
    if constexpr (NRES == 0) {
      sstore_tile();
      __syncthreads();
      if (kt + BKV < kv_end) gload_tile(kt + BKV);
    }

    // wave-uniform skips: rows past the end / tile entirely in the causal future of this row block
    const bool active = q0 + wave * 16 < q_len && !(a.causal && kt > pos0 + q0 + wave * 16 + 15);
    if (active) {
      // ---- S^T = K Q^T ------------------------------------------------------------------------
      float4v st[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        st[nb] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < HDK / KS; ++kk)
          st[nb] = M_::mma(M_::lds(&Kt[(nb * 16 + qi) * LDK + kk * KS], lane), qf[kk], st[nb]);
      }
      if constexpr (!BF) {
        // hipcc/ROCm 7.2 under-pads the VALU read of a v_mfma_f32_16x16x4_f32 result on gfx950
        // (40-cycle dependent latency): the LAST accumulator register was read before the final
        // k-step had landed (every 4th key/row slightly wrong).  Pad before the first VALU use.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15");
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- scale, bias, mask; column softmax of this lane's query -------------------------------
      float sv[NB][4];
      float mx = -INFINITY;
      const float relh_tile = rel_fast ? relh_s[il * a.kh + kt / BKV] : 0.f;
      const float c1 = a.scale * bsc;
      // masking only where a tile can hold an invalid key for one of this wave's 16 queries (wave-uniform)
      const bool need_mask = !rowpad && (kt + BKV > kv_len || (a.causal && kt + BKV - 1 > pos0 + q0 + wave * 16));
      if (rowpad) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int jh = (kt >> 4) + nb;  // this key block = window row jh
          const float bh = jh < a.kh ? relh_s[il * a.kh + jh] : -INFINITY;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float s = fmaf(st[nb][r], c1, bh + relw_pad[r]);
            sv[nb][r] = s;
            mx = fmaxf(mx, s);
          }
        }
      } else if (need_mask) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = kt + nb * 16 + 4 * g + r;
            float s = st[nb][r] * c1;
            if (rel_fast) {
              s += relh_tile + relw_reg[nb][r];
            } else if (has_rel) {
              int jh = (int)(((unsigned)j * kw_magic) >> 20), jw = j - jh * a.kw;
              if (jh >= a.kh) jh = 0, jw = 0;  // masked keys beyond kv_len: keep LDS reads in range
              s += relh_s[il * a.kh + jh] + relw_s[il * a.kw + jw];
            }
            const bool valid = j < kv_len && (!a.causal || j <= pos0 + iq);
            s = valid ? s : -INFINITY;
            sv[nb][r] = s;
            mx = fmaxf(mx, s);
          }
        }
      } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float s;
            if (rel_fast) {
              s = fmaf(st[nb][r], c1, relh_tile + relw_reg[nb][r]);
            } else if (has_rel) {
              const int j = kt + nb * 16 + 4 * g + r;
              const int jh = (int)(((unsigned)j * kw_magic) >> 20), jw = j - jh * a.kw;
              s = fmaf(st[nb][r], c1, relh_s[il * a.kh + jh] + relw_s[il * a.kw + jw]);
            } else {
              s = st[nb][r] * c1;
            }
            sv[nb][r] = s;
            mx = fmaxf(mx, s);
          }
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float rs = 0.f;
      if constexpr (BF) {
        // lazy rescaling: keep the old reference maximum unless the new one exceeds it by 2^8 (then P <= 256,
        // exactly representable scale in bf16 / f32), so O and l are rescaled in few tiles, and only when
        // some lane of the wave moved its reference (wave-uniform branch)
        const float m_new = mx > m_run + 8.f ? mx : m_run;  // m_run = -inf: any finite mx moves it
        const bool moved = m_new != m_run;
        const float mref = m_new == -INFINITY ? 0.f : m_new;
        if (__builtin_amdgcn_ballot_w64(moved) != 0) {
          const float alpha = __builtin_amdgcn_exp2f(m_run - mref);  // m_run = -inf -> 0; unmoved lanes: 2^0 = 1
          l_run *= alpha;
#pragma unroll
          for (int d = 0; d < DB; ++d) ot[d] *= alpha;
          m_run = m_new;
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sv[nb][r] = __builtin_amdgcn_exp2f(sv[nb][r] - mref);  // masked: 2^-inf = 0
            rs += sv[nb][r];
          }
        l_run += rs;
      } else {
        const float m_new = fmaxf(m_run, mx);
        const float mref = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = M_::fexp(m_run - mref);  // m_run = -inf -> 0
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sv[nb][r] = M_::fexp(sv[nb][r] - mref);  // masked: exp(-inf) = 0
            rs += sv[nb][r];
          }
        l_run = l_run * alpha + rs;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < DB; ++d) ot[d] *= alpha;
      }
      // ---- O^T += V^T P^T -----------------------------------------------------------------------
      if constexpr (BF) {
        // this lane's address inside its group's 4x16 block: row q = (lane&15)>>2, columns 4p
        const uint32_t vbase = (uint32_t)(reinterpret_cast<const char*>(Vs) - smem) +  // dynamic LDS starts at 0
                               (uint32_t)((slot * BKV + 4 * g + (qi >> 2)) * LDV + 4 * (qi & 3)) * 2u;
        // key blocks in pairs where the registers allow (not the 13-wave window form at its 128-VGPR limit)
        constexpr bool PAIRED = (DB == 4 || DB == 5 || DB == 8) && NB % 2 == 0 && NWV <= 8 && ANYREF_ATTN_PAIRED_V;
        if constexpr (PAIRED) {
#pragma unroll
          for (int nb = 0; nb < NB; nb += 2) {
            uint2v vt0[DB], vt1[DB];
            lds_tr_blocks2<DB>(vbase + (uint32_t)(nb * 16 * LDV * 2), vbase + (uint32_t)((nb + 1) * 16 * LDV * 2), vt0, vt1);
            __builtin_amdgcn_sched_barrier(0);
            const uint2v pb0 = uint2v{pack2_from_f32<T>(sv[nb][0], sv[nb][1]), pack2_from_f32<T>(sv[nb][2], sv[nb][3])};
            const uint2v pb1 = uint2v{pack2_from_f32<T>(sv[nb + 1][0], sv[nb + 1][1]), pack2_from_f32<T>(sv[nb + 1][2], sv[nb + 1][3])};
#pragma unroll
            for (int d = 0; d < DB; ++d)
              ot[d] = mfma_16x16x16<T>(__builtin_bit_cast(short4v, vt0[d]), __builtin_bit_cast(short4v, pb0), ot[d]);
#pragma unroll
            for (int d = 0; d < DB; ++d)
              ot[d] = mfma_16x16x16<T>(__builtin_bit_cast(short4v, vt1[d]), __builtin_bit_cast(short4v, pb1), ot[d]);
          }
        } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          uint2v vt[DB];
          lds_tr_blocks<DB>(vbase + (uint32_t)(nb * 16 * LDV * 2), vt);
          __builtin_amdgcn_sched_barrier(0);
          const uint2v pb = uint2v{pack2_from_f32<T>(sv[nb][0], sv[nb][1]), pack2_from_f32<T>(sv[nb][2], sv[nb][3])};
#pragma unroll
          for (int d = 0; d < DB; ++d)
            ot[d] = mfma_16x16x16<T>(__builtin_bit_cast(short4v, vt[d]), __builtin_bit_cast(short4v, pb), ot[d]);
        }
        }
      } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float* vrow = reinterpret_cast<const float*>(Vs) + (slot * BKV + nb * 16 + 4 * g + r) * LDV + qi;  // k-slot g <- key 4g+r
#pragma unroll
            for (int d = 0; d < DB; ++d) ot[d] = M_::mma(vrow[d * 16], sv[nb][r], ot[d]);
          }
      }
    }
    if constexpr (NRES == 0) __syncthreads();  // K/V tiles free for the next iteration
  }

  // ---- normalise + store: lane (query qi, group g) holds O[query][d = 16*db + 4g + r] ------------
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (a.kv_splits > 1) {
    if (q_ok) {
      const int64_t row = (((int64_t)split * a.B + b) * a.H + h) * a.Sq + iq;
      float* po = a.part_o + row * HD;
#pragma unroll
      for (int d = 0; d < DB; ++d) *reinterpret_cast<float4v*>(po + d * 16 + 4 * g) = ot[d];
      if (g == 0) {
        a.part_ml[row * 2] = m_run;
        a.part_ml[row * 2 + 1] = l_run;
      }
    }
    return;
  }
  if (q_ok) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    T* Ob = reinterpret_cast<T*>(a.O) + (int64_t)b * a.o_bs + (int64_t)h * a.o_hs + (int64_t)iq * a.o_rs;
    float* Obf = reinterpret_cast<float*>(a.O) + (int64_t)b * a.o_bs + (int64_t)h * a.o_hs + (int64_t)iq * a.o_rs;
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      const float4v v = ot[d] * inv;
      const int c = d * 16 + 4 * g;
      if (a.o_f32) {
        Obf[c] = v[0]; Obf[c + 1] = v[1]; Obf[c + 2] = v[2]; Obf[c + 3] = v[3];
      } else if constexpr (BF) {
        reinterpret_cast<uint16_t*>(Ob)[c] = from_f32<T>(v[0]).x;
        reinterpret_cast<uint16_t*>(Ob)[c + 1] = from_f32<T>(v[1]).x;
        reinterpret_cast<uint16_t*>(Ob)[c + 2] = from_f32<T>(v[2]).x;
        reinterpret_cast<uint16_t*>(Ob)[c + 3] = from_f32<T>(v[3]).x;
      } else if (a.o_split) {  // split-pair rows (the proj / o_proj GEMM's A operand in ANYREF_MODE_PARITY16)
        st4<sp16>(reinterpret_cast<sp16*>(a.O) + (int64_t)b * a.o_bs + (int64_t)iq * a.o_rs, (int)(h * a.o_hs) + c, v[0], v[1], v[2], v[3]);
      } else {
        Ob[c] = v[0]; Ob[c + 1] = v[1]; Ob[c + 2] = v[2]; Ob[c + 3] = v[3];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// SAM global attention (4096 x 4096 tokens, head dim 80, decomposed rel-pos bias from the P buffer), 16-bit only: the
// body above with TWO query blocks per wave (32 queries) -- every K fragment read from LDS feeds two MFMAs, every
// transposed V block two, the softmax of one block is independent work beside the other block's MFMAs (a wave with
// one block runs QK^T -> softmax -> P V as one dependent chain and its SIMD partner, aligned by the tile barriers, does
// the same phase at the same time: MFMA pipe 19 % busy, 48 % of the wave cycles parked), and the tile barriers /
// K / V staging are paid once per 32 queries.  The kw term of the bias (key tile = one row of the 64 x 64 grid) sits
// in registers for the whole kernel, loaded straight from P; the kh term is one LDS word per tile and block.
// Requirements (attn_launch): kw == 64 == BKV, Sk == kh * kw, Sq % (32 NWV) == 0, no masks, bias from rel_p.
// ---------------------------------------------------------------------------------------------
template <typename T, int HD, int NWV>
__device__ __forceinline__ void attn_g2_body(const AttnArgs& a, const int bx, const int by, const int bz, char* smem) {
  static_assert(sizeof(T) == 2 && HD == 80, "16-bit, head dim 80");
  using M_ = AMma<T>;
  constexpr int NT = NWV * 64, QB = 2, BKV = 64, BQ = 16 * QB * NWV;
  constexpr int KS = 32, VEC = 8, HDK = 96, LDK = HDK + VEC;
  constexpr int LDV = ((HD * 2 + 255) / 256 * 256 + 32) / 2;
  constexpr int NB = BKV / 16, DB = HD / 16;
  T* Ks = reinterpret_cast<T*>(smem);
  T* Vs = Ks + BKV * LDK;
  float* relh_s = reinterpret_cast<float*>(Vs + BKV * LDV);
  const int RH = a.kh + 1;  // row stride of relh_s (odd: the 16 queries of a lane group read 16 banks)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = lane & 15, g = lane >> 4;
  const int b = bz, h = by, q0 = bx * BQ;
  const T* Qb = reinterpret_cast<const T*>(a.Q) + (int64_t)b * a.q_bs + (int64_t)h * a.q_hs;
  const T* Kb = reinterpret_cast<const T*>(a.K) + (int64_t)b * a.k_bs + (int64_t)h * a.k_hs;
  const T* Vb = reinterpret_cast<const T*>(a.V) + (int64_t)b * a.v_bs + (int64_t)h * a.v_hs;
  constexpr float LOG2E = 1.4426950408889634f;
  const float* P = a.rel_p + (int64_t)h * a.rel_hs + ((int64_t)b * a.Sq + q0) * a.rel_ld;
  const int np = a.rel_ld / 2;

  int il[QB];
  short8 qf[QB][HDK / KS];
  float relw_reg[QB][NB][4];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    il[qb] = (wave * QB + qb) * 16 + qi;
    const T* qrow = Qb + (int64_t)(q0 + il[qb]) * a.q_rs;
#pragma unroll
    for (int kk = 0; kk < HDK / KS; ++kk) qf[qb][kk] = M_::glb(qrow, kk * KS, lane, true, HD);
    const int x = (q0 + il[qb]) % a.kw;
    const float* pw = P + (int64_t)il[qb] * a.rel_ld + np + x + a.kw - 1;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) relw_reg[qb][nb][r] = pw[-(nb * 16 + 4 * g + r)] * LOG2E;
  }
  for (int i = tid; i < BQ * a.kh; i += NT) {
    const int r = i / a.kh, c = i - r * a.kh;
    const int y = (q0 + r) / a.kw;
    relh_s[r * RH + c] = P[(int64_t)r * a.rel_ld + (y - c + a.kh - 1)] * LOG2E;
  }

  float m_run[QB], l_run[QB];
  float4v ot[QB][DB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -INFINITY;
    l_run[qb] = 0.f;
#pragma unroll
    for (int d = 0; d < DB; ++d) ot[qb][d] = float4v{0.f, 0.f, 0.f, 0.f};
  }

  constexpr int KVEC = HDK / VEC, VVEC = HD / VEC;
  constexpr int KPT = (BKV * KVEC + NT - 1) / NT, VPT = (BKV * VVEC + NT - 1) / NT;
  uint4v kreg[KPT], vreg[VPT];
  auto gload_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int v = tid + i * NT, row = v / KVEC, d = (v % KVEC) * VEC;
      kreg[i] = (v < BKV * KVEC && d < HD) ? *reinterpret_cast<const uint4v*>(Kb + (int64_t)(kt + row) * a.k_rs + d)
                                           : uint4v{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int v = tid + i * NT, row = v / VVEC, d = (v % VVEC) * VEC;
      vreg[i] = v < BKV * VVEC ? *reinterpret_cast<const uint4v*>(Vb + (int64_t)(kt + row) * a.v_rs + d) : uint4v{0, 0, 0, 0};
    }
  };
  auto sstore_tile = [&]() {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int v = tid + i * NT, row = v / KVEC, d = (v % KVEC) * VEC;
      if (v < BKV * KVEC) *reinterpret_cast<uint4v*>(&Ks[row * LDK + d]) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int v = tid + i * NT, row = v / VVEC, d = (v % VVEC) * VEC;
      if (v < BKV * VVEC) *reinterpret_cast<uint4v*>(&Vs[row * LDV + d]) = vreg[i];
    }
  };
  gload_tile(0);
  const float c1 = a.scale * LOG2E;
  const uint32_t vbase = (uint32_t)(reinterpret_cast<const char*>(Vs) - smem) +
                         (uint32_t)((4 * g + (qi >> 2)) * LDV + 4 * (qi & 3)) * 2u;
  for (int kt = 0; kt < a.Sk; kt += BKV) {
    sstore_tile();
    __syncthreads();  // (also orders the relh_s fill before its first read)
    if (kt + BKV < a.Sk) gload_tile(kt + BKV);
    // ---- S^T = K Q^T for both blocks: one K fragment read, two MFMAs ----
    float4v st[QB][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) st[qb][nb] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < HDK / KS; ++kk) {
        const short8 kf = M_::lds(&Ks[(nb * 16 + qi) * LDK + kk * KS], lane);
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) st[qb][nb] = M_::mma(kf, qf[qb][kk], st[qb][nb]);
      }
    }
    // ---- column softmax per block (log2 domain, lazy rescaling as in attn_body) ----
    uint2v pb[QB][NB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const float relh_tile = relh_s[il[qb] * RH + kt / BKV];
      float sv[NB][4];
      float mx = -INFINITY;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sc = fmaf(st[qb][nb][r], c1, relh_tile + relw_reg[qb][nb][r]);
          sv[nb][r] = sc;
          mx = fmaxf(mx, sc);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = mx > m_run[qb] + 8.f ? mx : m_run[qb];
      const bool moved = m_new != m_run[qb];
      const float mref = m_new == -INFINITY ? 0.f : m_new;
      if (__builtin_amdgcn_ballot_w64(moved) != 0) {
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - mref);
        l_run[qb] *= alpha;
#pragma unroll
        for (int d = 0; d < DB; ++d) ot[qb][d] *= alpha;
        m_run[qb] = m_new;
      }
      float rs = 0.f;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sv[nb][r] = __builtin_amdgcn_exp2f(sv[nb][r] - mref);
          rs += sv[nb][r];
        }
        pb[qb][nb] = uint2v{pack2_from_f32<T>(sv[nb][0], sv[nb][1]), pack2_from_f32<T>(sv[nb][2], sv[nb][3])};
      }
      l_run[qb] += rs;
    }
    // ---- O^T += V^T P^T: one transposed V block read, two MFMAs ----
#pragma unroll
    for (int nb = 0; nb < NB; nb += 2) {
      uint2v vt0[DB], vt1[DB];
      lds_tr_blocks2<DB>(vbase + (uint32_t)(nb * 16 * LDV * 2), vbase + (uint32_t)((nb + 1) * 16 * LDV * 2), vt0, vt1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int d = 0; d < DB; ++d)
          ot[qb][d] = mfma_16x16x16<T>(__builtin_bit_cast(short4v, vt0[d]), __builtin_bit_cast(short4v, pb[qb][nb]), ot[qb][d]);
#pragma unroll
        for (int d = 0; d < DB; ++d)
          ot[qb][d] = mfma_16x16x16<T>(__builtin_bit_cast(short4v, vt1[d]), __builtin_bit_cast(short4v, pb[qb][nb + 1]), ot[qb][d]);
      }
    }
    __syncthreads();  // K / V tile free for the next iteration
  }
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float l = l_run[qb];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    const int64_t off = (int64_t)b * a.o_bs + (int64_t)h * a.o_hs + (int64_t)(q0 + il[qb]) * a.o_rs;
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      const float4v v = ot[qb][d] * inv;
      const int c = d * 16 + 4 * g;
      if (a.o_f32) *reinterpret_cast<float4v*>(reinterpret_cast<float*>(a.O) + off + c) = v;
      else store4_from_f32<T>(reinterpret_cast<T*>(a.O) + off + c, v[0], v[1], v[2], v[3]);
    }
  }
}
template <typename T, int HD, int NWV>
__global__ __launch_bounds__(NWV * 64, 2) void attn_g2_kernel(AttnArgs a, int gx, int gy, int total) {  // 2 waves per SIMD
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // plain launch: gridDim.x == total; capped (AttnArgs::max_wg): every workgroup walks v, v + gridDim.x, ...
  for (int v = blockIdx.x; v < total; v += gridDim.x) {
    attn_g2_body<T, HD, NWV>(a, v % gx, (v / gx) % gy, v / (gx * gy), smem);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Split-pair attention (ANYREF_MODE_PARITY16): q, k, v are f32 in memory; on their way into registers / LDS every
// value is split into two bf16 terms (hi = bf16(x), lo = bf16(x - hi), 2^-18 relative) and both products run on the
// 16-bit MFMA as THREE passes each into the same f32 accumulator --
//     S^T = Kh Qh^T + Kh Ql^T + Kl Qh^T          O^T += Vh^T Ph^T + Vl^T Ph^T + Vh^T Pl^T
// (the lo x lo terms are 2^-18 of the result: below the pair's own rounding) -- instead of v_mfma_f32_16x16x4_f32 at 1/16
// of the rate: 3/16 of the f32 kernel's MFMA time at the same f32-level result.  Same transposed formulation as attn_body
// (lane = query, P stays in registers, V^T via ds_read_b64_tr_b16), streaming K / V tiles, log2-domain softmax with lazy
// rescaling; causal / kv_len / q_len masks, rel-pos bias from the P buffer (rel_p) or the rel_h / rel_w arrays; the output
// rows leave as f32 or as a split pair (o_split: the A operand of the proj / o_proj GEMM).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void split8(const float (&f)[8], short8& hi, short8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bf16 h = f2bf(f[i]);
    hi[i] = (short)h.x;
    lo[i] = (short)f2bf(f[i] - bf2f(h)).x;
  }
}
template <int HD, int NWV, int BKV>
__device__ __forceinline__ void attn_sp_body(const AttnArgs& a, const int bx, const int by, const int bz, char* smem) {
  constexpr int NT = NWV * 64, KS = 32, VEC = 8;
  constexpr int BQ = 16 * NWV, HDK = (HD + KS - 1) / KS * KS, LDK = HDK + VEC;
  constexpr int LDV = ((HD * 2 + 255) / 256 * 256 + 32) / 2;
  constexpr int NB = BKV / 16, DB = HD / 16;
  static_assert(HD % 16 == 0 && BKV % 16 == 0, "head dim / key tile must be multiples of 16");
  bf16* Kh = reinterpret_cast<bf16*>(smem);
  bf16* Kl = Kh + BKV * LDK;
  bf16* Vh = Kl + BKV * LDK;
  bf16* Vl = Vh + BKV * LDV;
  float* relh_s = reinterpret_cast<float*>(Vl + BKV * LDV);
  float* relw_s = relh_s + BQ * a.kh;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = lane & 15, g = lane >> 4;
  const int b = bz, h = by + a.h_off, q0 = bx * BQ;
  const int kv_len = a.kv_len ? a.kv_len[b] : a.Sk;
  const int q_len = a.q_len ? a.q_len[b] : a.Sq;
  if (q0 >= q_len) return;  // uniform per workgroup
  const int pos0 = a.q_pos0 ? a.q_pos0[b] : 0;
  int kv_end = kv_len;
  if (a.causal) {
    const int imax = (q0 + BQ < q_len ? q0 + BQ : q_len) - 1;
    if (pos0 + imax + 1 < kv_end) kv_end = pos0 + imax + 1;
  }
  const float* Qb = reinterpret_cast<const float*>(a.Q) + (int64_t)b * a.q_bs + (int64_t)h * a.q_hs;
  const float* Kb = reinterpret_cast<const float*>(a.K) + (int64_t)b * a.k_bs + (int64_t)h * a.k_hs;
  const float* Vb = reinterpret_cast<const float*>(a.V) + (int64_t)b * a.v_bs + (int64_t)h * a.v_hs;

  const int il = wave * 16 + qi, iq = q0 + il;
  const bool q_ok = iq < q_len;
  short8 qh[HDK / KS], ql[HDK / KS];
  {
    const float* qrow = Qb + (int64_t)(q_ok ? iq : 0) * a.q_rs;
#pragma unroll
    for (int kk = 0; kk < HDK / KS; ++kk) {
      const int d = kk * KS + 8 * g;
      float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (q_ok && d < HD) {
        const float4v x0 = *reinterpret_cast<const float4v*>(qrow + d), x1 = *reinterpret_cast<const float4v*>(qrow + d + 4);
        f[0] = x0[0]; f[1] = x0[1]; f[2] = x0[2]; f[3] = x0[3]; f[4] = x1[0]; f[5] = x1[1]; f[6] = x1[2]; f[7] = x1[3];
      }
      split8(f, qh[kk], ql[kk]);
    }
  }
  constexpr float LOG2E = 1.4426950408889634f;
  // f32 rel-pos TABLES (the windows, 2 k - 1 <= 32 entries): the bias rows of this block's queries are computed here -- R q^T on
  // the 16-bit MFMA with BOTH sides as bf16 pairs (Rl qh + Rh ql + Rh qh, the q fragments this wave already holds), then the
  // reference's shifted gather (get_rel_pos, image_encoder.py:321-392) as a scatter, exactly as the 16-bit resident-key kernel
  // does it.  Replaces the f32-MFMA rel-pos GEMM launch (40 us per layer at SAM-H) and its 17 MB P buffer round trip.
  const bool rel_tab = a.rel_tab_h != nullptr;
  const bool has_rel = a.rel_h != nullptr || a.rel_p != nullptr || rel_tab;
  if (rel_tab) {
    bf16* Rs = reinterpret_cast<bf16*>(relw_s + BQ * a.kw);  // [table][hi | lo][32][LDK], zero padded
    for (int i = tid; i < BQ * (a.kh + a.kw); i += NT) relh_s[i] = 0.f;  // (relw_s follows relh_s)
    constexpr int RV = LDK / VEC;
    for (int i = tid; i < 2 * 32 * RV; i += NT) {
      const int t = i / (32 * RV), e = (i / RV) % 32, d = (i % RV) * VEC;
      const int ne = 2 * (t == 0 ? a.kh : a.kw) - 1;
      const float* tab = reinterpret_cast<const float*>(t == 0 ? a.rel_tab_h : a.rel_tab_w) + (int64_t)e * a.rel_tab_ld + d;
      float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (e < ne && d < HD) {
        const float4v x0 = *reinterpret_cast<const float4v*>(tab), x1 = *reinterpret_cast<const float4v*>(tab + 4);
        f[0] = x0[0]; f[1] = x0[1]; f[2] = x0[2]; f[3] = x0[3]; f[4] = x1[0]; f[5] = x1[1]; f[6] = x1[2]; f[7] = x1[3];
      }
      short8 hi, lo;
      split8(f, hi, lo);
      *reinterpret_cast<short8*>(&Rs[((t * 2 + 0) * 32 + e) * LDK + d]) = hi;
      *reinterpret_cast<short8*>(&Rs[((t * 2 + 1) * 32 + e) * LDK + d]) = lo;
    }
    __syncthreads();
    float4v pb[2][2];  // [table][entry block]: P^T[entry 16 blk + 4 g + r][this lane's query]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        pb[t][blk] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < HDK / KS; ++kk) {
          const short8 rh = *reinterpret_cast<const short8*>(&Rs[((t * 2 + 0) * 32 + blk * 16 + qi) * LDK + kk * KS + 8 * g]);
          const short8 rl = *reinterpret_cast<const short8*>(&Rs[((t * 2 + 1) * 32 + blk * 16 + qi) * LDK + kk * KS + 8 * g]);
          pb[t][blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rl, qh[kk], pb[t][blk], 0, 0, 0);  // small terms first
          pb[t][blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rh, ql[kk], pb[t][blk], 0, 0, 0);
          pb[t][blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rh, qh[kk], pb[t][blk], 0, 0, 0);
        }
      }
    if (q_ok) {
      const int y = iq / a.kw, x = iq - y * a.kw;
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = blk * 16 + 4 * g + r;
          const int ch = y + a.kh - 1 - e, cw = x + a.kw - 1 - e;
          if (ch >= 0 && ch < a.kh) relh_s[il * a.kh + ch] = pb[0][blk][r] * LOG2E;
          if (cw >= 0 && cw < a.kw) relw_s[il * a.kw + cw] = pb[1][blk][r] * LOG2E;
        }
    }
    // (the barrier in front of the first key tile orders these stores before any bias read)
  } else if (a.rel_p) {
    const float* P = a.rel_p + (int64_t)h * a.rel_hs + ((int64_t)b * a.Sq + q0) * a.rel_ld;
    const int np = a.rel_ld / 2;
    for (int i = tid; i < BQ * a.kh; i += NT) {
      const int r = i / a.kh, c = i % a.kh;
      const int y = (q0 + r) / a.kw;
      relh_s[i] = q0 + r < q_len ? P[(int64_t)r * a.rel_ld + (y - c + a.kh - 1)] * LOG2E : 0.f;
    }
    for (int i = tid; i < BQ * a.kw; i += NT) {
      const int r = i / a.kw, c = i % a.kw;
      const int x = (q0 + r) % a.kw;
      relw_s[i] = q0 + r < q_len ? P[(int64_t)r * a.rel_ld + np + (x - c + a.kw - 1)] * LOG2E : 0.f;
    }
  } else if (has_rel) {
    const int Hall = a.h_total > 0 ? a.h_total : a.H;
    const float* rh = a.rel_h + ((int64_t)b * Hall + h) * a.Sq * a.kh;
    const float* rw = a.rel_w + ((int64_t)b * Hall + h) * a.Sq * a.kw;
    for (int i = tid; i < BQ * a.kh; i += NT) {
      const int r = i / a.kh, c = i % a.kh;
      relh_s[i] = q0 + r < q_len ? rh[(int64_t)(q0 + r) * a.kh + c] * LOG2E : 0.f;
    }
    for (int i = tid; i < BQ * a.kw; i += NT) {
      const int r = i / a.kw, c = i % a.kw;
      relw_s[i] = q0 + r < q_len ? rw[(int64_t)(q0 + r) * a.kw + c] * LOG2E : 0.f;
    }
  }
  // key tile = exactly one bias row (SAM global attention: kw == BKV): the kw term of this lane's keys never changes
  const bool rel_fast = has_rel && a.kw == BKV;
  const unsigned kw_magic = has_rel ? (1u << 20) / (unsigned)a.kw + 1u : 0u;  // j / kw for j < 4096, kw <= 64
  float relw_reg[NB][4];
  if (rel_fast) {
    __syncthreads();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) relw_reg[nb][r] = relw_s[il * a.kw + nb * 16 + 4 * g + r];
  }

  float m_run = -INFINITY, l_run = 0.f;
  float4v ot[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) ot[d] = float4v{0.f, 0.f, 0.f, 0.f};

  // K [BKV][HDK] and V [BKV][HD] tiles in units of 8 elements (two 16-byte f32 loads -> one 16-byte hi + one 16-byte lo LDS
  // store); register-staged one tile ahead
  constexpr int KVEC = HDK / VEC, VVEC = HD / VEC;
  constexpr int KPT = (BKV * KVEC + NT - 1) / NT, VPT = (BKV * VVEC + NT - 1) / NT;
  float4v kreg[KPT][2], vreg[VPT][2];
  auto gload_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int v = tid + i * NT, row = v / KVEC, d = (v % KVEC) * VEC;
      const bool ok = v < BKV * KVEC && kt + row < kv_end && d < HD;
      const float* p = Kb + (int64_t)(kt + row) * a.k_rs + d;
      kreg[i][0] = ok ? *reinterpret_cast<const float4v*>(p) : float4v{0.f, 0.f, 0.f, 0.f};
      kreg[i][1] = ok ? *reinterpret_cast<const float4v*>(p + 4) : float4v{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int v = tid + i * NT, row = v / VVEC, d = (v % VVEC) * VEC;
      const bool ok = v < BKV * VVEC && kt + row < kv_end;
      const float* p = Vb + (int64_t)(kt + row) * a.v_rs + d;
      vreg[i][0] = ok ? *reinterpret_cast<const float4v*>(p) : float4v{0.f, 0.f, 0.f, 0.f};
      vreg[i][1] = ok ? *reinterpret_cast<const float4v*>(p + 4) : float4v{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto sstore_tile = [&]() {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      const int v = tid + i * NT, row = v / KVEC, d = (v % KVEC) * VEC;
      if (v < BKV * KVEC) {
        const float f[8] = {kreg[i][0][0], kreg[i][0][1], kreg[i][0][2], kreg[i][0][3], kreg[i][1][0], kreg[i][1][1], kreg[i][1][2], kreg[i][1][3]};
        short8 hi, lo;
        split8(f, hi, lo);
        *reinterpret_cast<short8*>(&Kh[row * LDK + d]) = hi;
        *reinterpret_cast<short8*>(&Kl[row * LDK + d]) = lo;
      }
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int v = tid + i * NT, row = v / VVEC, d = (v % VVEC) * VEC;
      if (v < BKV * VVEC) {
        const float f[8] = {vreg[i][0][0], vreg[i][0][1], vreg[i][0][2], vreg[i][0][3], vreg[i][1][0], vreg[i][1][1], vreg[i][1][2], vreg[i][1][3]};
        short8 hi, lo;
        split8(f, hi, lo);
        *reinterpret_cast<short8*>(&Vh[row * LDV + d]) = hi;
        *reinterpret_cast<short8*>(&Vl[row * LDV + d]) = lo;
      }
    }
  };
  if (kv_end > 0) gload_tile(0);
  const float c1 = a.scale * LOG2E;
  const uint32_t voff = (uint32_t)((4 * g + (qi >> 2)) * LDV + 4 * (qi & 3)) * 2u;
  const uint32_t vbase_h = (uint32_t)(reinterpret_cast<const char*>(Vh) - smem) + voff;
  const uint32_t vbase_l = (uint32_t)(reinterpret_cast<const char*>(Vl) - smem) + voff;
  for (int kt = 0; kt < kv_end; kt += BKV) {
    sstore_tile();
    __syncthreads();
    if (kt + BKV < kv_end) gload_tile(kt + BKV);
    const bool active = q0 + wave * 16 < q_len && !(a.causal && kt > pos0 + q0 + wave * 16 + 15);
    if (active) {
      // ---- S^T = Kh Qh^T + Kh Ql^T + Kl Qh^T ----
      float4v st[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        st[nb] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < HDK / KS; ++kk) {
          const short8 kfh = *reinterpret_cast<const short8*>(&Kh[(nb * 16 + qi) * LDK + kk * KS + 8 * g]);
          const short8 kfl = *reinterpret_cast<const short8*>(&Kl[(nb * 16 + qi) * LDK + kk * KS + 8 * g]);
          st[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfl, qh[kk], st[nb], 0, 0, 0);  // small terms first
          st[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfh, ql[kk], st[nb], 0, 0, 0);
          st[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfh, qh[kk], st[nb], 0, 0, 0);
        }
      }
      // ---- scale, bias, mask; column softmax of this lane's query (log2 domain) ----
      float sv[NB][4];
      float mx = -INFINITY;
      const float relh_tile = rel_fast ? relh_s[il * a.kh + kt / BKV] : 0.f;
      const bool need_mask = kt + BKV > kv_len || (a.causal && kt + BKV - 1 > pos0 + q0 + wave * 16);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = kt + nb * 16 + 4 * g + r;
          float sc = st[nb][r] * c1;
          if (rel_fast) {
            sc += relh_tile + relw_reg[nb][r];
          } else if (has_rel) {
            int jh = (int)(((unsigned)j * kw_magic) >> 20), jw = j - jh * a.kw;
            if (jh >= a.kh) jh = 0, jw = 0;  // keys past the end (masked below): keep the LDS reads in range
            sc += relh_s[il * a.kh + jh] + relw_s[il * a.kw + jw];
          }
          if (need_mask) {
            const bool valid = j < kv_len && (!a.causal || j <= pos0 + iq);
            sc = valid ? sc : -INFINITY;
          }
          sv[nb][r] = sc;
          mx = fmaxf(mx, sc);
        }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = mx > m_run + 8.f ? mx : m_run;  // lazy rescaling (attn_body)
      const bool moved = m_new != m_run;
      const float mref = m_new == -INFINITY ? 0.f : m_new;
      if (__builtin_amdgcn_ballot_w64(moved) != 0) {
        const float alpha = __builtin_amdgcn_exp2f(m_run - mref);
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < DB; ++d) ot[d] *= alpha;
        m_run = m_new;
      }
      float rs = 0.f;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sv[nb][r] = __builtin_amdgcn_exp2f(sv[nb][r] - mref);
          rs += sv[nb][r];
        }
      l_run += rs;
      // ---- O^T += Vh^T Ph^T + Vl^T Ph^T + Vh^T Pl^T ----
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        uint2v vth[DB], vtl[DB];
        lds_tr_blocks<DB>(vbase_h + (uint32_t)(nb * 16 * LDV * 2), vth);
        lds_tr_blocks<DB>(vbase_l + (uint32_t)(nb * 16 * LDV * 2), vtl);
        __builtin_amdgcn_sched_barrier(0);
        uint16_t ph[4], pl[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) sp_split(sv[nb][r], ph[r], pl[r]);
        const uint2v pbh = uint2v{(uint32_t)ph[0] | ((uint32_t)ph[1] << 16), (uint32_t)ph[2] | ((uint32_t)ph[3] << 16)};
        const uint2v pbl = uint2v{(uint32_t)pl[0] | ((uint32_t)pl[1] << 16), (uint32_t)pl[2] | ((uint32_t)pl[3] << 16)};
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          ot[d] = mfma_16x16x16<bf16>(__builtin_bit_cast(short4v, vtl[d]), __builtin_bit_cast(short4v, pbh), ot[d]);
          ot[d] = mfma_16x16x16<bf16>(__builtin_bit_cast(short4v, vth[d]), __builtin_bit_cast(short4v, pbl), ot[d]);
          ot[d] = mfma_16x16x16<bf16>(__builtin_bit_cast(short4v, vth[d]), __builtin_bit_cast(short4v, pbh), ot[d]);
        }
      }
    }
    __syncthreads();  // K / V tiles free for the next iteration
  }
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (q_ok) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* Obf = reinterpret_cast<float*>(a.O) + (int64_t)b * a.o_bs + (int64_t)h * a.o_hs + (int64_t)iq * a.o_rs;
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      const float4v v = ot[d] * inv;
      const int c = d * 16 + 4 * g;
      if (a.o_split)
        st4<sp16>(reinterpret_cast<sp16*>(a.O) + (int64_t)b * a.o_bs + (int64_t)iq * a.o_rs, (int)(h * a.o_hs) + c, v[0], v[1], v[2], v[3]);
      else
        *reinterpret_cast<float4v*>(Obf + c) = v;
    }
  }
}
template <int HD, int NWV, int BKV>
__global__ __launch_bounds__(NWV * 64) void attn_sp_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_sp_body<HD, NWV, BKV>(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, smem);
}
template <int HD, int NWV, int BKV>
static void attn_sp_launch_cfg(const AttnArgs& a, hipStream_t s) {
  constexpr int BQ = 16 * NWV, HDK = (HD + 31) / 32 * 32, LDK = HDK + 8, LDV = ((HD * 2 + 255) / 256 * 256 + 32) / 2;
  size_t lds = 2 * 2 * (size_t)(BKV * LDK + BKV * LDV);
  if (a.rel_h || a.rel_p || a.rel_tab_h) lds += sizeof(float) * BQ * (a.kh + a.kw);
  if (a.rel_tab_h) {
    if (2 * a.kh - 1 > 32 || 2 * a.kw - 1 > 32 || a.Sq != a.kh * a.kw || a.rel_tab_ld % 4 || ((uintptr_t)a.rel_tab_h & 15) ||
        ((uintptr_t)a.rel_tab_w & 15))
      throw std::runtime_error("split-pair attention: rel-pos tables need k <= 16, Sq == kh * kw and 16-byte aligned f32 rows");
    lds += 2 * 2 * 2 * 32 * (size_t)LDK;  // two tables x (hi, lo) x 32 entries, bf16
  }
  if (lds > 160 * 1024) throw std::runtime_error("split-pair attention: LDS budget exceeded");
  auto kern = &attn_sp_kernel<HD, NWV, BKV>;
  static KernelAttrOnce once;
  ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024);
  static const std::string tag = "attn_sp16_hd" + std::to_string(HD) + "_w" + std::to_string(NWV);
  const double flops = 4.0 * a.B * a.H * (double)a.Sq * a.Sk * HD * (a.causal ? 0.5 : 1.0);
  const double bytes = (double)a.B * a.H * HD * 4.0 * (2.0 * a.Sq + 2.0 * a.Sk);
  ProfScope prof(tag.c_str(), flops, bytes, s);
  const int gx = cdiv(a.Sq, BQ);
  if (a.max_wg > 0 && (int64_t)gx * a.H * a.B > a.max_wg) {
    // CU share of the side stream (the SAM encoder beside the decode loop): the launch is cut into groups of batch items
    // (windows) or, where one item alone exceeds the cap (global attention: 32 query blocks x 16 heads), groups of heads
    const int per_b = std::max(1, a.max_wg / (gx * a.H)), per_h = gx * a.H > a.max_wg ? std::max(1, a.max_wg / gx) : a.H;
    for (int b0 = 0; b0 < a.B; b0 += per_b)
      for (int h0 = 0; h0 < a.H; h0 += per_h) {
        AttnArgs c = a;
        c.B = std::min(per_b, a.B - b0);
        c.H = std::min(per_h, a.H - h0);
        c.h_off = h0;
        c.h_total = a.H;
        c.Q = reinterpret_cast<const float*>(a.Q) + (int64_t)b0 * a.q_bs;
        c.K = reinterpret_cast<const float*>(a.K) + (int64_t)b0 * a.k_bs;
        c.V = reinterpret_cast<const float*>(a.V) + (int64_t)b0 * a.v_bs;
        c.O = a.o_split ? (void*)(reinterpret_cast<sp16*>(a.O) + (int64_t)b0 * a.o_bs) : (void*)(reinterpret_cast<float*>(a.O) + (int64_t)b0 * a.o_bs);
        if (a.rel_p) c.rel_p = a.rel_p + (int64_t)b0 * a.Sq * a.rel_ld;
        if (a.rel_h) c.rel_h = a.rel_h + (int64_t)b0 * a.H * a.Sq * a.kh;
        if (a.rel_w) c.rel_w = a.rel_w + (int64_t)b0 * a.H * a.Sq * a.kw;
        if (a.q_len) c.q_len = a.q_len + b0;
        if (a.kv_len) c.kv_len = a.kv_len + b0;
        if (a.q_pos0) c.q_pos0 = a.q_pos0 + b0;
        hipLaunchKernelGGL(kern, dim3(gx, c.H, c.B), dim3(NWV * 64), lds, s, c);
      }
    return;
  }
  hipLaunchKernelGGL(kern, dim3(gx, a.H, a.B), dim3(NWV * 64), lds, s, a);
}
// whether launch_attention<float> takes the split-pair kernel for this call (AttnArgs::sp16 set by the caller)
static bool attn_sp_takes(const AttnArgs& a) {
  static const bool off = getenv("ANYREF_NO_SP16_ATTN") != nullptr;  // lab knob: the f32-MFMA kernel instead
  if (off || !a.sp16 || (a.hd != 64 && a.hd != 80 && a.hd != 128)) return false;
  if (a.o_f32 && a.o_split) return false;
  if (a.rel_tab_h && !attention_takes_rel_tables(4, a.hd, a.Sq, a.Sk, a.kh, a.kw, true)) return false;
  if ((a.rel_h || a.rel_p) && (a.kw > 64 || a.kh * a.kw > 4096)) return false;  // (the multiply-shift j / kw)
  if (a.o_rs % 4 || a.o_hs % 4 || a.o_bs % 4 || ((uintptr_t)a.O & 15)) return false;
  return true;
}
template <int HD>
static void attn_sp_launch(const AttnArgs& a, hipStream_t s) {
  if (a.Sq >= 1024 && !a.causal) {  // SAM global attention
    attn_sp_launch_cfg<HD, 8, 64>(a, s);
    return;
  }
  if constexpr (HD == 80) {
    // 14 x 14 windows (196 tokens): 2 x 112 queries, 3 key tiles of 80 (one 13-wave block per window-head would stage K / V
    // once, but 13 waves leave 128 VGPRs each: 132 - 204 bytes of scratch in every tile size tried)
    if (a.Sq > 192 && a.Sq <= 224 && !a.causal) {
      attn_sp_launch_cfg<HD, 7, 80>(a, s);
      return;
    }
  }
  attn_sp_launch_cfg<HD, 4, 64>(a, s);
}

template <typename T, int HD, int NWV = 4, int BKV_ = 0, int NRES = 0>
__global__ __launch_bounds__(NWV * 64) void attn_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_body<T, HD, NWV, BKV_, NRES>(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, smem);
}
// Capped grid (AttnArgs::max_wg workgroups): every workgroup walks the (query block, head, batch) triples
// v, v + gridDim.x, ... -- the SAM encoder's attention on the side stream, which must leave CUs to the decode GEMVs
// of the main stream (as gemm_glds_kernel<.., PERSIST>).  No key splits in this form.
template <typename T, int HD, int NWV, int BKV_, int NRES>
__global__ __launch_bounds__(NWV * 64) void attn_walk_kernel(AttnArgs a, int gx, int gy, int total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int v = blockIdx.x; v < total; v += gridDim.x) {
    attn_body<T, HD, NWV, BKV_, NRES>(a, v % gx, (v / gx) % gy, v / (gx * gy), smem);
    __syncthreads();  // the next triple's LDS tiles are not written before every wave has left this one
  }
}

// merge of the kv_splits partial results: out = sum_s w_s O_s / sum_s w_s l_s, w_s = exp(m_s - max m)
// (exp2 when the partials are in the log2 domain: bf16 path)
template <typename T>
__global__ __launch_bounds__(256) void attn_combine_kernel(AttnArgs a, int HD) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t rows = (int64_t)a.B * a.H * a.Sq;
  if (i >= rows * HD) return;
  const int d = (int)(i % HD);
  const int64_t row = i / HD;
  const int iq = (int)(row % a.Sq), h = (int)((row / a.Sq) % a.H), b = (int)(row / ((int64_t)a.Sq * a.H));
  float M = -INFINITY;
  for (int s = 0; s < a.kv_splits; ++s) M = fmaxf(M, a.part_ml[(s * rows + row) * 2]);
  float o = 0.f, l = 0.f;
  for (int s = 0; s < a.kv_splits; ++s) {
    const float m = a.part_ml[(s * rows + row) * 2];
    const float w = m == -INFINITY ? 0.f : (sizeof(T) == 2 ? exp2f(m - M) : expf(m - M));
    o = fmaf(w, a.part_o[(s * rows + row) * HD + d], o);
    l = fmaf(w, a.part_ml[(s * rows + row) * 2 + 1], l);
  }
  const float v = l > 0.f ? o / l : 0.f;
  const int64_t off = (int64_t)b * a.o_bs + (int64_t)h * a.o_hs + (int64_t)iq * a.o_rs + d;
  if (a.o_f32) reinterpret_cast<float*>(a.O)[off] = v;
  else if (a.o_split) st1<sp16>(reinterpret_cast<sp16*>(a.O) + (int64_t)b * a.o_bs + (int64_t)iq * a.o_rs, (int)(h * a.o_hs) + d, v);
  else reinterpret_cast<T*>(a.O)[off] = from_f32<T>(v);
}

// per-stream partial-result workspace (grown on demand, like the split-K slabs in gemm.hip)
namespace {
struct AttnWs { hipStream_t s; float* p; size_t cap; };
// process-wide, keyed by the stream handle (unique while the stream lives): a handle destroyed on another thread
// (Python GC) still finds and frees the workspaces of the streams it owns.  Entries are only touched under the lock.
std::mutex& attn_pool_mu() {
  static std::mutex mu;
  return mu;
}
std::vector<AttnWs>& attn_pool() {
  static std::vector<AttnWs> pool;
  return pool;
}
}  // namespace
void attn_release_workspace(hipStream_t s) {
  std::lock_guard<std::mutex> lock(attn_pool_mu());
  auto& pool = attn_pool();
  for (size_t i = 0; i < pool.size(); ++i)
    if (pool[i].s == s) {
      (void)hipFree(pool[i].p);
      pool.erase(pool.begin() + i);
      return;
    }
}
static float* attn_workspace(hipStream_t s, size_t bytes) {
  using Ws = AttnWs;
  std::lock_guard<std::mutex> lock(attn_pool_mu());
  auto& pool = attn_pool();
  for (auto& w : pool)
    if (w.s == s) {
      if (w.cap < bytes) {
        HIP_TRY(hipStreamSynchronize(s));
        (void)hipFree(w.p);
        HIP_TRY(hipMalloc((void**)&w.p, bytes));
        w.cap = bytes;
      }
      return w.p;
    }
  Ws w{s, nullptr, bytes};
  HIP_TRY(hipMalloc((void**)&w.p, bytes));
  pool.push_back(w);
  return w.p;
}

template <typename T, int HD, int NWV, int BKVP, int NRES = 0>
static void attn_launch_cfg(const AttnArgs& a_in, hipStream_t s) {
  AttnArgs a = a_in;
  constexpr int KS = AMma<T>::KS, VEC = AMma<T>::VEC, BKV = BKVP > 0 ? BKVP : AttnTile<T>::BKV, BQ = 16 * NWV;
  constexpr int HDK = (HD + KS - 1) / KS * KS, LDK = HDK + VEC;
  constexpr int LDV = sizeof(T) == 2 ? ((HD * 2 + 255) / 256 * 256 + 32) / 2 : HD + VEC;
  size_t lds = sizeof(T) * (BKV * LDK + BKV * LDV) * (NRES > 0 ? NRES : 1);
  if (NRES > 0 && (a.Sk > NRES * BKV || (a.Sq > BQ && a.rel_tab_h) || a.kv_len || a.causal))
    throw std::runtime_error("attention: resident-key form needs Sk <= NRES * BKV (and one query block with rel-pos tables)");
  if (a.rel_h || a.rel_p || a.rel_tab_h) lds += sizeof(float) * BQ * (a.kh + a.kw);
  if (a.rel_tab_h) {
    if (NRES == 0 || sizeof(T) != 2 || 2 * a.kh - 1 > 32 || 2 * a.kw - 1 > 32 || a.rel_tab_ld % VEC ||
        ((uintptr_t)a.rel_tab_h & 15) || ((uintptr_t)a.rel_tab_w & 15))
      throw std::runtime_error("attention: rel-pos tables are taken by the bf16 resident-key form only (k <= 16)");
    lds += sizeof(T) * 2 * 32 * LDK;
  }
  if (lds > 160 * 1024) throw std::runtime_error("attention: LDS budget exceeded");
  static KernelAttrOnce once;
  ensure_dyn_lds(once, reinterpret_cast<const void*>(&attn_kernel<T, HD, NWV, BKVP, NRES>), 160 * 1024);
  // few query blocks over many keys: split the keys until ~256 workgroups (at least 256 keys per split)
  a.kv_splits = 1;
  const int64_t wgs = (int64_t)cdiv(a.Sq, BQ) * a.H * a.B;
  if (!a.causal && !a.kv_len && wgs < 128 && a.Sk >= 1024) {
    int sp = (int)(256 / wgs);
    sp = sp > a.Sk / 256 ? a.Sk / 256 : sp;
    a.kv_splits = sp > 16 ? 16 : (sp < 1 ? 1 : sp);
  }
  if (a.kv_splits > 1) {
    const size_t rows = (size_t)a.kv_splits * a.B * a.H * a.Sq;
    float* ws = attn_workspace(s, rows * (HD + 2) * sizeof(float));
    a.part_o = ws;
    a.part_ml = ws + rows * HD;
  }
  dim3 grid(cdiv(a.Sq, BQ) * a.kv_splits, a.H, a.B);
  static const std::string tag = std::string(is_half16<T>::value ? "attn_f16_hd" : sizeof(T) == 2 ? "attn_bf16_hd" : "attn_f32_hd") +
                                 std::to_string(HD) +
                                 (NWV == 4 ? "" : "_w" + std::to_string(NWV)) + (NRES > 0 ? "_res" : "");
  const double flops = 4.0 * a.B * a.H * (double)a.Sq * a.Sk * HD * (a.causal ? 0.5 : 1.0);
  const double bytes = (double)a.B * a.H * HD * sizeof(T) * (2.0 * a.Sq + 2.0 * a.Sk);
  ProfScope prof(tag.c_str(), flops, bytes, s);
  if constexpr (sizeof(T) == 2 && HD == 80 && NWV >= 8) {  // the SAM encoder's two attention forms
    const int total = (int)(grid.x * grid.y * grid.z);
    if (a.max_wg > 0 && a.kv_splits == 1 && total > a.max_wg) {
      if constexpr (NRES > 0) {
        // resident-key window form (13 waves = 128 VGPRs per wave: a walking loop around the body spills): the cap is
        // kept by launching the windows in groups of max_wg / (query blocks x heads)
        const int per = std::max(1, a.max_wg / (int)(grid.x * grid.y));
        for (int b0 = 0; b0 < a.B; b0 += per) {
          AttnArgs c = a;
          c.B = std::min(per, a.B - b0);
          c.Q = reinterpret_cast<const T*>(a.Q) + (int64_t)b0 * a.q_bs;
          c.K = reinterpret_cast<const T*>(a.K) + (int64_t)b0 * a.k_bs;
          c.V = reinterpret_cast<const T*>(a.V) + (int64_t)b0 * a.v_bs;
          c.O = reinterpret_cast<T*>(a.O) + (int64_t)b0 * a.o_bs;
          if (a.rel_p) c.rel_p = a.rel_p + (int64_t)b0 * a.Sq * a.rel_ld;
          if (a.rel_h) c.rel_h = a.rel_h + (int64_t)b0 * a.H * a.Sq * a.kh;
          if (a.rel_w) c.rel_w = a.rel_w + (int64_t)b0 * a.H * a.Sq * a.kw;
          if (a.q_len) c.q_len = a.q_len + b0;
          if (a.q_pos0) c.q_pos0 = a.q_pos0 + b0;
          hipLaunchKernelGGL((attn_kernel<T, HD, NWV, BKVP, NRES>), dim3(grid.x, grid.y, c.B), dim3(NWV * 64), lds, s, c);
        }
        return;
      } else {
        auto kw = &attn_walk_kernel<T, HD, NWV, BKVP, NRES>;
        static KernelAttrOnce oncew;
        ensure_dyn_lds(oncew, reinterpret_cast<const void*>(kw), 160 * 1024);
        hipLaunchKernelGGL(kw, dim3(a.max_wg), dim3(NWV * 64), lds, s, a, (int)grid.x, (int)grid.y, total);
        return;
      }
    }
  }
  hipLaunchKernelGGL((attn_kernel<T, HD, NWV, BKVP, NRES>), grid, dim3(NWV * 64), lds, s, a);
  if (a.kv_splits > 1) {
    const int64_t n = (int64_t)a.B * a.H * a.Sq * HD;
    hipLaunchKernelGGL((attn_combine_kernel<T>), dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, a, HD);
  }
}

template <typename T, int HD, int NWV>
static void attn_g2_launch(const AttnArgs& a, hipStream_t s) {
  constexpr int BQ = 32 * NWV, LDK = 96 + 8, LDV = ((HD * 2 + 255) / 256 * 256 + 32) / 2;
  const size_t lds = sizeof(T) * 64 * (LDK + LDV) + sizeof(float) * BQ * (a.kh + 1);
  auto kern = &attn_g2_kernel<T, HD, NWV>;
  static KernelAttrOnce once;
  ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), 160 * 1024);
  const int gx = a.Sq / BQ, gy = a.H, total = gx * gy * a.B;
  static const std::string tag = std::string(is_half16<T>::value ? "attn_f16_hd" : "attn_bf16_hd") + std::to_string(HD) + "_g2w" +
                                 std::to_string(NWV);
  const double flops = 4.0 * a.B * a.H * (double)a.Sq * a.Sk * HD;
  const double bytes = (double)a.B * a.H * HD * sizeof(T) * (2.0 * a.Sq + 2.0 * a.Sk);
  ProfScope prof(tag.c_str(), flops, bytes, s);
  const int grid = a.max_wg > 0 && total > a.max_wg ? a.max_wg : total;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NWV * 64), lds, s, a, gx, gy, total);
}

template <typename T, int HD>
static void attn_launch(const AttnArgs& a, hipStream_t s) {
  // SAM windows (14 x 14 = 196 tokens, bf16, head dim 80): one workgroup of 13 waves per (window, head) with all
  // keys resident as 5 tiles of 48 (48: the largest tile that stays under the 128 VGPRs 13 waves leave without
  // spilling -- 64 x 4 spills 8 dwords, 80 x 3 spills 31).  73.7 -> 47.5 us per window layer.  Other 193..240
  // token windows keep the streaming 7 waves x 80 keys form (2 x 3 tiles).
  if constexpr (sizeof(T) == 2 && HD == 80) {
    if (a.Sq == a.Sk && a.Sq > 192 && a.Sq <= 208 && !a.causal && !a.kv_len && !a.q_len) {
      attn_launch_cfg<T, HD, 13, 48, 5>(a, s);
      return;
    }
    if (a.Sq == a.Sk && a.Sq > 192 && a.Sq <= 240 && !a.causal) {
      attn_launch_cfg<T, HD, 7, 80>(a, s);
      return;
    }
    // SAM global attention with the bias in the P buffer and one 64-key tile per grid row: two query blocks per wave
    if (a.Sq >= 1024 && !a.causal && a.rel_p && a.kw == 64 && a.kh * a.kw == a.Sk && a.Sq == a.Sk && !a.kv_len && !a.q_len &&
        !a.q_pos0 && a.o_rs % 4 == 0 && a.o_hs % 4 == 0 && a.o_bs % 4 == 0 && !((uintptr_t)a.O & 15) && a.rel_ld % 2 == 0) {
      // (lab knob: 0 = the general kernel below, 4 / 8 waves; SAM-H, 16 heads, scratch/bench_attn_global.py: 314 us general,
      // 195 us with 4 waves -- two workgroups per CU --, 186 us with 8)
      static const int g2 = getenv("ANYREF_ATTN_G2") ? atoi(getenv("ANYREF_ATTN_G2")) : 8;
      if (g2 == 4 && a.Sq % 128 == 0) {
        attn_g2_launch<T, HD, 4>(a, s);
        return;
      }
      if (g2 == 8 && a.Sq % 256 == 0) {
        attn_g2_launch<T, HD, 8>(a, s);
        return;
      }
    }
    // SAM global attention (4096 tokens): 8 waves share each K/V tile (128 queries per workgroup): 337 -> 311 us;
    // 6 waves 435 us, 4 waves (the default below) 337 us
    if (a.Sq >= 1024 && !a.causal) {
      attn_launch_cfg<T, HD, 8, 64>(a, s);
      return;
    }
  }
  // CLIP ViT-L (257 tokens, head dim 64): 5 query blocks per head, each with ALL keys / values resident (5 tiles of 64,
  // 138 KB) instead of streaming them tile by tile
  if constexpr (sizeof(T) == 2 && HD == 64) {
    static const bool no_res = getenv("ANYREF_ATTN_NO_RES64") != nullptr;
    if (!no_res && a.Sq == a.Sk && a.Sk > 192 && a.Sk <= 320 && !a.causal && !a.kv_len && !a.q_len && !a.rel_h && !a.rel_p &&
        !a.rel_tab_h) {
      attn_launch_cfg<T, HD, 4, 64, 5>(a, s);
      return;
    }
  }
  attn_launch_cfg<T, HD, 4, 0>(a, s);
}

bool attention_takes_rel_tables(int elem_bytes, int hd, int Sq, int Sk, int kh, int kw, bool split) {
  // split: f32 operands multiplied as bf16 pairs (ANYREF_MODE_PARITY16), f32 tables: attn_sp_kernel, any block of a kh x kw window
  if (split) return elem_bytes == 4 && (hd == 64 || hd == 80 || hd == 128) && Sq == Sk && Sq == kh * kw && kh <= 16 && kw <= 16;
  return elem_bytes == 2 && hd == 80 && Sq == Sk && Sq > 192 && Sq <= 208 && kh <= 16 && kw <= 16;  // = attn_launch
}

template <typename T>
void launch_attention(const AttnArgs& a, hipStream_t s) {
  if (a.B <= 0 || a.Sq <= 0) return;
  constexpr int VEC = AMma<T>::VEC;
  if (a.o_split && (sizeof(T) != 4 || a.o_f32 || a.o_rs % 64 || a.o_bs % 64 || a.o_hs % 4))
    throw std::runtime_error("attention: a split-pair output goes with f32 operands and whole-row strides");
  if (a.q_rs % VEC || a.k_rs % VEC || a.v_rs % VEC || a.q_hs % VEC || a.k_hs % VEC || a.v_hs % VEC ||
      a.q_bs % VEC || a.k_bs % VEC || a.v_bs % VEC)
    throw std::runtime_error("attention: strides must be multiples of 16 bytes");
  if constexpr (std::is_same<T, float>::value) {
    if (attn_sp_takes(a)) {
      if (a.hd == 64) attn_sp_launch<64>(a, s);
      else if (a.hd == 80) attn_sp_launch<80>(a, s);
      else attn_sp_launch<128>(a, s);
      return;
    }
  }
  switch (a.hd) {
    case 16: attn_launch<T, 16>(a, s); break;
    case 32: attn_launch<T, 32>(a, s); break;
    case 64: attn_launch<T, 64>(a, s); break;
    case 80: attn_launch<T, 80>(a, s); break;
    case 128: attn_launch<T, 128>(a, s); break;
    default: throw std::runtime_error("attention: unsupported head dim (16/32/64/80/128)");
  }
}
template void launch_attention<float>(const AttnArgs&, hipStream_t);
template void launch_attention<bf16>(const AttnArgs&, hipStream_t);
template void launch_attention<f16>(const AttnArgs&, hipStream_t);

template <typename T, int HD>
__global__ __launch_bounds__(512) void decode_attn_kernel(const float* __restrict__ qkv, const int* __restrict__ pos,
                                                          const float* __restrict__ cs_tab, T* __restrict__ kc,
                                                          T* __restrict__ vc, int maxS, int H, float scale,
                                                          float* __restrict__ out, T* __restrict__ q_keep) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  decode_attn_body<T, HD, false>(qkv, pos, cs_tab, kc, vc, maxS, H, scale, out, q_keep, blockIdx.x, blockIdx.y, sm);
}

template <typename T>
bool launch_decode_attn(const float* qkv, int B, int H, int hd, const int* pos, const float* cs_tab, void* kc,
                        void* vc, int maxS, float scale, float* out, void* q_keep, hipStream_t s) {
  constexpr int VEC = Vec16<T>::N;
  auto go = [&](auto tagHD) {
    constexpr int HD = decltype(tagHD)::value;
    const size_t lds = decode_attn_lds<T, HD>(maxS);
    const double kvbytes = 2.0 * B * H * HD * sizeof(T) * (maxS / 2);  // nominal: half-full cache
    ProfScope prof(sizeof(T) == 2 ? "decode_attn_bf16" : "decode_attn_f32", 0.0, kvbytes, s);
    hipLaunchKernelGGL((decode_attn_kernel<T, HD>), dim3(H, B), dim3(512), lds, s, qkv, pos, cs_tab,
                       reinterpret_cast<T*>(kc), reinterpret_cast<T*>(vc), maxS, H, scale, out,
                       reinterpret_cast<T*>(q_keep));
  };
  if (maxS > 12000) return false;
  if (hd == 128) go(std::integral_constant<int, 128>());
  else if (hd == 64) go(std::integral_constant<int, 64>());
  else if (hd == 32 && sizeof(T) == 2) go(std::integral_constant<int, 32>());
  else if (hd == 32) return false;
  else return false;
  return true;
}
template bool launch_decode_attn<float>(const float*, int, int, int, const int*, const float*, void*, void*, int, float,
                                        float*, void*, hipStream_t);
template bool launch_decode_attn<bf16>(const float*, int, int, int, const int*, const float*, void*, void*, int, float,
                                       float*, void*, hipStream_t);

// ---------------------------------------------------------------------------------------------
// Head-mean attention row of one query (rephrase branch).  One workgroup per batch element.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_row_mean_kernel(const T* __restrict__ q, int64_t q_bs, int64_t q_hs,
                                                            const T* __restrict__ K, int64_t k_bs, int64_t k_rs,
                                                            int64_t k_hs, const int* __restrict__ kv_len, int H,
                                                            int hd, float scale, float* __restrict__ out,
                                                            int ld_out) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = kv_len[b];
  __shared__ float red[4];
  __shared__ float qs[256];
  float* o = out + (int64_t)b * ld_out;
  for (int j = tid; j < n; j += 256) o[j] = 0.f;
  for (int h = 0; h < H; ++h) {
    __syncthreads();
    for (int d = tid; d < hd; d += 256) qs[d] = to_f32<T>(q[(int64_t)b * q_bs + (int64_t)h * q_hs + d]);
    __syncthreads();
    // pass 1: max
    float mx = -INFINITY;
    for (int j = tid; j < n; j += 256) {
      const T* kr = K + (int64_t)b * k_bs + (int64_t)j * k_rs + (int64_t)h * k_hs;
      float s = 0.f;
      for (int d = 0; d < hd; ++d) s = fmaf(qs[d], to_f32<T>(kr[d]), s);
      mx = fmaxf(mx, s * scale);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int j = tid; j < n; j += 256) {
      const T* kr = K + (int64_t)b * k_bs + (int64_t)j * k_rs + (int64_t)h * k_hs;
      float s = 0.f;
      for (int d = 0; d < hd; ++d) s = fmaf(qs[d], to_f32<T>(kr[d]), s);
      sum += expf(s * scale - mx);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = red[0] + red[1] + red[2] + red[3];
    for (int j = tid; j < n; j += 256) {
      const T* kr = K + (int64_t)b * k_bs + (int64_t)j * k_rs + (int64_t)h * k_hs;
      float s = 0.f;
      for (int d = 0; d < hd; ++d) s = fmaf(qs[d], to_f32<T>(kr[d]), s);
      o[j] += expf(s * scale - mx) / sum / (float)H;
    }
  }
}
template <typename T>
void launch_attn_row_mean(const void* q, int64_t q_bs, int64_t q_hs, const void* K, int64_t k_bs, int64_t k_rs,
                          int64_t k_hs, const int* kv_len, int B, int H, int hd, float scale, float* out,
                          int ld_out, hipStream_t s) {
  if (hd > 256) throw std::runtime_error("attn_row_mean: head dim > 256");
  hipLaunchKernelGGL((attn_row_mean_kernel<T>), dim3(B), dim3(256), 0, s, reinterpret_cast<const T*>(q), q_bs,
                     q_hs, reinterpret_cast<const T*>(K), k_bs, k_rs, k_hs, kv_len, H, hd, scale, out, ld_out);
}
template void launch_attn_row_mean<float>(const void*, int64_t, int64_t, const void*, int64_t, int64_t, int64_t,
                                          const int*, int, int, int, float, float*, int, hipStream_t);
template void launch_attn_row_mean<bf16>(const void*, int64_t, int64_t, const void*, int64_t, int64_t, int64_t,
                                         const int*, int, int, int, float, float*, int, hipStream_t);

// ---------------------------------------------------------------------------------------------
// SAM decomposed relative-position bias tables (image_encoder.py:354-392; get_rel_pos with
// q_size == k_size: R[q,k] = tab[q - k + size - 1]).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void rel_pos_kernel(const T* __restrict__ q, int64_t q_bs, int64_t q_rs, int64_t q_hs,
                               const float* __restrict__ tab_h, const float* __restrict__ tab_w, int B, int H,
                               int size, int hd, float* __restrict__ rel_h, float* __restrict__ rel_w) {
  const int S = size * size;
  const int64_t total = (int64_t)B * H * S * size;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(t % size);
    const int i = (int)((t / size) % S);
    const int h = (int)((t / ((int64_t)size * S)) % H);
    const int b = (int)(t / ((int64_t)size * S * H));
    const int y = i / size, x = i % size;
    const T* qr = q + (int64_t)b * q_bs + (int64_t)i * q_rs + (int64_t)h * q_hs;
    const float* th = tab_h + (int64_t)(y - k + size - 1) * hd;
    const float* tw = tab_w + (int64_t)(x - k + size - 1) * hd;
    float ah = 0.f, aw = 0.f;
    for (int c = 0; c < hd; ++c) {
      const float qv = to_f32<T>(qr[c]);
      ah = fmaf(qv, th[c], ah);
      aw = fmaf(qv, tw[c], aw);
    }
    rel_h[t] = ah;
    rel_w[t] = aw;
  }
}
template <typename T>
void launch_rel_pos(const void* q, int64_t q_bs, int64_t q_rs, int64_t q_hs, const float* tab_h,
                    const float* tab_w, int B, int H, int size, int hd, float* rel_h, float* rel_w,
                    hipStream_t s) {
  const int64_t total = (int64_t)B * H * size * size * size;
  if (total <= 0) return;
  const int grid = (int)(cdiv64(total, 256) < 65536 ? cdiv64(total, 256) : 65536);
  hipLaunchKernelGGL((rel_pos_kernel<T>), dim3(grid), dim3(256), 0, s, reinterpret_cast<const T*>(q), q_bs, q_rs,
                     q_hs, tab_h, tab_w, B, H, size, hd, rel_h, rel_w);
}
template void launch_rel_pos<float>(const void*, int64_t, int64_t, int64_t, const float*, const float*, int, int,
                                    int, int, float*, float*, hipStream_t);
template void launch_rel_pos<bf16>(const void*, int64_t, int64_t, int64_t, const float*, const float*, int, int,
                                   int, int, float*, float*, hipStream_t);

}  // namespace anyref

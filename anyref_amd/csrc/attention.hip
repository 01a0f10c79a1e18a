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

#include "kernels.h"

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

// ---- B-operand fragments of P.V from a ROW-major V tile --------------------------------------
// f32: one scalar per lane (B[k = lane>>4][n = lane&15]).
// bf16: the 8 k-values of a lane's column sit in 8 different rows; gfx950's ds_read_b64_tr_b16
// (guide T10) transposes 4 rows x 16 columns per 16-lane group in the LDS read itself, so V is
// staged with plain 16-byte row stores (no 2-byte transposing scatter).  Lane 4q+p of a group
// supplies the address of row q, columns 4p..4p+3; lane i receives column i of the 4 rows.  Two
// reads (rows +0 and +4) make one 8-element fragment.  Reads and their wait live in ONE asm
// statement (guide §5.7 form (i)); EXEC is all ones here (only wave-uniform control flow above).
typedef __attribute__((ext_vector_type(2))) uint32_t uint2v;

template <int ROW4>  // byte offset of 4 rows
__device__ inline void lds_tr_frag1(uint32_t addr, uint4v& f0) {
  uint2v a0, b0;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %2\n\t"
      "ds_read_b64_tr_b16 %1, %2 offset:%3\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(a0), "=&v"(b0)
      : "v"(addr), "i"(ROW4)
      : "memory");
  f0 = uint4v{a0[0], a0[1], b0[0], b0[1]};
}
template <int ROW4>
__device__ inline void lds_tr_frag4(uint32_t addr, uint4v& f0, uint4v& f1, uint4v& f2, uint4v& f3) {
  uint2v a0, b0, a1, b1, a2, b2, a3, b3;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8\n\t"
      "ds_read_b64_tr_b16 %1, %8 offset:%9\n\t"
      "ds_read_b64_tr_b16 %2, %8 offset:32\n\t"
      "ds_read_b64_tr_b16 %3, %8 offset:%10\n\t"
      "ds_read_b64_tr_b16 %4, %8 offset:64\n\t"
      "ds_read_b64_tr_b16 %5, %8 offset:%11\n\t"
      "ds_read_b64_tr_b16 %6, %8 offset:96\n\t"
      "ds_read_b64_tr_b16 %7, %8 offset:%12\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(a0), "=&v"(b0), "=&v"(a1), "=&v"(b1), "=&v"(a2), "=&v"(b2), "=&v"(a3), "=&v"(b3)
      : "v"(addr), "i"(ROW4), "i"(ROW4 + 32), "i"(ROW4 + 64), "i"(ROW4 + 96)
      : "memory");
  f0 = uint4v{a0[0], a0[1], b0[0], b0[1]};
  f1 = uint4v{a1[0], a1[1], b1[0], b1[1]};
  f2 = uint4v{a2[0], a2[1], b2[0], b2[1]};
  f3 = uint4v{a3[0], a3[1], b3[0], b3[1]};
}

// Workgroup = 4 waves; every wave owns RB row-blocks of 16 queries (BQ = 64*RB queries per
// workgroup).  Only RB = 1 is instantiated (see attn_launch).
template <typename T, int HD, int RB>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs a) {
  using M_ = AMma<T>;
  constexpr int KS = M_::KS, VEC = M_::VEC;
  constexpr bool BF = sizeof(T) == 2;
  constexpr int BKV = AttnTile<T>::BKV;
  constexpr int BQ = 64 * RB;
  constexpr int HDK = (HD + KS - 1) / KS * KS;  // QK^T contraction length (zero padded)
  constexpr int LDK = HDK + VEC;                // K tile row stride
  constexpr int LDV = HD + VEC;                 // V tile row stride (row-major [key][d])
  constexpr int LDP = BKV + VEC;
  constexpr int NB = BKV / 16;  // key blocks of one tile
  constexpr int DB = HD / 16;   // output d blocks
  static_assert(HD % 16 == 0, "head dim must be a multiple of 16");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Ks = reinterpret_cast<T*>(smem);
  T* Vs = Ks + BKV * LDK;
  T* Ps = Vs + BKV * LDV;
  float* relh_s = reinterpret_cast<float*>(Ps + 4 * 16 * LDP);
  float* relw_s = relh_s + BQ * a.kh;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * BQ;
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

  // Q fragments (A operand) of this wave's RB row-blocks; row-block rb starts at local row wrow0 + 16*rb
  const int wrow0 = wave * 16 * RB;
  typename M_::Frag qf[RB][HDK / KS];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int qr = q0 + wrow0 + rb * 16 + (lane & 15);
    const bool ok = qr < q_len;
    const T* qrow = Qb + (int64_t)(ok ? qr : 0) * a.q_rs;
#pragma unroll
    for (int kk = 0; kk < HDK / KS; ++kk) qf[rb][kk] = M_::glb(qrow, kk * KS, lane, ok, HD);
  }
  const bool has_rel = a.rel_h != nullptr || a.rel_p != nullptr;
  if (a.rel_p) {
    const float* P = a.rel_p + (int64_t)h * a.rel_hs + ((int64_t)b * a.Sq + q0) * a.rel_ld;
    const int np = a.rel_ld / 2;
    for (int i = tid; i < BQ * a.kh; i += 256) {
      const int r = i / a.kh, c = i % a.kh;
      const int y = (q0 + r) / a.kw;
      relh_s[i] = q0 + r < q_len ? P[(int64_t)r * a.rel_ld + (y - c + a.kh - 1)] : 0.f;
    }
    for (int i = tid; i < BQ * a.kw; i += 256) {
      const int r = i / a.kw, c = i % a.kw;
      const int x = (q0 + r) % a.kw;
      relw_s[i] = q0 + r < q_len ? P[(int64_t)r * a.rel_ld + np + (x - c + a.kw - 1)] : 0.f;
    }
  } else if (has_rel) {
    const float* rh = a.rel_h + ((int64_t)b * a.H + h) * a.Sq * a.kh;
    const float* rw = a.rel_w + ((int64_t)b * a.H + h) * a.Sq * a.kw;
    for (int i = tid; i < BQ * a.kh; i += 256) {
      const int r = i / a.kh, c = i % a.kh;
      relh_s[i] = q0 + r < q_len ? rh[(int64_t)(q0 + r) * a.kh + c] : 0.f;
    }
    for (int i = tid; i < BQ * a.kw; i += 256) {
      const int r = i / a.kw, c = i % a.kw;
      relw_s[i] = q0 + r < q_len ? rw[(int64_t)(q0 + r) * a.kw + c] : 0.f;
    }
  }

  float m_run[RB][4], l_run[RB][4];
  float4v o[RB][DB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      m_run[rb][r] = -INFINITY;
      l_run[rb][r] = 0.f;
    }
#pragma unroll
    for (int d = 0; d < DB; ++d) o[rb][d] = float4v{0.f, 0.f, 0.f, 0.f};
  }

  T* Pw = Ps + wave * 16 * LDP;  // wave-private P staging (C layout -> A operand)

  for (int kt = 0; kt < kv_end; kt += BKV) {
    // ---- stage K [BKV][HDK] and V [BKV][HD], both row-major, 16-byte vectors ------------------
    constexpr int KVEC = HDK / VEC;
    for (int v = tid; v < BKV * KVEC; v += 256) {
      const int row = v / KVEC, d = (v % KVEC) * VEC;
      const int j = kt + row;
      uint4v val = uint4v{0, 0, 0, 0};
      if (j < kv_end && d < HD) val = *reinterpret_cast<const uint4v*>(Kb + (int64_t)j * a.k_rs + d);
      *reinterpret_cast<uint4v*>(&Ks[row * LDK + d]) = val;
    }
    constexpr int VVEC = HD / VEC;
    for (int v = tid; v < BKV * VVEC; v += 256) {
      const int row = v / VVEC, d = (v % VVEC) * VEC;
      const int j = kt + row;
      uint4v val = uint4v{0, 0, 0, 0};
      if (j < kv_end) val = *reinterpret_cast<const uint4v*>(Vb + (int64_t)j * a.v_rs + d);
      *reinterpret_cast<uint4v*>(&Vs[row * LDV + d]) = val;
    }
    __syncthreads();

    int jh[NB], jw[NB];
    if (has_rel) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int j = kt + nb * 16 + (lane & 15);
        jh[nb] = j / a.kw;
        jw[nb] = j % a.kw;
        if (jh[nb] >= a.kh) {  // masked keys beyond kv_len: keep LDS reads in range
          jh[nb] = 0;
          jw[nb] = 0;
        }
      }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int lrow0 = wrow0 + rb * 16;      // first local row of this row-block
      if (q0 + lrow0 >= q_len) continue;      // wave-uniform: nothing to do for rows past the end
      if (a.causal && kt > pos0 + q0 + lrow0 + 15) continue;  // tile entirely in the future of these rows
      // ---- S = Q K^T ------------------------------------------------------------------------
      float4v sc[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        sc[nb] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < HDK / KS; ++kk)
          sc[nb] = M_::mma(qf[rb][kk], M_::lds(&Ks[(nb * 16 + (lane & 15)) * LDK + kk * KS], lane), sc[nb]);
      }
      if constexpr (!BF) {
        // hipcc/ROCm 7.2 under-pads the VALU read of a v_mfma_f32_16x16x4_f32 result on gfx950
        // (40-cycle dependent latency, MI355X_MICROARCH "cycle constants"): the LAST accumulator
        // register (r = 3) was read before the final k-step had landed -- rows 3,7,11,15 of every
        // tile came out slightly wrong.  Pad explicitly before the first VALU use.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15");
        asm volatile("s_nop 15");
        asm volatile("s_nop 15");
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- scale, bias, mask, online softmax ------------------------------------------------
      __builtin_amdgcn_wave_barrier();  // previous row-block's P reads precede these P writes
      float al[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int il = lrow0 + 4 * (lane >> 4) + r, i = q0 + il;
        float sv[NB];  // scores of this row as scalars (never write through a vector-element lvalue)
        float mx = -INFINITY;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int j = kt + nb * 16 + (lane & 15);
          const float raw = sc[nb][r];
          float s = raw * a.scale;
          if (has_rel) s += relh_s[il * a.kh + jh[nb]] + relw_s[il * a.kw + jw[nb]];
          const bool valid = j < kv_len && (!a.causal || j <= pos0 + i);
          s = valid ? s : -INFINITY;
          sv[nb] = s;
          mx = fmaxf(mx, s);
        }
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        const float m_new = fmaxf(m_run[rb][r], mx);
        const float mref = m_new == -INFINITY ? 0.f : m_new;
        al[r] = M_::fexp(m_run[rb][r] - mref);  // m_run = -inf -> 0
        float rs = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const float p = M_::fexp(sv[nb] - mref);  // masked: exp(-inf) = 0
          rs += p;
          Pw[(4 * (lane >> 4) + r) * LDP + nb * 16 + (lane & 15)] = from_f32<T>(p);
        }
        l_run[rb][r] = l_run[rb][r] * al[r] + rs;
        m_run[rb][r] = m_new;
      }
      {
        const float4v av = float4v{al[0], al[1], al[2], al[3]};
#pragma unroll
        for (int d = 0; d < DB; ++d) o[rb][d] *= av;
      }
      // P is wave-private, but the writes must have LANDED before other lanes' data is read back
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      // ---- O += P V -------------------------------------------------------------------------
#pragma unroll
      for (int kk = 0; kk < BKV / KS; ++kk) {
        const typename M_::Frag pf = M_::lds(&Pw[(lane & 15) * LDP + kk * KS], lane);
        if constexpr (BF) {
          // lane -> (row q, 4 columns p) of its 16-lane group's 4x16 block; k-group g = lane>>4
          const uint32_t vaddr =
              (uint32_t)(uintptr_t)(reinterpret_cast<const char*>(Vs) - smem) +
              (uint32_t)((kk * 32 + 8 * (lane >> 4) + ((lane & 15) >> 2)) * LDV + 4 * (lane & 3)) * 2u;
          constexpr int ROW4 = 4 * LDV * 2;
          uint4v vf[DB];
          if constexpr (DB >= 4) lds_tr_frag4<ROW4>(vaddr, vf[0], vf[1], vf[2], vf[3]);
          if constexpr (DB == 8) lds_tr_frag4<ROW4>(vaddr + 128, vf[4], vf[5], vf[6], vf[7]);
          if constexpr (DB == 5) lds_tr_frag1<ROW4>(vaddr + 128, vf[4]);
          if constexpr (DB < 4) {
#pragma unroll
            for (int d = 0; d < DB; ++d) lds_tr_frag1<ROW4>(vaddr + 32 * d, vf[d]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int d = 0; d < DB; ++d) o[rb][d] = M_::mma(pf, __builtin_bit_cast(short8, vf[d]), o[rb][d]);
        } else {
#pragma unroll
          for (int d = 0; d < DB; ++d) {
            const typename M_::Frag vf = Vs[(kk * KS + (lane >> 4)) * LDV + d * 16 + (lane & 15)];
            o[rb][d] = M_::mma(pf, vf, o[rb][d]);
          }
        }
      }
    }
    __syncthreads();  // K/V tiles free for the next iteration
  }

  // ---- normalise + store --------------------------------------------------------------------
  T* Ob = reinterpret_cast<T*>(a.O) + (int64_t)b * a.o_bs + (int64_t)h * a.o_hs;
  float* Obf = reinterpret_cast<float*>(a.O) + (int64_t)b * a.o_bs + (int64_t)h * a.o_hs;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float l = l_run[rb][r];
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) l += __shfl_xor(l, off, 64);
      const int i = q0 + wrow0 + rb * 16 + 4 * (lane >> 4) + r;
      if (i < q_len) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          const int64_t off = (int64_t)i * a.o_rs + d * 16 + (lane & 15);
          if (a.o_f32)
            Obf[off] = o[rb][d][r] * inv;
          else
            Ob[off] = from_f32<T>(o[rb][d][r] * inv);
        }
      }
    }
  }
}

template <typename T, int HD, int RB>
static void attn_launch_rb(const AttnArgs& a, hipStream_t s) {
  constexpr int KS = AMma<T>::KS, VEC = AMma<T>::VEC, BKV = AttnTile<T>::BKV;
  constexpr int HDK = (HD + KS - 1) / KS * KS, LDK = HDK + VEC;
  constexpr int LDV = HD + VEC, LDP = BKV + VEC, BQ = 64 * RB;
  size_t lds = sizeof(T) * (BKV * LDK + BKV * LDV + 4 * 16 * LDP);
  if (a.rel_h || a.rel_p) lds += sizeof(float) * BQ * (a.kh + a.kw);
  if (lds > 160 * 1024) throw std::runtime_error("attention: LDS budget exceeded");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<T, HD, RB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  dim3 grid(cdiv(a.Sq, BQ), a.H, a.B);
  static const std::string tag = std::string(sizeof(T) == 2 ? "attn_bf16_hd" : "attn_f32_hd") + std::to_string(HD);
  const double flops = 4.0 * a.B * a.H * (double)a.Sq * a.Sk * HD * (a.causal ? 0.5 : 1.0);
  const double bytes = (double)a.B * a.H * HD * sizeof(T) * (2.0 * a.Sq + 2.0 * a.Sk);
  ProfScope prof(tag.c_str(), flops, bytes, s);
  hipLaunchKernelGGL((attn_kernel<T, HD, RB>), grid, dim3(256), lds, s, a);
}

template <typename T, int HD>
static void attn_launch(const AttnArgs& a, hipStream_t s) {
  // One 16-row block per wave.  RB = 2 / 4 (more MFMAs per staged K/V tile) was measured SLOWER on
  // MI355X: 164-255 VGPRs leave one wave per SIMD and the loop is issue/latency- not staging-bound
  // (SAM window 81 -> 87 / 133 us, SAM global 694 -> 1010 us per layer).
  attn_launch_rb<T, HD, 1>(a, s);
}

template <typename T>
void launch_attention(const AttnArgs& a, hipStream_t s) {
  if (a.B <= 0 || a.Sq <= 0) return;
  constexpr int VEC = AMma<T>::VEC;
  if (a.q_rs % VEC || a.k_rs % VEC || a.v_rs % VEC || a.q_hs % VEC || a.k_hs % VEC || a.v_hs % VEC ||
      a.q_bs % VEC || a.k_bs % VEC || a.v_bs % VEC)
    throw std::runtime_error("attention: strides must be multiples of 16 bytes");
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

// ---------------------------------------------------------------------------------------------
// Decode-step attention (one query per sequence), fused with RoPE + KV-cache append.  Latency
// bound: one 256-thread workgroup per (head, sequence); LPK = hd/VEC lanes share one key (16-byte
// loads straight from the cache), KPI = 256/LPK keys per sweep.
// ---------------------------------------------------------------------------------------------
template <typename T, int HD>
__global__ __launch_bounds__(512) void decode_attn_kernel(const float* __restrict__ qkv, const int* __restrict__ pos,
                                                          const float* __restrict__ cs_tab, T* __restrict__ kc,
                                                          T* __restrict__ vc, int maxS, int H, float scale,
                                                          float* __restrict__ out, T* __restrict__ q_keep) {
  constexpr int NT = 512, NWV = NT / 64, UN = 4;
  constexpr int VEC = Vec16<T>::N, LPK = HD / VEC, KPI = NT / LPK, HALF = HD / 2;
  static_assert(LPK <= 64 && (LPK & (LPK - 1)) == 0, "lanes per key must be a power of two within a wave");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* q_s = sm;                 // [HD] rotated, T-rounded, pre-scaled query
  float* k_s = q_s + HD;           // [HD] this step's key (as stored in the cache)
  float* v_s = k_s + HD;           // [HD]
  float* part = v_s + HD;          // [KPI][HD]
  float* red = part + KPI * HD;    // [2*NWV]
  float* sc = red + 2 * NWV;       // [maxS] scores / probabilities
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.x, b = blockIdx.y;
  const int p = pos[b], n = p + 1;
  const float* row = qkv + (int64_t)b * 3 * H * HD;
  const int64_t cbase = ((int64_t)b * maxS * H + h) * HD;   // + j*H*HD for key j

  if (tid < HALF) {
    const int d = tid;
    const float cs = cs_tab[((int64_t)p * 2) * HALF + d], sn = cs_tab[((int64_t)p * 2 + 1) * HALF + d];
    const float q1 = row[h * HD + d], q2 = row[h * HD + d + HALF];
    const float k1 = row[(H + h) * HD + d], k2 = row[(H + h) * HD + d + HALF];
    const T qa = from_f32<T>(q1 * cs - q2 * sn), qb = from_f32<T>(q2 * cs + q1 * sn);
    const T ka = from_f32<T>(k1 * cs - k2 * sn), kb = from_f32<T>(k2 * cs + k1 * sn);
    const T va = from_f32<T>(row[(2 * H + h) * HD + d]), vb = from_f32<T>(row[(2 * H + h) * HD + d + HALF]);
    const int64_t co = cbase + (int64_t)p * H * HD;
    kc[co + d] = ka; kc[co + d + HALF] = kb;
    vc[co + d] = va; vc[co + d + HALF] = vb;
    if (q_keep) { q_keep[co + d] = qa; q_keep[co + d + HALF] = qb; }
    q_s[d] = to_f32<T>(qa) * scale; q_s[d + HALF] = to_f32<T>(qb) * scale;
    k_s[d] = to_f32<T>(ka); k_s[d + HALF] = to_f32<T>(kb);
    v_s[d] = to_f32<T>(va); v_s[d + HALF] = to_f32<T>(vb);
  }
  __syncthreads();
  const int sub = tid % LPK, slice = tid / LPK;
  float qf[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) qf[i] = q_s[sub * VEC + i];
  // ---- scores: UN keys per thread in flight (the loop is latency-, not bandwidth-bound) ----
  for (int j0 = 0; j0 < n; j0 += KPI * UN) {
    uint4v kv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = j0 + u * KPI + slice;
      kv[u] = j < p ? *reinterpret_cast<const uint4v*>(kc + cbase + (int64_t)j * H * HD + sub * VEC)
                    : uint4v{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = j0 + u * KPI + slice;
      float kf[VEC];
      Vec16<T>::unpack(kv[u], kf);
      if (j == p) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) kf[i] = k_s[sub * VEC + i];
      }
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) dot = fmaf(qf[i], kf[i], dot);
#pragma unroll
      for (int o = 1; o < LPK; o <<= 1) dot += __shfl_xor(dot, o, 64);
      if (sub == 0 && j < n) sc[j] = dot;
    }
  }
  __syncthreads();
  // ---- softmax over sc[0,n) ----
  float mx = -INFINITY;
  for (int j = tid; j < n; j += NT) mx = fmaxf(mx, sc[j]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int w = 1; w < NWV; ++w) mx = fmaxf(mx, red[w]);
  float sum = 0.f;
  for (int j = tid; j < n; j += NT) {
    const float e = expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[NWV + wave] = sum;
  __syncthreads();
  sum = 0.f;
#pragma unroll
  for (int w = 0; w < NWV; ++w) sum += red[NWV + w];
  // ---- O = P V ----
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  for (int j0 = 0; j0 < n; j0 += KPI * UN) {
    uint4v vv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = j0 + u * KPI + slice;
      vv[u] = j < p ? *reinterpret_cast<const uint4v*>(vc + cbase + (int64_t)j * H * HD + sub * VEC)
                    : uint4v{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = j0 + u * KPI + slice;
      float vf[VEC];
      Vec16<T>::unpack(vv[u], vf);
      if (j == p) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) vf[i] = v_s[sub * VEC + i];
      }
      const float pj = j < n ? sc[j] : 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = fmaf(pj, vf[i], acc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) part[slice * HD + sub * VEC + i] = acc[i];
  __syncthreads();
  if (tid < HD) {
    float o = 0.f;
#pragma unroll
    for (int sI = 0; sI < KPI; ++sI) o += part[sI * HD + tid];
    out[((int64_t)b * H + h) * HD + tid] = o / sum;
  }
}

template <typename T>
bool launch_decode_attn(const float* qkv, int B, int H, int hd, const int* pos, const float* cs_tab, void* kc,
                        void* vc, int maxS, float scale, float* out, void* q_keep, hipStream_t s) {
  constexpr int VEC = Vec16<T>::N;
  auto go = [&](auto tagHD) {
    constexpr int HD = decltype(tagHD)::value;
    constexpr int KPI = 512 / (HD / VEC);
    const size_t lds = sizeof(float) * (3 * HD + KPI * HD + 16 + maxS + KPI * 4);
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

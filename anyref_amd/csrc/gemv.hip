// Decode GEMV (weight streaming, HBM-bound) and the skinny f32 GEMV of the mask decoder / [SEG] hand-off.
// Split from gemm.hip so that the two compile (and are iterated on) independently.
#include <cstdlib>
#include <stdexcept>
#include <type_traits>

#include "kernels.h"

namespace anyref {

// ANYREF_GEMV_GRID=n: decode GEMV workgroups (measurement knob, read once; never set in production)
// ANYREF_GEMV_PAIR=0: one wave per row group (the round-2 work split) instead of wave pairs sharing a group's K chunks
static bool gemv_pair_knob() {
  static const bool p = !(getenv("ANYREF_GEMV_PAIR") && atoi(getenv("ANYREF_GEMV_PAIR")) == 0);
  return p;
}
static int gemv_grid_knob() {
  static const int g = getenv("ANYREF_GEMV_GRID") ? atoi(getenv("ANYREF_GEMV_GRID")) : 0;
  return g;
}

// ---------------------------------------------------------------------------------------------
// Decode GEMV: weight-streaming, HBM-bound.  One workgroup (8 waves) stages the (optionally
// RMS-normalised) activation rows in LDS as T, then every wave streams R weight rows at a time
// with 16-byte loads straight to VGPRs (no LDS round trip for once-read weights:
// cdna_hip_programming.md §5 "GEMV / M <= 16" row) and reduces across the wave.
// ---------------------------------------------------------------------------------------------
// W8: weights are fp8 e4m3 bytes with one f32 scale per output row (weight-only quantisation, activations
// stay T = bf16): 16 weights per 16-byte load, converted two at a time by v_cvt_pk_f32_fp8, the row scale
// applied once to the reduced sum.  Half the HBM bytes per step of the bf16 stream.
// PAIR: waves 2i and 2i + 1 of a workgroup share their row groups, each sweeping every second K chunk; the odd wave
// hands its partial sums to the even one through LDS (one barrier per group, fixed order: deterministic).  With one
// wave per row group N = 4096 (o_proj / down_proj at 7B) is 2048 row pairs for the 4096 waves of the grid: half of
// them idle, half the bytes in flight per CU (kernel-side stamps: median workgroup 14.7 us of a 17.5 us down_proj
// launch).  Used for those shapes only (gemv_dispatch).
#ifndef ANYREF_GEMV_NO_DOT2
#define ANYREF_GEMV_NO_DOT2 0
#endif
constexpr bool NO_DOT2 = ANYREF_GEMV_NO_DOT2;
template <typename T, int NB, bool DUAL, int XPT, bool W8 = false, bool PAIR = false>  // XPT: x elements per thread in registers, K <= 512 * XPT
__global__ __launch_bounds__(512) void gemv_kernel(GemvArgs a, int b0, int nb) {
  static_assert(!W8 || sizeof(T) == 2, "fp8 weights go with bf16 activations");
  // T = sp16 (ANYREF_MODE_PARITY16): bf16 weights, exactly as stored, against the f32 activation row kept in LDS as f32
  // (16 - 44 KB) -- same bytes per decode step as the bf16 mode, f32 products and sums
  using WE = std::conditional_t<is_split<T>::value, bf16, T>;   // weight element
  using XE = std::conditional_t<is_split<T>::value, float, T>;  // staged activation element
  using WT = std::conditional_t<W8, uint8_t, WE>;
  constexpr int VN = W8 ? 16 : Vec16<WE>::N;  // weights per 16-byte load
  constexpr int R = DUAL ? 1 : 2;   // output rows per wave per pass
  constexpr int RW = 2;             // weight rows streamed per pass (DUAL: gate row + up row)
  // 16-byte loads per row in flight per lane.  8 measured slower; 2 (68 instead of 100 VGPRs, so that a GEMV
  // workgroup fits beside a resident 256^2 GEMM workgroup of the co-running SAM stream) measured equal within
  // noise, alone and under the overlap
  constexpr int UNR = 4;
  // packed bf16 dot products (v_dot2c_f32_bf16) for every batch size of the bf16 mode, batch 1 included; the f32 and the
  // split-pair (sp16) builds keep the unpack + f32 FMA chain
  constexpr bool DOT2 = std::is_same<T, bf16>::value && !W8 && !NO_DOT2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  XE* xs = reinterpret_cast<XE*>(smem);  // [NB][K]
  __shared__ float red[NB][8];
  __shared__ float red2[2][4][2][NB];  // PAIR: partial sums of the odd waves, double-buffered over the groups
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K;
  // kernel-side timestamps (kernels.h: StampArgs): off in production (one uniform branch)
  __shared__ unsigned long long st_t[2];
  __shared__ unsigned st_cnt;
  unsigned long long t_begin = 0;
  if (a.stamp.base) {
    t_begin = wall_clock64();
    if (tid == 0) {
      st_t[0] = ~0ull;
      st_t[1] = 0;
      st_cnt = 0;
    }
  }

  const WT* __restrict__ W = reinterpret_cast<const WT*>(a.W);
  const WT* __restrict__ W2 = reinterpret_cast<const WT*>(a.W2);
  const int ldw = a.ldw > 0 ? a.ldw : K;  // 2K: gate / up rows interleaved in one matrix
  const int nwaves = gridDim.x * 8;
  const int gw = blockIdx.x * 8 + wave;
  const int ngroups = cdiv(a.N, R);
  constexpr int CH = 64 * VN * UNR;  // K elements one wave sweeps per chunk
  const int nch = cdiv(K, CH);
  // Flattened (row group, K chunk) work list of this wave, software-pipelined one chunk deep: the
  // loads of item t+1 are in flight while item t is multiplied (and while x is being staged).
  const int unit = PAIR ? gw >> 1 : gw, nunits = PAIR ? nwaves >> 1 : nwaves, half = PAIR ? (gw & 1) : 0;
  const int nchp = PAIR ? (nch + 1) >> 1 : nch;  // chunk slots per wave and group
  // (PAIR: every wave walks the same number of groups -- they all meet at the hand-off barrier)
  const int my_groups = PAIR ? cdiv(ngroups, nunits) : (gw < ngroups ? (ngroups - gw + nwaves - 1) / nwaves : 0);
  const int items = my_groups * nchp;
  uint4v wcur[UNR][RW], wnxt[UNR][RW];
  auto load_item = [&](int t, uint4v (&w)[UNR][RW]) {
    const int g = unit + (t / nchp) * nunits, c = PAIR ? 2 * (t % nchp) + half : t % nchp;
    const int n0 = g * R;
    const bool live = !PAIR || (g < ngroups && c < nch);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int k = live ? c * CH + u * 64 * VN + lane * VN : K;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int n = n0 + r < a.N ? n0 + r : a.N - 1;
        w[u][r] = k < K ? __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(W + (int64_t)n * ldw + k))
                        : uint4v{0, 0, 0, 0};
      }
      if (DUAL)
        w[u][RW - 1] = k < K ? __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(W2 + (int64_t)n0 * ldw + k))
                             : uint4v{0, 0, 0, 0};
    }
  };
  // x goes FIRST into the (in-order) vector-memory queue: its wait then leaves the weight prefetch
  // issued right behind it in flight.  Issued the other way round, the x wait also waited for the
  // first weight chunk (measured: x staged 4-9 us into a 10-22 us kernel).
  if (!a.gain) {
    // no RMSNorm (o_proj / down_proj inputs): a plain f32 -> T copy with 16-byte loads.  The register
    // path below issues XPT predicated dword loads per thread; at K = 11008 (XPT = 24) that staging
    // alone cost ~3 us of a 21 us kernel.  Same values, same rounding: bit-identical.
    constexpr int XV = (XPT + 3) / 4;  // float4 per thread
    float4v xv[NB][XV];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float* x = a.x + (int64_t)(b0 + (b < nb ? b : 0)) * a.ldx;
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int k = (tid + i * 512) * 4;
        xv[b][i] = (b < nb && k < K) ? *reinterpret_cast<const float4v*>(x + k) : float4v{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (items > 0) load_item(0, wcur);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b >= nb) continue;
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int k = (tid + i * 512) * 4;
        if (k < K) store4_from_f32<XE>(&xs[b * K + k], xv[b][i][0], xv[b][i][1], xv[b][i][2], xv[b][i][3]);  // one LDS store
      }
    }
  } else {
    // RMSNorm path: x and the gain as 16-byte loads (XV per thread and row; launch_gemv checks the alignment) -- dword
    // loads were XPT per row: 40 load instructions per thread at 4 batch rows in front of the first weight load
    constexpr int XV = (XPT + 3) / 4;  // float4 per thread
    float4v xr[NB][XV], gr[XV];
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int k = (tid + i * 512) * 4;
      gr[i] = k < K ? *reinterpret_cast<const float4v*>(a.gain + k) : float4v{1.f, 1.f, 1.f, 1.f};
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float* x = a.x + (int64_t)(b0 + (b < nb ? b : 0)) * a.ldx;
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int k = (tid + i * 512) * 4;
        xr[b][i] = (b < nb && k < K) ? *reinterpret_cast<const float4v*>(x + k) : float4v{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (items > 0) load_item(0, wcur);
    // sums of squares of ALL batch rows behind ONE barrier (a barrier per row: NB dependent LDS round trips per launch)
    float scale[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < XV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += xr[b][i][e] * xr[b][i][e];
      ss = wave_sum(ss);
      if (lane == 0) red[b][wave] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) tot += red[b][w];
      scale[b] = rsqrtf(tot / (float)K + a.eps);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b >= nb) continue;
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int k = (tid + i * 512) * 4;
        if (k < K) {
          const float4v v = xr[b][i] * scale[b] * gr[i];
          store4_from_f32<XE>(&xs[b * K + k], v[0], v[1], v[2], v[3]);
          // the normalised row itself is an output of the step (last-layer hidden state before lm_head)
          if (a.xn_out && blockIdx.x == 0)
            *reinterpret_cast<float4v*>(a.xn_out + (int64_t)(a.xn_row_map ? a.xn_row_map[b0 + b] : b0 + b) * a.xn_ld + k) = v;
        }
      }
    }
}
  __syncthreads();

  float acc[RW][NB];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
  for (int t = 0; t < items; ++t) {
    if (t + 1 < items) load_item(t + 1, wnxt);
    const int ci = t % nchp, c = PAIR ? 2 * ci + half : ci;
    const bool live = !PAIR || (unit + (t / nchp) * nunits < ngroups && c < nch);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int k = live ? c * CH + u * 64 * VN + lane * VN : K;
      if (k < K) {
        if constexpr (DOT2) {
          // 16-bit weights with more than one batch row: both operands stay PACKED (8 bf16 per 16 bytes) and go through
          // v_dot2c_f32_bf16, 4 instructions per row pair and batch row instead of 16 unpacks + 8 FMAs -- with 4 batch
          // rows per pass the unpack + FMA form keeps the VALU ~60 % busy and the step is no longer HBM-bound
          // (decode step at batch 4: 4.1 ms against 2.9 ms at batch 1)
          typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
          uint4v xv[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b)
            xv[b] = b < nb ? *reinterpret_cast<const uint4v*>(&xs[b * K + k]) : uint4v{0, 0, 0, 0};
#pragma unroll
          for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const uint32_t wj = wcur[u][r][j], xj = xv[b][j];
                acc[r][b] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, wj), __builtin_bit_cast(bf16x2, xj),
                                                            acc[r][b], false);
              }
        } else {
        float xf[NB][VN];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (b < nb) {
            constexpr int XV = Vec16<XE>::N;  // x elements per 16-byte LDS read
#pragma unroll
            for (int h = 0; h < VN / XV; ++h) {
              const uint4v xv = *reinterpret_cast<const uint4v*>(&xs[b * K + k + h * XV]);
              Vec16<XE>::unpack(xv, &xf[b][h * XV]);
            }
          } else {
#pragma unroll
            for (int i = 0; i < VN; ++i) xf[b][i] = 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
          float wf[VN];
          if constexpr (W8) unpack_fp8x16(wcur[u][r], wf);
          else Vec16<WE>::unpack(wcur[u][r], wf);
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < VN; ++i) acc[r][b] = fmaf(wf[i], xf[b][i], acc[r][b]);
        }
        }
      }
    }
    if (ci == nchp - 1) {  // row group finished: reduce across the wave and store
      const int n0 = (unit + (t / nchp) * nunits) * R;
#pragma unroll
      for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[r][b] = wave_sum(acc[r][b]);
      if constexpr (PAIR) {
        const int buf = (t / nchp) & 1;
        if (half == 1 && lane == 0) {
#pragma unroll
          for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int b = 0; b < NB; ++b) red2[buf][wave >> 1][r][b] = acc[r][b];
        }
        __syncthreads();
        if (half == 0) {
#pragma unroll
          for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[r][b] += red2[buf][wave >> 1][r][b];
        }
      }
      // Every lane holds every reduced sum (xor butterfly): lane i < R * NB finishes output (r, b) = (i / NB, i % NB) --
      // scale / bias / activation / residual load / store side by side.  One lane walking the R * NB outputs is a chain of
      // dependent residual load -> store round trips (y may alias the residual, so the compiler keeps their order): 8 per
      // row group at 4 batch rows, on the wave's critical path at the end of the launch.
      if (lane < R * NB && half == 0) {
        const int r = lane / NB, b = lane % NB, n = n0 + r;
        float v = 0.f, v2 = 0.f;
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
#pragma unroll
          for (int bb = 0; bb < NB; ++bb)
            if (lane == rr * NB + bb) {
              v = acc[rr][bb];
              v2 = DUAL ? acc[RW - 1][bb] : 0.f;
            }
        if (n < a.N && b < nb) {
          if constexpr (W8) {
            v *= a.wscale[(int64_t)n * a.ws_stride];
            if (DUAL) v2 *= a.wscale2[(int64_t)n * a.ws_stride];
          }
          if (a.bias) v += a.bias[n];
          if (DUAL)
            v = apply_act(v, ACT_SILU) * v2;
          else
            v = apply_act(v, a.act);
          const int64_t o = (int64_t)(b0 + b) * a.ldy + n;
          if (a.resid) v += a.resid[o];
          a.y[o] = v;
        }
      }
#pragma unroll
      for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
    }
    if (t + 1 < items) {
#pragma unroll
      for (int u = 0; u < UNR; ++u)
#pragma unroll
        for (int r = 0; r < RW; ++r) wcur[u][r] = wnxt[u][r];
    }
  }
  if (a.stamp.base && lane == 0) {
    // every wave folds its span into the workgroup's (LDS atomics; the init is ordered by the barrier after the x
    // stage); the wave whose count comes back last has seen all of them and writes the workgroup's slot
    atomicMin(&st_t[0], t_begin);
    atomicMax(&st_t[1], (unsigned long long)wall_clock64());
    if (atomicAdd(&st_cnt, 1u) == 7u) {
      const int e = *a.stamp.epoch;
      if (e < a.stamp.max_epoch) {
        unsigned long long* p = a.stamp.base + (size_t)e * a.stamp.stride + (size_t)blockIdx.x * 2;
        p[0] = st_t[0];
        p[1] = st_t[1];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Decode GEMV for 5 - 8 batch rows (bf16 activations; bf16 or fp8 weights), ONE pass over the weights.
//
// The packed-dot kernel above multiplies a weight row by every batch row on the VALU and re-reads the batch rows' x chunks from
// LDS for every 16 bytes of weights: at 8 rows that is VALU- and LDS-bound (8-row packed dot, fp8: 1.0 - 1.5 TB/s; two 4-row
// passes: every byte twice), and the tiled GEMM (M = 8 rows of a 64-row tile) streamed the decode rows' weights at 2 TB/s.
// Here the products go to the matrix cores WITHOUT giving up the GEMV's row-contiguous loads: v_mfma_f32_4x4x4 (sixteen
// independent 4 x 4 x 4 blocks per wave instruction) takes lane 4 q + j as (row j, K positions of block q), so a 16-byte load
// per lane is FOUR weight rows x 256 contiguous bytes (16 lanes per row) -- the MFMA-native 16 x 16 x 32 layout is 16 rows x 64
// bytes per load instruction and streams at ~4 TB/s (round 3 / 4 negative results).  Block q multiplies its K positions of the
// four rows by the same K positions of four batch rows (B operand: 8 bytes per lane from the LDS x stage, two MFMAs cover
// eight batch rows) and keeps a 4 x 4 partial sum; the sixteen blocks are added by an xor butterfly at the end of a row group.
// fp8 weights are widened to packed bf16 pairs by v_cvt_scalef32_pk_bf16_fp8 (scale 1, exact): 8 VALU + 8 MFMA instructions per
// 16-byte load instead of 144 VALU.  x stage: row stride = 64 bytes modulo 256 and (fp8) the two 16-byte halves of a block's 16
// positions in two planes, so the 16 lanes of a read phase (4 batch rows x 4 blocks) tile the banks.
// K longer than the LDS stage (down_proj: 11008 / 13824) is taken as TWO K halves inside the launch (x restaged behind a
// barrier, the sums of the wave's one row group stay in registers); inputs with an RMSNorm (K <= 6144) in one.
// Weights are LOADED row-contiguous (lane 16 j + q: 256 bytes of a row per 16 lanes) and turned into the MFMA's lane order by
// four ds_bpermute per 16-byte load; three chunk buffers (this item + two in flight, 8 KB per wave each) rotate through moves.
// Measured (scratch/bench_gemv8.py under rocprofv3, cold weights, 8 rows, us per launch; 13B shapes):
//   fp8  qkv 27.9 (2.8 TB/s)  o 14.3 - 23  gate/up 41.5 (3.4 TB/s)  down 30 (two K halves);  two 4-row passes: 80 / 43 / 122 / 74
//   bf16 qkv 34.8 (4.5 TB/s)  gate/up 57.8 (4.9 TB/s);  two 4-row passes: 79 / 112
// The main loop streams at the HBM rate (time = bytes / 6 TB/s + ~12 us): 2.5 us launch + 5 us x stage of 8 rows + ~3 us first
// chunk + tail.  C5 (13B fp8, batch 8): 186.0 -> 181.1 ms per batch against the split-K GEMM decode path (M = 8 rows of a
// 64 / 128-row tile at 1.9 TB/s); 7B bf16 at batch 8: decode step 4.13 -> 4.15 ms alone, 132.4 -> 128.5 ms per call with masks.
// ---------------------------------------------------------------------------------------------
template <bool W8, bool DUAL, int XV>  // XV: float4 of x per thread and row in the RMSNorm staging (K <= 2048 XV)
__global__ __launch_bounds__(512) void gemv_rows8_kernel(GemvArgs a, int b0, int nb, int kh0, int rs) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  constexpr int NB = 8, UNR = 8;
  constexpr int EPL = W8 ? 16 : 8;       // weights per lane and load
  constexpr int SEG = 16 * EPL;          // K positions one load instruction covers per row (16 lanes)
  constexpr int CH = UNR * SEG;          // ... one chunk
  using WT = std::conditional_t<W8, uint8_t, bf16>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float red[NB][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 3, q = lane >> 2;
  const int K = a.K;
  const int ldw = a.ldw > 0 ? a.ldw : K;
  const WT* __restrict__ W = reinterpret_cast<const WT*>(a.W);
  const WT* __restrict__ W2 = reinterpret_cast<const WT*>(a.W2);
  constexpr int R = DUAL ? 2 : 4;  // outputs per row group
  const int ngroups = cdiv(a.N, R);
  const int nwaves = gridDim.x * 8, gw = blockIdx.x * 8 + wave;
  const int my_groups = gw < ngroups ? (ngroups - gw + nwaves - 1) / nwaves : 0;
  const int nkh = kh0 < K ? 2 : 1;
  const int nch0 = cdiv(kh0, CH), nch1 = nkh == 2 ? cdiv(K - kh0, CH) : 0;
  // items in (K half, group, chunk) order, one chunk prefetched
  const int items0 = my_groups * nch0, items = items0 + my_groups * nch1;
  const int pl = rs > 0 ? (rs - 64) / 2 : 0;  // plane size in bytes (fp8: two planes per row)
  uint4v w0[UNR], w1[UNR], w2[UNR];  // this item's weights and the next two items' in flight (8 KB per wave each)
  auto item_pos = [&](int t, int& g, int& kbeg, int& kend) __attribute__((always_inline)) {
    int gi, c;
    if (t < items0) {
      gi = t / nch0; c = t - gi * nch0;
      kbeg = c * CH; kend = kh0;
    } else {
      const int t1 = t - items0;
      gi = t1 / nch1; c = t1 - gi * nch1;
      kbeg = kh0 + c * CH; kend = K;
    }
    g = gw + gi * nwaves;
  };
  auto load_item = [&](int t, uint4v (&w)[UNR]) __attribute__((always_inline)) {
    int g, kbeg, kend;
    item_pos(t, g, kbeg, kend);
    // LOADED row-contiguous: lane 16 jl + ql takes 16 bytes of row jl at block ql (16 lanes = 256 contiguous bytes: four 64-byte
    // requests; with the MFMA's own lane order, 4 q + j, every 16-byte piece of a wave load is a request of its own and the
    // launch streams at 1.3 - 1.8 TB/s) and turned into the MFMA order by ds_bpermute when it is used (mma_item)
    // rows 4 g .. 4 g + 3, or (SwiGLU) gate rows 2 g, 2 g + 1 and up rows 2 g, 2 g + 1
    const int jl = lane >> 4, ql = lane & 15;
    int n = DUAL ? g * 2 + (jl & 1) : g * 4 + jl;
    n = n < a.N ? n : a.N - 1;
    const WT* row = ((DUAL && jl >= 2) ? W2 : W) + (int64_t)n * ldw;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int k = kbeg + u * SEG + ql * EPL;
      w[u] = k < kend ? __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(row + k)) : uint4v{0, 0, 0, 0};
    }
  };
  // LDS position (bytes) of K position kl (multiple of 4) of batch row b inside the current stage
  auto xpos = [&](int b, int kl) __attribute__((always_inline)) {
    if constexpr (W8) return b * rs + ((kl >> 3) & 1) * pl + (kl >> 4) * 16 + (kl & 7) * 2;
    else return b * rs + kl * 2;
  };
  // (x goes FIRST into the in-order vector-memory queue, as in gemv_kernel: behind two chunks of weights the stage would wait for
  //  an HBM round trip it does not depend on)
  auto first_chunks = [&]() __attribute__((always_inline)) {
    if (items > 0) load_item(0, w0);
    if (items > 1) load_item(1, w1);
  };
  float scale[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) scale[b] = 1.f;
  if (a.gain) {
    // RMSNorm inputs (one K stage): rows through registers, sums of squares of all rows behind one barrier
    float4v xr[NB][XV], gr[XV];
#pragma unroll
    for (int i = 0; i < XV; ++i) {
      const int k = (tid + i * 512) * 4;
      gr[i] = k < K ? *reinterpret_cast<const float4v*>(a.gain + k) : float4v{1.f, 1.f, 1.f, 1.f};
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float* x = a.x + (int64_t)(b0 + (b < nb ? b : 0)) * a.ldx;
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int k = (tid + i * 512) * 4;
        xr[b][i] = (b < nb && k < K) ? *reinterpret_cast<const float4v*>(x + k) : float4v{0.f, 0.f, 0.f, 0.f};
      }
    }
    first_chunks();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < XV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += xr[b][i][e] * xr[b][i][e];
      ss = wave_sum(ss);
      if (lane == 0) red[b][wave] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) tot += red[b][w];
      scale[b] = rsqrtf(tot / (float)K + a.eps);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int k = (tid + i * 512) * 4;
        if (k < K) {
          const float4v v = xr[b][i] * scale[b] * gr[i];  // rows >= nb: zeros
          store4_from_f32<bf16>(reinterpret_cast<bf16*>(smem + xpos(b, k)), v[0], v[1], v[2], v[3]);
          if (a.xn_out && blockIdx.x == 0 && b < nb)
            *reinterpret_cast<float4v*>(a.xn_out + (int64_t)(a.xn_row_map ? a.xn_row_map[b0 + b] : b0 + b) * a.xn_ld + k) = v;
        }
      }
    }
  }
  auto stage_plain = [&](int kbase, int klen, bool then_weights) __attribute__((always_inline)) {  // f32 -> bf16 copy of x[:, kbase : kbase + klen) (no norm)
    // four rows at a time, every load of the four in flight before the first LDS store (a load -> store loop over rows and
    // 2048-column pieces is ~24 dependent L2 round trips: 6 us of a 17 us o_proj launch)
    constexpr int NI = 5;  // float4 per thread and row: stage <= 9472 columns
    const int n4 = klen / 4;
#pragma unroll
    for (int bb = 0; bb < NB; bb += 4) {
      float4v r[4][NI];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float* x = a.x + (int64_t)(b0 + (bb + b < nb ? bb + b : 0)) * a.ldx + kbase;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int c = tid + i * 512;
          r[b][i] = (bb + b < nb && c < n4) ? *reinterpret_cast<const float4v*>(x + c * 4) : float4v{0.f, 0.f, 0.f, 0.f};
        }
      }
      if (then_weights && bb == 0) first_chunks();
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int c = tid + i * 512;
          if (c < n4) store4_from_f32<bf16>(reinterpret_cast<bf16*>(smem + xpos(bb + b, c * 4)), r[b][i][0], r[b][i][1], r[b][i][2], r[b][i][3]);
        }
    }
  };
  if (!a.gain) stage_plain(0, kh0, true);
  __syncthreads();

  float4v ac[2];  // [batch rows 0-3 | 4-7] of the current row group (two-K-half launches: the wave's only group)
  ac[0] = ac[1] = float4v{0.f, 0.f, 0.f, 0.f};
  auto mma_item = [&](const uint4v (&wcur)[UNR], int kbeg_l, int kend_l) __attribute__((always_inline)) {  // kbeg_l: chunk start inside the current stage
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      // (no branch: a masked load is sixteen zero bytes and every lane must take part in the permute; positions past the end of
      //  the stage read the x of position 0 -- finite values against zero weights)
      int kl = kbeg_l + u * SEG + q * EPL;
      kl = kl < kend_l ? kl : 0;
      {
        // x of this block's positions for batch rows j and j + 4
        uint4v xa[2][W8 ? 2 : 1];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          xa[h][0] = *reinterpret_cast<const uint4v*>(smem + xpos(j + 4 * h, kl));
          if constexpr (W8) xa[h][1] = *reinterpret_cast<const uint4v*>(smem + xpos(j + 4 * h, kl + 8));
        }
        // weights of (row j, block q) from the lane that loaded them
        uint32_t wt[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) wt[e] = (uint32_t)__builtin_amdgcn_ds_bpermute((16 * j + q) * 4, (int)wcur[u][e]);
#pragma unroll
        for (int m = 0; m < EPL / 4; ++m) {
          short4v_ af;
          if constexpr (W8) {
            const uint32_t wj = wt[m];
            const bf16x2 lo = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(wj, 1.0f, false);
            const bf16x2 hi = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(wj, 1.0f, true);
            const uint2v t = uint2v{__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi)};
            af = __builtin_bit_cast(short4v_, t);
          } else {
            const uint32_t w0 = wt[2 * m], w1 = wt[2 * m + 1];
            const uint2v t = uint2v{w0, w1};
            af = __builtin_bit_cast(short4v_, t);
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const uint32_t x0 = xa[h][m >> 1][2 * (m & 1)], x1 = xa[h][m >> 1][2 * (m & 1) + 1];
            const uint2v t = uint2v{x0, x1};
            ac[h] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(af, __builtin_bit_cast(short4v_, t), ac[h], 0, 0, 0);
          }
        }
      }
    }
  };
  auto finish_group = [&](int g) __attribute__((always_inline)) {
    // add the sixteen blocks' partial sums (lanes of equal j), then lane l < 8 finishes batch row (l & 3) + 4 (l >> 2)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = ac[h][r];
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        ac[h][r] = v;
      }
    if (lane < 8) {
      const int b = (lane & 3) + 4 * (lane >> 2);
      const float4v t = (lane >> 2) ? ac[1] : ac[0];
      float v[4] = {t[0], t[1], t[2], t[3]};  // D[i = weight row][this lane's batch row]
      if (b < nb) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int n = g * R + r;
          if (n >= a.N) continue;
          float o = v[r], o2 = DUAL ? v[r + 2] : 0.f;
          if constexpr (W8) {
            o *= a.wscale[(int64_t)n * a.ws_stride];
            if (DUAL) o2 *= a.wscale2[(int64_t)n * a.ws_stride];
          }
          if (a.bias) o += a.bias[n];
          if (DUAL) o = apply_act(o, ACT_SILU) * o2;
          else o = apply_act(o, a.act);
          const int64_t off = (int64_t)(b0 + b) * a.ldy + n;
          if (a.resid) o += a.resid[off];
          a.y[off] = o;
        }
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) ac[h] = float4v{0.f, 0.f, 0.f, 0.f};
  };
  // (one copy of the multiply code: the three buffers rotate through register moves, 64 v_mov per 64 MFMAs)
  auto run = [&](int t_begin, int t_end) __attribute__((always_inline)) {
    for (int t = t_begin; t < t_end; ++t) {
      if (t + 2 < items) load_item(t + 2, w2);  // (chunks of the second K half are requested before its x is staged)
      int g, kbeg, kend;
      item_pos(t, g, kbeg, kend);
      const bool second = t >= items0;
      const int base = second ? kh0 : 0;
      mma_item(w0, kbeg - base, kend - base);
      const int nch = second ? nch1 : nch0, tt = second ? t - items0 : t;
      const bool last_chunk = tt % nch == nch - 1;
      if (last_chunk && (nkh == 1 || second)) finish_group(g);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        w0[u] = w1[u];
        w1[u] = w2[u];
      }
    }
  };
  run(0, items0);
  if (nkh == 2) {  // every wave of the workgroup, with or without a row group of its own
    __syncthreads();
    stage_plain(kh0, K - kh0, false);
    __syncthreads();
    run(items0, items);
  }
}

// 5 - 8 rows of a bf16 GEMV in one pass (gemv_rows8_kernel); false: shape not taken (the caller falls back to passes of 4)
static bool gemv_rows8_launch(const GemvArgs& a, int b0, int nb, hipStream_t s) {
  static const bool off = getenv("ANYREF_GEMV_ROWS8") && atoi(getenv("ANYREF_GEMV_ROWS8")) == 0;
  if (off) return false;
  const bool w8 = a.w_fp8 != 0, dual = a.W2 != nullptr;
  const int epl = w8 ? 16 : 8, ch = 8 * 16 * epl;
  const int ldw = a.ldw > 0 ? a.ldw : a.K;
  const size_t wsz = w8 ? 1 : 2;
  if (a.K % epl || (ldw * wsz) % 16 || ((uintptr_t)a.W & 15) || (dual && ((uintptr_t)a.W2 & 15)) || a.K % 4) return false;
  constexpr int KH_MAX = 9472;  // 8 rows x (2 KH_MAX rounded to 256 + 64) bytes <= 150 KB
  int kh0 = a.K;
  if (a.gain) {
    if (a.K > 6144) return false;  // (rows through registers: 24 float4 per thread)
  } else if (a.K > KH_MAX) {
    kh0 = cdiv(cdiv(a.K, 2), ch) * ch;
    if (kh0 > KH_MAX || kh0 >= a.K) return false;
    if (cdiv(a.N, dual ? 2 : 4) > 256 * 8) return false;  // a wave keeps the sums of its ONE row group across the halves
  }
  const int rs = cdiv(kh0 * 2, 256) * 256 + 64;
  const size_t lds = (size_t)8 * rs;
  const int grid = 256;
  const double wbytes = (double)a.N * a.K * wsz * (dual ? 2 : 1) + (double)nb * (a.K + a.N) * 4;
  char tag[48];
  snprintf(tag, sizeof(tag), "gemv_rows8_%s%s", w8 ? "fp8w" : "bf16", dual ? "_swiglu" : "");
  ProfScope prof(tag, 2.0 * nb * a.N * (double)a.K * (dual ? 2 : 1), wbytes, s);
  auto launch = [&](auto w8_t, auto dual_t, auto xv_t) {
    constexpr bool W8 = decltype(w8_t)::value, DUAL = decltype(dual_t)::value;
    constexpr int XV = decltype(xv_t)::value;
    auto kern = &gemv_rows8_kernel<W8, DUAL, XV>;
    static KernelAttrOnce once;
    ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), 152 * 1024);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a, b0, nb, kh0, rs);
  };
  using TT = std::true_type;
  using FF = std::false_type;
  using X2 = std::integral_constant<int, 2>;
  using X3 = std::integral_constant<int, 3>;
  auto by_xv = [&](auto w8_t, auto dual_t) {
    if (a.gain && a.K > 4096) launch(w8_t, dual_t, X3());
    else launch(w8_t, dual_t, X2());
  };
  if (w8) {
    if (dual) by_xv(TT(), TT());
    else by_xv(TT(), FF());
  } else {
    if (dual) by_xv(FF(), TT());
    else by_xv(FF(), FF());
  }
  return true;
}

template <typename T, int NB>
static void gemv_dispatch(const GemvArgs& a_in, int b0, int nb, hipStream_t s) {
  GemvArgs a = a_in;
  const size_t lds = (size_t)NB * a.K * (is_split<T>::value ? 4 : sizeof(T));
  if (lds > 150 * 1024) throw std::runtime_error("gemv: K too large for the LDS activation stage");
  // one or two 8-wave workgroups per CU depending on the LDS the activation stage needs
  // (512 workgroups measured best for N*K of 34-262 MB; 256 / 1024 / 2048 were 3-30 % slower)
  int grid = 256 * (lds > 76 * 1024 ? 1 : 2);
  const int grid_rule = grid;  // (the wave-pair decision below goes by this one: the same sums whatever grid is asked for)
  if (a.grid > 0 && a.grid < grid) grid = a.grid;
  if (gemv_grid_knob() > 0) grid = gemv_grid_knob();
  auto go = [&](auto xpt_tag) {
    constexpr int XPT = decltype(xpt_tag)::value;
    // algorithmic bytes: every weight element once (+ the tiny activation / output vectors)
    const double wsz = a.w_fp8 ? 1.0 : (is_split<T>::value ? 2.0 : (double)sizeof(T));
    const double wbytes = (double)a.N * a.K * wsz * (a.W2 ? 2 : 1) + (double)nb * (a.K + a.N) * 4;
    // one tag per kernel instantiation, so a tag's average can be checked against rocprofv3's per-kernel one
    char tag[40];
    snprintf(tag, sizeof(tag), "gemv_%s%s_x%d", a.w_fp8 ? "fp8w" : (is_split<T>::value ? "sp16" : sizeof(T) == 2 ? "bf16" : "f32"),
             a.W2 ? "_swiglu" : "", XPT);
    ProfScope prof(tag, 2.0 * nb * a.N * (double)a.K * (a.W2 ? 2 : 1), wbytes, s);
    if (g_stamp && g_stamp->on) a.stamp = g_stamp->slot(tag, wbytes, grid);
    auto launch = [&](auto dual_t, auto w8_t, auto pair_t) {
      constexpr bool DUAL = decltype(dual_t)::value, W8 = decltype(w8_t)::value, PAIR = decltype(pair_t)::value;
      auto kern = &gemv_kernel<T, NB, DUAL, XPT, W8, PAIR>;
      static KernelAttrOnce once;  // per instantiation, per device
      ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), 150 * 1024);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a, b0, nb);
    };
    using TT = std::true_type;
    using FF = std::false_type;
    // wave pairs where single waves would leave half of the grid without a row group (N = 4096 at 7B: o_proj 8.5 ->
    // 8.05 us, down_proj 17.45 -> 16.7 us); with more groups than waves the plain split is faster (qkv 17.9 vs 18.8 us,
    // gate/up 30.9 vs 31.6: the hand-off barrier per group costs more than the better balance returns)
    const bool pair = gemv_pair_knob() && cdiv(a.N, a.W2 ? 1 : 2) * 2 <= (gemv_grid_knob() > 0 ? grid : grid_rule) * 8;
    auto by_pair = [&](auto dual_t, auto w8_t) {
      if (pair) launch(dual_t, w8_t, TT());
      else launch(dual_t, w8_t, FF());
    };
    if constexpr (sizeof(T) == 2) {
      if (a.w_fp8) {
        if (a.W2) by_pair(TT(), TT());
        else by_pair(FF(), TT());
        return;
      }
    }
    if (a.W2) by_pair(TT(), FF());
    else by_pair(FF(), FF());
  };
  if (a.K <= 512 * 8)
    go(std::integral_constant<int, 8>());
  else if (a.K <= 512 * 24)
    go(std::integral_constant<int, 24>());
  else if (a.K <= 512 * 32)
    go(std::integral_constant<int, 32>());
  else
    throw std::runtime_error("gemv: K > 16384 not supported");
}

template <typename T>
void launch_gemv(const GemvArgs& a, hipStream_t s) {
  const int VN = a.w_fp8 ? 16 : Vec16<std::conditional_t<is_split<T>::value, bf16, T>>::N;
  if (a.K % VN || ((uintptr_t)a.W & 15)) throw std::runtime_error("gemv: K must be a multiple of 16 bytes");
  if (a.w_fp8 && (sizeof(T) != 2 || !a.wscale || (a.W2 && !a.wscale2)))
    throw std::runtime_error("gemv: fp8 weights need the bf16 mode and per-row scales");
  if (((uintptr_t)a.x & 15) || a.ldx % 4 || a.K % 4 || (a.gain && ((uintptr_t)a.gain & 15)) ||
      (a.xn_out && (((uintptr_t)a.xn_out & 15) || a.xn_ld % 4)))
    throw std::runtime_error("gemv: x / gain / xn_out rows must be 16-byte aligned");
  constexpr int NBMAX = sizeof(T) == 2 ? 4 : 2;
  for (int b0 = 0; b0 < a.B;) {
    const int left = a.B - b0;
    if constexpr (std::is_same<T, bf16>::value) {
      // 5 - 8 rows left: one pass over the weights on the 4 x 4 x 4 MFMA form where the shape allows
      if (left > 4 && gemv_rows8_launch(a, b0, left < 8 ? left : 8, s)) {
        b0 += left < 8 ? left : 8;
        continue;
      }
    }
    const int nb = left < NBMAX ? left : NBMAX;
    if (nb == 1)
      gemv_dispatch<T, 1>(a, b0, nb, s);
    else if (nb == 2)
      gemv_dispatch<T, 2>(a, b0, nb, s);
    else
      gemv_dispatch<T, NBMAX>(a, b0, nb, s);
    b0 += nb;
  }
}
template void launch_gemv<float>(const GemvArgs&, hipStream_t);

// Skinny f32 GEMMs of the mask decoder and the [SEG] hand-off (M <= 8 token rows, f32 weights in every mode)
// through the weight-streaming kernel above: the 64 x 64 MFMA tile kernel runs them as 4-64 workgroups walking
// K in 16-wide steps, a chain of dependent global-load round trips (text_hidden_fcs[0], 4096 x 4096 f32 = 64 MB
// on 64 workgroups: 146 us; here every weight row is one wave's 16-byte loads and the grid covers the rows).
void launch_gemv_skinny_f32(const GemvArgs& a, hipStream_t s) {
  if (a.B < 1 || a.B > 8 || a.K > 4096 || a.K % 4 || a.W2 || a.w_fp8 || a.gain || ((uintptr_t)a.W & 15) ||
      ((uintptr_t)a.x & 15) || a.ldx % 4)
    throw std::runtime_error("gemv_skinny_f32: 1..8 rows, K <= 4096, K % 4 == 0, 16-byte aligned rows");
  const int ngroups = cdiv(a.N, 2);
  const int grid = ngroups >= 2048 ? 512 : cdiv(ngroups, 4);  // a wave pair per row group
  auto go = [&](auto nb_tag) {
    constexpr int NB = decltype(nb_tag)::value;
    auto kern = &gemv_kernel<float, NB, false, 8, false, true>;
    static KernelAttrOnce once;
    ensure_dyn_lds(once, reinterpret_cast<const void*>(kern), 150 * 1024);
    ProfScope prof("gemv_f32_skinny", 2.0 * a.B * a.N * (double)a.K, (double)a.N * a.K * 4 + (double)a.B * (a.K + a.N) * 4,
                   s);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), (size_t)NB * a.K * sizeof(float), s, a, 0, a.B);
  };
  if (a.B == 1) go(std::integral_constant<int, 1>());
  else if (a.B == 2) go(std::integral_constant<int, 2>());
  else if (a.B <= 4) go(std::integral_constant<int, 4>());
  else go(std::integral_constant<int, 8>());
}
template void launch_gemv<bf16>(const GemvArgs&, hipStream_t);
template void launch_gemv<sp16>(const GemvArgs&, hipStream_t);

}  // namespace anyref

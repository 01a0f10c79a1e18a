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
  for (int b0 = 0; b0 < a.B; b0 += NBMAX) {
    const int nb = a.B - b0 < NBMAX ? a.B - b0 : NBMAX;
    if (nb == 1)
      gemv_dispatch<T, 1>(a, b0, nb, s);
    else if (nb == 2)
      gemv_dispatch<T, 2>(a, b0, nb, s);
    else
      gemv_dispatch<T, NBMAX>(a, b0, nb, s);
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

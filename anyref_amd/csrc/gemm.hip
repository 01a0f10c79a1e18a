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
#include <cstdlib>
#include <stdexcept>
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
template <typename T, int BM, int BN>
__device__ inline void gemm_epilogue(const GemmArgs& a, float4v (&acc)[BM / 32][BN / 32], int m0, int n0, int z,
                                     int lane, int wr, int wc) {
  constexpr int MI = BM / 32, NI = BN / 32;
  const float* bias = a.bias ? a.bias + (int64_t)z * a.sBias : nullptr;
  const float* resid = a.resid ? a.resid + (int64_t)z * a.sR : nullptr;
  float* Cf = reinterpret_cast<float*>(a.C) + (int64_t)z * a.sC;
  T* Ct = reinterpret_cast<T*>(a.C) + (int64_t)z * a.sC;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wr * (BM / 2) + i * 16 + (lane >> 4) * 4 + r;
      if (m >= a.M) continue;
      const int dm = a.row_map ? a.row_map[m] : m;
      if (dm < 0) continue;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + wc * (BN / 2) + j * 16 + (lane & 15);
        if (n >= a.N) continue;
        float v = acc[i][j][r] * a.alpha;
        if (bias) v += bias[n];
        v = apply_act(v, a.act);
        if (resid) v += resid[(int64_t)dm * a.ldr + n];
        if (a.c_f32)
          Cf[(int64_t)dm * a.ldc + n] = v;
        else
          Ct[(int64_t)dm * a.ldc + n] = from_f32<T>(v);
      }
    }
  }
}

template <typename T, int BM, int BN, int BK>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
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
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
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
        for (int j = 0; j < NI; ++j) acc[i][j] = M_::mma(af[i], bf[j], acc[i][j]);
    }
    __syncthreads();
  }

  gemm_epilogue<T, BM, BN>(a, acc, m0, n0, z, lane, wr, wc);
}

// ---------------------------------------------------------------------------------------------
// bf16 main GEMM: BM x 128 x 64 tile, operands staged HBM -> LDS by global_load_lds (16 B/lane,
// no VGPR round trip), two LDS buffers, one barrier per K tile (guide §5.5 "minimum 2-phase").
// LDS rows are 128 B; a wave-instruction fills 8 rows linearly, so the bank-conflict swizzle
// (16-B chunk index ^ (row & 7)) is applied to the per-lane SOURCE address and again on the
// fragment read (guide rule 21).  Rows past M / N are clamped to the last valid row (their
// results are never stored); K must be a multiple of 64 here (launcher falls back otherwise).
// Tiles are walked M-fastest inside an XCD-contiguous chunk of the grid so that the blocks that
// share a weight panel hit the same L2 (T1, bijective remap).
// ---------------------------------------------------------------------------------------------
template <int BM>
__global__ __launch_bounds__(256) void gemm_glds_kernel(GemmArgs a, int tiles_m, int tiles_n) {
  constexpr int BN = 128, BK = 64;
  constexpr int MI = BM / 32, NI = BN / 32;
  constexpr int A_INST = BM / 8, B_INST = BN / 8, PER_WAVE = (A_INST + B_INST) / 4;
  constexpr int TILE_A = BM * BK, TILE_B = BN * BK;  // elements
  __shared__ __attribute__((aligned(16))) bf16 lds[2 * (TILE_A + TILE_B)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware, bijective block remap; then M-fastest tile order
  const int nwg = tiles_m * tiles_n;
  int id = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = id % 8;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8;
  }
  const int pm = id % tiles_m, pn = id / tiles_m;
  const int m0 = pm * BM, n0 = pn * BN;
  const int z = blockIdx.z;
  const bf16* __restrict__ A = reinterpret_cast<const bf16*>(a.A) + (int64_t)z * a.sA;
  const bf16* __restrict__ W = reinterpret_cast<const bf16*>(a.W) + (int64_t)z * a.sW;

  // per-lane source pointers of this wave's staging instructions (tile 0), advanced by BK per tile
  const bf16* src[PER_WAVE];
  int dst[PER_WAVE];  // LDS element offset of the instruction's 1 KiB destination within a buffer
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int inst = wave + 4 * i;  // A instructions first, then B
    const bool isA = inst < A_INST;
    const int li = isA ? inst : inst - A_INST;
    const int r = li * 8 + (lane >> 3), pc = lane & 7, lc = pc ^ (r & 7);
    if (isA) {
      int gm = m0 + r;
      gm = gm < a.M ? gm : a.M - 1;
      src[i] = A + (int64_t)gm * a.lda + lc * 8;
      dst[i] = li * 512;
    } else {
      int gn = n0 + r;
      gn = gn < a.N ? gn : a.N - 1;
      src[i] = W + (int64_t)gn * a.ldw + lc * 8;
      dst[i] = TILE_A + li * 512;
    }
  }
  auto stage = [&](int buf) {
    bf16* base = lds + buf * (TILE_A + TILE_B);
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                       (__attribute__((address_space(3))) void*)(base + dst[i]), 16, 0, 0);
      src[i] += BK;
    }
  };

  float4v acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

  const int nt = a.K / BK;
  stage(0);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) stage(buf ^ 1);
    const bf16* As = lds + buf * (TILE_A + TILE_B);
    const bf16* Bs = As + TILE_A;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      const int pc = ((ks * 4 + (lane >> 4)) ^ (lane & 7)) * 8;  // swizzled 16-B chunk of this lane
      short8 af[MI], bfr[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        af[i] = *reinterpret_cast<const short8*>(As + (wr * (BM / 2) + i * 16 + (lane & 15)) * BK + pc);
#pragma unroll
      for (int j = 0; j < NI; ++j)
        bfr[j] = *reinterpret_cast<const short8*>(Bs + (wc * (BN / 2) + j * 16 + (lane & 15)) * BK + pc);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  gemm_epilogue<bf16, BM, BN>(a, acc, m0, n0, z, lane, wr, wc);
}

static bool launch_gemm_glds(const GemmArgs& a, hipStream_t s) {
  // measured on MI355X (scratch/bench_gemm.py): the glds kernel wins when at least one full wave of
  // 128x128 tiles exists; below that the register-staged kernel's smaller tiles fill the chip better
  if (a.K % 64 || a.K < 64) return false;
  const int tn = cdiv(a.N, 128);
  const bool bm128 = true;
  if ((int64_t)cdiv(a.M, 128) * tn * a.batch < 256) return false;
  const int tm = cdiv(a.M, 128);
  const double flops = 2.0 * a.M * a.N * (double)a.K * a.batch;
  const double bytes = ((double)a.M * a.K + (double)a.N * a.K) * 2 * a.batch + (double)a.M * a.N * (a.c_f32 ? 4 : 2) * a.batch;
  ProfScope prof(bm128 ? "gemm_bf16_glds_128x128" : "gemm_bf16_glds_64x128", flops, bytes, s);
  dim3 grid(tm * tn, 1, a.batch);
  if (bm128)
    hipLaunchKernelGGL((gemm_glds_kernel<128>), grid, dim3(256), 0, s, a, tm, tn);
  else
    hipLaunchKernelGGL((gemm_glds_kernel<64>), grid, dim3(256), 0, s, a, tm, tn);
  return true;
}

template <typename T>
void launch_gemm(const GemmArgs& a, hipStream_t s) {
  constexpr int VEC = Mma<T>::VEC;
  constexpr int BK = sizeof(T) == 2 ? 64 : 16;
  if (a.M <= 0 || a.N <= 0) return;
  if (a.K % VEC || a.lda % VEC || a.ldw % VEC || ((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15) ||
      (a.sA % VEC) || (a.sW % VEC))
    throw std::runtime_error("gemm: K/lda/ldw must be multiples of 16 bytes and operands 16-byte aligned");
  static const bool force_old = getenv("ANYREF_GEMM_OLD") != nullptr;  // A/B switch for microbenchmarks
  if (sizeof(T) == 2 && a.M >= 48 && !force_old && launch_gemm_glds(a, s)) return;
  // tile choice: fewest padded rows first, then enough workgroups to cover the 256 CUs
  const int waste128 = cdiv(a.M, 128) * 128 - a.M, waste64 = cdiv(a.M, 64) * 64 - a.M;
  bool bm128 = waste128 <= waste64 + 16;
  int bn = 128;
  auto blocks = [&](int bm, int bnn) { return (int64_t)cdiv(a.M, bm) * cdiv(a.N, bnn) * a.batch; };
  if (blocks(bm128 ? 128 : 64, 128) < 256) {
    if (bm128 && blocks(64, 128) >= 2 * blocks(128, 128)) bm128 = false;
    if (blocks(bm128 ? 128 : 64, 128) < 256) bn = 64;
  }
  dim3 block(256);
  const double flops = 2.0 * a.M * a.N * (double)a.K * a.batch;
  const double bytes = ((double)a.M * a.K + (double)a.N * a.K) * sizeof(T) * a.batch +
                       (double)a.M * a.N * (a.c_f32 ? 4 : sizeof(T)) * a.batch;
  const char* tag = sizeof(T) == 2 ? (bm128 ? (bn == 128 ? "gemm_bf16_128x128" : "gemm_bf16_128x64")
                                            : (bn == 128 ? "gemm_bf16_64x128" : "gemm_bf16_64x64"))
                                   : (bm128 ? (bn == 128 ? "gemm_f32_128x128" : "gemm_f32_128x64")
                                            : (bn == 128 ? "gemm_f32_64x128" : "gemm_f32_64x64"));
  ProfScope prof(tag, flops, bytes, s);
  if (bm128 && bn == 128) {
    dim3 grid(cdiv(a.N, 128), cdiv(a.M, 128), a.batch);
    hipLaunchKernelGGL((gemm_kernel<T, 128, 128, BK>), grid, block, 0, s, a);
  } else if (bm128) {
    dim3 grid(cdiv(a.N, 64), cdiv(a.M, 128), a.batch);
    hipLaunchKernelGGL((gemm_kernel<T, 128, 64, BK>), grid, block, 0, s, a);
  } else if (bn == 128) {
    dim3 grid(cdiv(a.N, 128), cdiv(a.M, 64), a.batch);
    hipLaunchKernelGGL((gemm_kernel<T, 64, 128, BK>), grid, block, 0, s, a);
  } else {
    dim3 grid(cdiv(a.N, 64), cdiv(a.M, 64), a.batch);
    hipLaunchKernelGGL((gemm_kernel<T, 64, 64, BK>), grid, block, 0, s, a);
  }
}
template void launch_gemm<float>(const GemmArgs&, hipStream_t);
template void launch_gemm<bf16>(const GemmArgs&, hipStream_t);

// ---------------------------------------------------------------------------------------------
// Decode GEMV: weight-streaming, HBM-bound.  One workgroup (8 waves) stages the (optionally
// RMS-normalised) activation rows in LDS as T, then every wave streams R weight rows at a time
// with 16-byte loads straight to VGPRs (no LDS round trip for once-read weights:
// cdna_hip_programming.md §5 "GEMV / M <= 16" row) and reduces across the wave.
// ---------------------------------------------------------------------------------------------
template <typename T, int NB, bool DUAL>
__global__ __launch_bounds__(512) void gemv_kernel(GemvArgs a, int b0, int nb) {
  constexpr int VN = Vec16<T>::N;
  constexpr int R = DUAL ? 1 : 2;   // output rows per wave per pass
  constexpr int RW = 2;             // weight rows streamed per pass (DUAL: gate row + up row)
  constexpr int UNR = 4;            // 16-byte loads per row in flight per lane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* xs = reinterpret_cast<T*>(smem);  // [NB][K]
  __shared__ float red[NB][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K;

  const T* __restrict__ W = reinterpret_cast<const T*>(a.W);
  const T* __restrict__ W2 = reinterpret_cast<const T*>(a.W2);
  const int nwaves = gridDim.x * 8;
  const int gw = blockIdx.x * 8 + wave;
  const int ngroups = cdiv(a.N, R);
  constexpr int CH = 64 * VN * UNR;  // K elements one wave sweeps per chunk
  const int nch = cdiv(K, CH);
  // Flattened (row group, K chunk) work list of this wave, software-pipelined one chunk deep: the
  // loads of item t+1 are in flight while item t is multiplied (and while x is being staged).
  const int my_groups = gw < ngroups ? (ngroups - gw + nwaves - 1) / nwaves : 0;
  const int items = my_groups * nch;
  uint4v wcur[UNR][RW], wnxt[UNR][RW];
  auto load_item = [&](int t, uint4v (&w)[UNR][RW]) {
    const int g = gw + (t / nch) * nwaves, c = t % nch;
    const int n0 = g * R;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int k = c * CH + u * 64 * VN + lane * VN;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int n = n0 + r < a.N ? n0 + r : a.N - 1;
        w[u][r] = k < K ? __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(W + (int64_t)n * K + k))
                        : uint4v{0, 0, 0, 0};
      }
      if (DUAL)
        w[u][RW - 1] = k < K ? __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(W2 + (int64_t)n0 * K + k))
                             : uint4v{0, 0, 0, 0};
    }
  };
  if (items > 0) load_item(0, wcur);

  // stage x (with fused RMSNorm) while the first weight chunk is in flight --------------------
  for (int b = 0; b < nb; ++b) {
    const float* x = a.x + (int64_t)(b0 + b) * a.ldx;
    float scale = 1.f;
    if (a.gain) {
      float ss = 0.f;
      for (int k = tid; k < K; k += 512) {
        float v = x[k];
        ss += v * v;
      }
      ss = wave_sum(ss);
      if (lane == 0) red[b][wave] = ss;
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) tot += red[b][w];
      scale = rsqrtf(tot / (float)K + a.eps);
    }
    for (int k = tid; k < K; k += 512) {
      float v = x[k] * scale;
      if (a.gain) v *= a.gain[k];
      xs[b * K + k] = from_f32<T>(v);
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
    const int c = t % nch;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
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
        for (int r = 0; r < RW; ++r) {
          float wf[VN];
          Vec16<T>::unpack(wcur[u][r], wf);
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < VN; ++i) acc[r][b] = fmaf(wf[i], xf[b][i], acc[r][b]);
        }
      }
    }
    if (c == nch - 1) {  // row group finished: reduce across the wave and store
      const int n0 = (gw + (t / nch) * nwaves) * R;
#pragma unroll
      for (int r = 0; r < RW; ++r)
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
            if (a.bias) v += a.bias[n];
            if (DUAL)
              v = apply_act(v, ACT_SILU) * acc[RW - 1][b];
            else
              v = apply_act(v, a.act);
            const int64_t o = (int64_t)(b0 + b) * a.ldy + n;
            if (a.resid) v += a.resid[o];
            a.y[o] = v;
          }
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
}

template <typename T, int NB>
static void gemv_dispatch(const GemvArgs& a, int b0, int nb, hipStream_t s) {
  const size_t lds = (size_t)NB * a.K * sizeof(T);
  if (lds > 150 * 1024) throw std::runtime_error("gemv: K too large for the LDS activation stage");
  // one or two 8-wave workgroups per CU depending on the LDS the activation stage needs
  static const int grid_mul = getenv("ANYREF_GEMV_GRID") ? atoi(getenv("ANYREF_GEMV_GRID")) : 2;  // tuning knob
  const int grid = 256 * (lds > 76 * 1024 ? 1 : grid_mul);
  static bool attr_set = false;  // per instantiation
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_kernel<T, NB, true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_kernel<T, NB, false>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  // algorithmic bytes: every weight element once (+ the tiny activation / output vectors)
  const double wbytes = (double)a.N * a.K * sizeof(T) * (a.W2 ? 2 : 1) + (double)nb * (a.K + a.N) * 4;
  ProfScope prof(sizeof(T) == 2 ? (a.W2 ? "gemv_bf16_swiglu" : "gemv_bf16") : (a.W2 ? "gemv_f32_swiglu" : "gemv_f32"),
                 2.0 * nb * a.N * (double)a.K * (a.W2 ? 2 : 1), wbytes, s);
  if (a.W2)
    hipLaunchKernelGGL((gemv_kernel<T, NB, true>), dim3(grid), dim3(512), lds, s, a, b0, nb);
  else
    hipLaunchKernelGGL((gemv_kernel<T, NB, false>), dim3(grid), dim3(512), lds, s, a, b0, nb);
}

template <typename T>
void launch_gemv(const GemvArgs& a, hipStream_t s) {
  constexpr int VN = Vec16<T>::N;
  if (a.K % VN || ((uintptr_t)a.W & 15)) throw std::runtime_error("gemv: K must be a multiple of 16 bytes");
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
template void launch_gemv<bf16>(const GemvArgs&, hipStream_t);

}  // namespace anyref

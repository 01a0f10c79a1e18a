// GEMM kernel lab: variants of the bf16 MFMA GEMM, timed and checked against each other on the hot shapes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I anyref_amd/csrc -o scratch/bin/gemm_lab scratch/lab/gemm_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "common.h"
using namespace anyref;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct GA { const bf16* A; const bf16* W; bf16* C; int M, N, K; };
#include <utility>
template <int... Is, typename F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }

// ---------------- V0: the shipped structure (single LDS buffer, write-before-barrier, 2 barriers per tile) --------------
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void v0(GA a) {
  constexpr int BK = 64, VEC = 8, LD = BK + VEC, MI = BM / 32, NI = BN / 32, KV = BK / VEC;
  constexpr int AV = BM * KV / 256, WV = BN * KV / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* As = reinterpret_cast<bf16*>(smem);
  bf16* Ws = As + BM * LD;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN), nwg = tiles_m * tiles_n;
  int id = blockIdx.x;
  { const int q = nwg / 8, r = nwg % 8, xcd = id % 8; id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8; }
  const int m0 = (id % tiles_m) * BM, n0 = (id / tiles_m) * BN;
  float4v acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = float4v{0, 0, 0, 0};
  uint4v ra[AV], rw[WV];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int v = tid + i * 256, row = v / KV, kv = v % KV, gm = m0 + row, gk = k0 + kv * VEC;
      ra[i] = (gm < a.M && gk < a.K) ? *reinterpret_cast<const uint4v*>(a.A + (int64_t)gm * a.K + gk) : uint4v{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = tid + i * 256, row = v / KV, kv = v % KV, gn = n0 + row, gk = k0 + kv * VEC;
      rw[i] = (gn < a.N && gk < a.K) ? *reinterpret_cast<const uint4v*>(a.W + (int64_t)gn * a.K + gk) : uint4v{0, 0, 0, 0};
    }
  };
  auto sstore = [&](bf16* Ad, bf16* Wd) {
#pragma unroll
    for (int i = 0; i < AV; ++i) { const int v = tid + i * 256, row = v / KV, kv = v % KV; *reinterpret_cast<uint4v*>(&Ad[row * LD + kv * VEC]) = ra[i]; }
#pragma unroll
    for (int i = 0; i < WV; ++i) { const int v = tid + i * 256, row = v / KV, kv = v % KV; *reinterpret_cast<uint4v*>(&Wd[row * LD + kv * VEC]) = rw[i]; }
  };
  auto compute = [&](const bf16* Ad, const bf16* Wd) {
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      short8 af[MI], bfr[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const short8*>(&Ad[(wr * (BM / 2) + i * 16 + (lane & 15)) * LD + ks * 32 + 8 * (lane >> 4)]);
#pragma unroll
      for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const short8*>(&Wd[(wc * (BN / 2) + j * 16 + (lane & 15)) * LD + ks * 32 + 8 * (lane >> 4)]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  gload(0);
  for (int k0 = 0; k0 < a.K; k0 += BK) {
    sstore(As, Ws);
    __syncthreads();
    if (k0 + BK < a.K) gload(k0 + BK);
    compute(As, Ws);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int m = m0 + wr * (BM / 2) + i * 16 + (lane & 15), n = n0 + wc * (BN / 2) + j * 16 + 4 * (lane >> 4);
      if (m < a.M && n < a.N) {
        const float4v v = acc[i][j];
        const uint32_t lo = (uint32_t)f2bf(v[0]).x | ((uint32_t)f2bf(v[1]).x << 16), hi = (uint32_t)f2bf(v[2]).x | ((uint32_t)f2bf(v[3]).x << 16);
        *reinterpret_cast<uint2*>(a.C + (int64_t)m * a.N + n) = make_uint2(lo, hi);
      }
    }
}

// ---------------- V1: two LDS buffers, ONE barrier per tile, write tile t+1 after the barrier, re-issue t+2 at once ------
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void v1(GA a) {
  constexpr int BK = 64, VEC = 8, LD = BK + VEC, MI = BM / 32, NI = BN / 32, KV = BK / VEC;
  constexpr int AV = BM * KV / 256, WV = BN * KV / 256, TILE = (BM + BN) * LD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* S = reinterpret_cast<bf16*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN), nwg = tiles_m * tiles_n;
  int id = blockIdx.x;
  { const int q = nwg / 8, r = nwg % 8, xcd = id % 8; id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8; }
  const int m0 = (id % tiles_m) * BM, n0 = (id / tiles_m) * BN;
  float4v acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = float4v{0, 0, 0, 0};
  uint4v ra[AV], rw[WV];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int v = tid + i * 256, row = v / KV, kv = v % KV, gm = m0 + row, gk = k0 + kv * VEC;
      ra[i] = (gm < a.M && gk < a.K) ? *reinterpret_cast<const uint4v*>(a.A + (int64_t)gm * a.K + gk) : uint4v{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < WV; ++i) {
      const int v = tid + i * 256, row = v / KV, kv = v % KV, gn = n0 + row, gk = k0 + kv * VEC;
      rw[i] = (gn < a.N && gk < a.K) ? *reinterpret_cast<const uint4v*>(a.W + (int64_t)gn * a.K + gk) : uint4v{0, 0, 0, 0};
    }
  };
  auto sstore = [&](bf16* Ad, bf16* Wd) {
#pragma unroll
    for (int i = 0; i < AV; ++i) { const int v = tid + i * 256, row = v / KV, kv = v % KV; *reinterpret_cast<uint4v*>(&Ad[row * LD + kv * VEC]) = ra[i]; }
#pragma unroll
    for (int i = 0; i < WV; ++i) { const int v = tid + i * 256, row = v / KV, kv = v % KV; *reinterpret_cast<uint4v*>(&Wd[row * LD + kv * VEC]) = rw[i]; }
  };
  auto compute = [&](const bf16* Ad, const bf16* Wd) {
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      short8 af[MI], bfr[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const short8*>(&Ad[(wr * (BM / 2) + i * 16 + (lane & 15)) * LD + ks * 32 + 8 * (lane >> 4)]);
#pragma unroll
      for (int j = 0; j < NI; ++j) bfr[j] = *reinterpret_cast<const short8*>(&Wd[(wc * (BN / 2) + j * 16 + (lane & 15)) * LD + ks * 32 + 8 * (lane >> 4)]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  const int nt = cdiv(a.K, BK);
  gload(0);
  sstore(S, S + BM * LD);          // tile 0 -> buffer 0
  if (nt > 1) gload(BK);           // tile 1 in registers
  for (int t = 0; t < nt; ++t) {
    bf16* cur = S + (t & 1) * TILE;
    bf16* nxt = S + ((t + 1) & 1) * TILE;
    __syncthreads();               // tile t visible; everyone finished reading `nxt` (tile t-1)
    if (t + 1 < nt) {
      sstore(nxt, nxt + BM * LD);  // tile t+1 (in registers since last iteration)
      if (t + 2 < nt) gload((t + 2) * BK);
    }
    compute(cur, cur + BM * LD);
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int m = m0 + wr * (BM / 2) + i * 16 + (lane & 15), n = n0 + wc * (BN / 2) + j * 16 + 4 * (lane >> 4);
      if (m < a.M && n < a.N) {
        const float4v v = acc[i][j];
        const uint32_t lo = (uint32_t)f2bf(v[0]).x | ((uint32_t)f2bf(v[1]).x << 16), hi = (uint32_t)f2bf(v[2]).x | ((uint32_t)f2bf(v[3]).x << 16);
        *reinterpret_cast<uint2*>(a.C + (int64_t)m * a.N + n) = make_uint2(lo, hi);
      }
    }
}


// ---------------- V2: 256x256x64 tile, 8 waves (2 x 4, 128x64 each), global_load_lds into two swizzled LDS buffers,
//                  one raw barrier per tile ---------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;
template <int BM, int BN, int WM, int WN>   // WM x WN waves
__global__ __launch_bounds__(WM * WN * 64) void v2(GA a) {
  constexpr int BK = 64, NT = WM * WN * 64, NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
  constexpr int ROWB = BK * 2;                      // 128 bytes per tile row
  constexpr int TILEB = (BM + BN) * ROWB;           // one stage
  constexpr int RA = BM / (NW * 8), RW = BN / (NW * 8);   // glds rounds (8 rows per wave-instruction)
  static_assert(BM % (NW * 8) == 0 && BN % (NW * 8) == 0, "tile rows must split over the waves");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN), nwg = tiles_m * tiles_n;
  int id = blockIdx.x;
  { const int q = nwg / 8, r = nwg % 8, xcd = id % 8; id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8; }
  const int m0 = (id % tiles_m) * BM, n0 = (id / tiles_m) * BN;
  float4v acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = float4v{0, 0, 0, 0};

  // staging: lane l of a wave-instruction lands at LDS base + 16*l = row (l>>3), physical 16-byte chunk (l&7);
  // the chunk it must FETCH is the logical one, c = p ^ ((row>>1)&7)  (swizzle on the source, same involution on the read)
  const int srow = lane >> 3, sp = lane & 7;
  auto stage = [&](int buf, int t) {
    char* base = smem + buf * TILEB;
    const int k0 = t * BK;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int row = (r * NW + wave) * 8 + srow;
      const int c = sp ^ ((row >> 1) & 7);
      int gm = m0 + row; gm = gm < a.M ? gm : a.M - 1;
      const bf16* src = a.A + (int64_t)gm * a.K + k0 + c * 8;
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(base + (r * NW + wave) * 8 * ROWB), 16, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int row = (r * NW + wave) * 8 + srow;
      const int c = sp ^ ((row >> 1) & 7);
      int gn = n0 + row; gn = gn < a.N ? gn : a.N - 1;
      const bf16* src = a.W + (int64_t)gn * a.K + k0 + c * 8;
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(base + BM * ROWB + (r * NW + wave) * 8 * ROWB), 16, 0, 0);
    }
  };
  auto compute = [&](int buf) {
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
        const int row = wc * TN + j * 16 + (lane & 15);
        bfr[j] = *reinterpret_cast<const short8*>(Wb + row * ROWB + ((c ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  const int nt = a.K / BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) stage((t + 1) & 1, t + 1);
    compute(t & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int m = m0 + wr * TM + i * 16 + (lane & 15), n = n0 + wc * TN + j * 16 + 4 * (lane >> 4);
      if (m < a.M && n < a.N) {
        const float4v v = acc[i][j];
        const uint32_t lo = (uint32_t)f2bf(v[0]).x | ((uint32_t)f2bf(v[1]).x << 16), hi = (uint32_t)f2bf(v[2]).x | ((uint32_t)f2bf(v[3]).x << 16);
        *reinterpret_cast<uint2*>(a.C + (int64_t)m * a.N + n) = make_uint2(lo, hi);
      }
    }
}

// ---------------- V3: V2 with NS stage buffers: tile t+NS-1 is requested right after the barrier of tile t,
//                  counted vmcnt leaves NS-2 tiles in flight across the barrier ------------------------------------------
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int BM, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__(WM * WN * 64) void v3(GA a) {
  constexpr int BK = 64, NT = WM * WN * 64, NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
  constexpr int ROWB = BK * 2, TILEB = (BM + BN) * ROWB;
  constexpr int RA = BM / (NW * 8), RW = BN / (NW * 8), LPT = RA + RW;
  static_assert((NS - 2) * LPT <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN), nwg = tiles_m * tiles_n;
  int id = blockIdx.x;
  { const int q = nwg / 8, r = nwg % 8, xcd = id % 8; id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8; }
  int gm = (int)(sqrtf((float)(nwg > 8 ? nwg / 8 : 1)) + 0.5f); gm = gm < 1 ? 1 : (gm > tiles_m ? tiles_m : gm);
  const int per = gm * tiles_n, g = id / per, first = g * gm, gsz = tiles_m - first < gm ? tiles_m - first : gm;
  const int m0 = (first + (id % per) % gsz) * BM, n0 = ((id % per) / gsz) * BN;
  float4v acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = float4v{0, 0, 0, 0};
  const int srow = lane >> 3, sp = lane & 7;
  const bf16* asrc[RA]; const bf16* wsrc[RW];
#pragma unroll
  for (int r = 0; r < RA; ++r) { const int row = (r * NW + wave) * 8 + srow; int gmr = m0 + row; gmr = gmr < a.M ? gmr : a.M - 1; asrc[r] = a.A + (int64_t)gmr * a.K + ((sp ^ ((row >> 1) & 7)) << 3); }
#pragma unroll
  for (int r = 0; r < RW; ++r) { const int row = (r * NW + wave) * 8 + srow; int gn = n0 + row; gn = gn < a.N ? gn : a.N - 1; wsrc[r] = a.W + (int64_t)gn * a.K + ((sp ^ ((row >> 1) & 7)) << 3); }
  auto stage = [&](auto buf_c, int t) {
    constexpr int buf = decltype(buf_c)::value;
    char* base = smem + buf * TILEB;
#pragma unroll
    for (int r = 0; r < RA; ++r) __builtin_amdgcn_global_load_lds((gas_ptr)(asrc[r] + t * BK), (las_ptr)(base + (r * NW + wave) * 8 * ROWB), 16, 0, 0);
#pragma unroll
    for (int r = 0; r < RW; ++r) __builtin_amdgcn_global_load_lds((gas_ptr)(wsrc[r] + t * BK), (las_ptr)(base + BM * ROWB + (r * NW + wave) * 8 * ROWB), 16, 0, 0);
  };
  auto compute = [&](auto buf_c) {
    constexpr int buf = decltype(buf_c)::value;
    const char* Ab = smem + buf * TILEB;
    const char* Wb = Ab + BM * ROWB;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      short8 af[MI], bfr[NI];
      const int c = ks * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < MI; ++i) { const int row = wr * TM + i * 16 + (lane & 15); af[i] = *reinterpret_cast<const short8*>(Ab + row * ROWB + ((c ^ ((row >> 1) & 7)) << 4)); }
#pragma unroll
      for (int j = 0; j < NI; ++j) { const int row = wc * TN + j * 16 + (lane & 15); bfr[j] = *reinterpret_cast<const short8*>(Wb + row * ROWB + ((c ^ ((row >> 1) & 7)) << 4)); }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  const int nt = a.K / BK;
  // prologue: tiles 0 .. NS-2 (nt >= NS-1 assumed by the host)
  static_for(std::make_integer_sequence<int, NS - 1>{}, [&](auto b) { if (decltype(b)::value < nt) stage(b, decltype(b)::value); });
  for (int t0 = 0; t0 < nt; t0 += NS) {
    static_for(std::make_integer_sequence<int, NS>{}, [&](auto b) {
      constexpr int B = decltype(b)::value;
      const int t = t0 + B;
      if (t < nt) {
        // tiles still in flight behind tile t: min(NS-2, nt-1-t)
        const int behind = nt - 1 - t;
        if (behind >= NS - 2) wait_vm<(NS - 2) * LPT>();
        else if (NS > 3 && behind == 1) wait_vm<LPT>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (t + NS - 1 < nt) stage(std::integral_constant<int, (B + NS - 1) % NS>(), t + NS - 1);
        compute(b);
      }
    });
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int m = m0 + wr * TM + i * 16 + (lane & 15), n = n0 + wc * TN + j * 16 + 4 * (lane >> 4);
      if (m < a.M && n < a.N) {
        const float4v v = acc[i][j];
        const uint32_t lo = (uint32_t)f2bf(v[0]).x | ((uint32_t)f2bf(v[1]).x << 16), hi = (uint32_t)f2bf(v[2]).x | ((uint32_t)f2bf(v[3]).x << 16);
        *reinterpret_cast<uint2*>(a.C + (int64_t)m * a.N + n) = make_uint2(lo, hi);
      }
    }
}

// ---------------- V6: W streamed global -> VGPR per wave (no LDS for the weight operand), A through LDS-DMA -------------
// Skinny-M prefill shapes are bound by bytes in flight per CU (LDS capacity x latency): with the weight tiles in a
// register ring only A occupies LDS, so NS can be 5-8 instead of 2-3.  NW waves, each owns TN = BN / NW columns.
template <int BM, int BN, int NW, int NS, bool PACKED = false>
__global__ __launch_bounds__(NW * 64) void v6(GA a) {
  constexpr int BK = 64;
  constexpr int TN = BN / NW, MI = BM / 16, NI = TN / 16;
  constexpr int ROWB = BK * 2, TILEB = BM * ROWB;
  constexpr int RA = BM / (NW * 8), LPT = RA + 2 * NI;
  static_assert(BM % (NW * 8) == 0 && (NS - 2) * LPT <= 63, "tile / vmcnt range");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN), nwg = tiles_m * tiles_n;
  int id = blockIdx.x;
  { const int q = nwg / 8, r = nwg % 8, xcd = id % 8; id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8; }
  const int m0 = (id % tiles_m) * BM, n0 = (id / tiles_m) * BN;   // M-fastest: the M tiles of a W panel are neighbours
  float4v acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = float4v{0, 0, 0, 0};
  const int srow = lane >> 3, sp = lane & 7;
  const bf16* asrc[RA]; const bf16* wsrc[NI];
#pragma unroll
  for (int r = 0; r < RA; ++r) { const int row = (r * NW + wave) * 8 + srow; int gmr = m0 + row; gmr = gmr < a.M ? gmr : a.M - 1; asrc[r] = a.A + (int64_t)gmr * a.K + ((sp ^ ((row >> 1) & 7)) << 3); }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    int gn = n0 + wave * TN + j * 16 + (lane & 15); gn = gn < a.N ? gn : a.N - 1;
    if (PACKED) {  // timing probe: W as [N/16][K/32][64 lanes][8] fragments (results are wrong on row-major W)
      int g16 = (n0 + wave * TN) / 16 + j; g16 = g16 < a.N / 16 ? g16 : a.N / 16 - 1;
      wsrc[j] = a.W + ((int64_t)g16 * (a.K / 32) * 64 + lane) * 8;
    } else wsrc[j] = a.W + (int64_t)gn * a.K + 8 * (lane >> 4);
  }
  short8 wreg[NS][2][NI];
  auto stage = [&](auto buf_c, int t) {
    constexpr int buf = decltype(buf_c)::value;
    char* base = smem + buf * TILEB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < NI; ++j) wreg[buf][ks][j] = *reinterpret_cast<const short8*>(wsrc[j] + (PACKED ? (t * 2 + ks) * 512 : t * BK + ks * 32));
#pragma unroll
    for (int r = 0; r < RA; ++r) __builtin_amdgcn_global_load_lds((gas_ptr)(asrc[r] + t * BK), (las_ptr)(base + (r * NW + wave) * 8 * ROWB), 16, 0, 0);
  };
  auto compute = [&](auto buf_c) {
    constexpr int buf = decltype(buf_c)::value;
    const char* Ab = smem + buf * TILEB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      short8 af[MI];
      const int c = ks * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < MI; ++i) { const int row = i * 16 + (lane & 15); af[i] = *reinterpret_cast<const short8*>(Ab + row * ROWB + ((c ^ ((row >> 1) & 7)) << 4)); }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[buf][ks][j], af[i], acc[i][j], 0, 0, 0);
    }
  };
  const int nt = a.K / BK;
  static_for(std::make_integer_sequence<int, NS - 1>{}, [&](auto b) { stage(b, decltype(b)::value); });  // nt >= NS - 1 (host)
  // steady state: whole groups of NS tiles with a tile to request for every one of them -- branch-free, so that the
  // compiler's own vmcnt for the register ring is the exact count and not a conservative vmcnt(0) at a join
  int t0 = 0;
  for (; t0 + 2 * NS - 1 <= nt; t0 += NS) {
    static_for(std::make_integer_sequence<int, NS>{}, [&](auto b) {
      constexpr int B = decltype(b)::value;
      wait_vm<(NS - 2) * LPT>();
      __builtin_amdgcn_s_barrier();
      stage(std::integral_constant<int, (B + NS - 1) % NS>(), t0 + B + NS - 1);
      compute(b);
    });
  }
  for (; t0 < nt; t0 += NS) {
    static_for(std::make_integer_sequence<int, NS>{}, [&](auto b) {
      constexpr int B = decltype(b)::value;
      const int t = t0 + B;
      if (t < nt) {
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (t + NS - 1 < nt) stage(std::integral_constant<int, (B + NS - 1) % NS>(), t + NS - 1);
        compute(b);
      }
    });
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int m = m0 + i * 16 + (lane & 15), n = n0 + wave * TN + j * 16 + 4 * (lane >> 4);
      if (m < a.M && n < a.N) {
        const float4v v = acc[i][j];
        const uint32_t lo = (uint32_t)f2bf(v[0]).x | ((uint32_t)f2bf(v[1]).x << 16), hi = (uint32_t)f2bf(v[2]).x | ((uint32_t)f2bf(v[3]).x << 16);
        *reinterpret_cast<uint2*>(a.C + (int64_t)m * a.N + n) = make_uint2(lo, hi);
      }
    }
}

template <typename K>
static float run(K kern, int BM, int BN, size_t lds, GA a, int iters, int threads = 256) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  dim3 grid(cdiv(a.M, BM) * cdiv(a.N, BN));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(threads), lds, 0, a);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(threads), lds, 0, a);
  CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters * 1e3f;
}

template <typename K>
static float run_cold(K kern, int BM, int BN, size_t lds, GA a, const std::vector<bf16*>& Ws, int iters, int threads) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  dim3 grid(cdiv(a.M, BM) * cdiv(a.N, BN));
  for (size_t i = 0; i < Ws.size(); ++i) { a.W = Ws[i]; hipLaunchKernelGGL(kern, grid, dim3(threads), lds, 0, a); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) { a.W = Ws[i % Ws.size()]; hipLaunchKernelGGL(kern, grid, dim3(threads), lds, 0, a); }
  CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters * 1e3f;
}

int main() {
  const int shapes[][3] = {{320, 22016, 4096}, {320, 12288, 4096}, {320, 4096, 4096}, {320, 4096, 11008}};
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    std::vector<uint16_t> hA((size_t)M * K), hW((size_t)N * K);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f * 2.f - 1.f; };
    for (auto& v : hA) v = f2bf_host(rnd()).x;
    for (auto& v : hW) v = f2bf_host(rnd() * 0.05f).x;
    bf16 *A, *C0, *C1;
    CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&C0, (size_t)M * N * 2)); CK(hipMalloc(&C1, (size_t)M * N * 2));
    CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    const int nW = (int)(700e6 / ((double)N * K * 2)) + 1;   // rotate weights past the 256 MiB Infinity Cache
    std::vector<bf16*> Ws(nW);
    for (auto& w : Ws) { CK(hipMalloc(&w, hW.size() * 2)); CK(hipMemcpy(w, hW.data(), hW.size() * 2, hipMemcpyHostToDevice)); }
    const int it = 40;
    GA a0{A, Ws[0], C0, M, N, K}, a1{A, Ws[0], C1, M, N, K};
    std::vector<uint16_t> h0((size_t)M * N), h1((size_t)M * N);
    auto check = [&]() {
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h1.data(), C1, h1.size() * 2, hipMemcpyDeviceToHost));
      size_t bad = 0; for (size_t i = 0; i < h0.size(); ++i) bad += h0[i] != h1[i];
      CK(hipMemset(C1, 0, (size_t)M * N * 2));
      return bad;
    };
    const float r2 = run_cold(v3<64, 256, 1, 4, 2>, 64, 256, 2 * 320 * 128, a0, Ws, it, 256);
    const float r3 = run_cold(v3<64, 256, 1, 4, 3>, 64, 256, 3 * 320 * 128, a0, Ws, it, 256);
    run_cold(v3<64, 256, 1, 4, 3>, 64, 256, 3 * 320 * 128, a0, Ws, 1, 256);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h0.data(), C0, h0.size() * 2, hipMemcpyDeviceToHost));
    printf("M=%5d N=%6d K=%5d cold(%d W) | v3 64x256: NS2 %6.1f NS3 %6.1f |", M, N, K, nW, r2, r3);
    { const float t = run_cold(v6<64, 256, 8, 4>, 64, 256, 4 * 64 * 128, a1, Ws, it, 512); printf(" v6 64x256/8w NS4 %6.1f (bad %zu)", t, check()); }
    { const float t = run_cold(v6<64, 256, 8, 6>, 64, 256, 6 * 64 * 128, a1, Ws, it, 512); printf(" NS6 %6.1f (bad %zu)", t, check()); }
    { const float t = run_cold(v6<64, 256, 8, 8>, 64, 256, 8 * 64 * 128, a1, Ws, it, 512); printf(" NS8 %6.1f (bad %zu)", t, check()); }
    { const float t = run_cold(v6<64, 256, 8, 4, true>, 64, 256, 4 * 64 * 128, a1, Ws, it, 512); printf(" | PACKED NS4 %6.1f", t); check(); }
    { const float t = run_cold(v6<64, 256, 8, 8, true>, 64, 256, 8 * 64 * 128, a1, Ws, it, 512); printf(" NS8 %6.1f", t); check(); }
    { const float t = run_cold(v6<320, 128, 8, 3, true>, 320, 128, 3 * 320 * 128, a1, Ws, it, 512); printf(" 320x128 NS3 %6.1f", t); check(); }
    { const float t = run_cold(v6<160, 128, 4, 4, true>, 160, 128, 4 * 160 * 128, a1, Ws, it, 256); printf(" 160x128 NS4 %6.1f", t); check(); }
    { const float t = run_cold(v6<64, 128, 4, 6>, 64, 128, 6 * 64 * 128, a1, Ws, it, 256); printf(" | 64x128/4w NS6 %6.1f (bad %zu)", t, check()); }
    { const float t = run_cold(v6<160, 128, 4, 4>, 160, 128, 4 * 160 * 128, a1, Ws, it, 256); printf(" | 160x128/4w NS4 %6.1f (bad %zu)", t, check()); }
    { const float t = run_cold(v6<320, 128, 8, 3>, 320, 128, 3 * 320 * 128, a1, Ws, it, 512); printf(" | 320x128/8w NS3 %6.1f (bad %zu)", t, check()); }
    printf("\n");
    fflush(stdout);
    hipFree(A); hipFree(C0); hipFree(C1); for (auto w : Ws) hipFree(w);
  }
  return 0;
}

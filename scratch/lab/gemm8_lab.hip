// Lab: phase-split variants of the 256x256 LDS-DMA GEMM against the shipped structure (v3 of gemm_lab.hip) on the
// SAM-H shapes.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I anyref_amd/csrc -o scratch/bin/gemm8_lab scratch/lab/gemm8_lab.hip
#define main gemm_lab_main_unused
#include "gemm_lab.hip"
#undef main

// V4: v3<256,256,2,4,2>'s data movement with the K-tile's work cut into 4 quadrant phases per wave
// (rows i in {0,1} x cols j in {0,1} of its 128x64 output, 16 MFMAs each, order 00 01 11 10 so one operand's
// fragments carry over), the next tile's 8 LDS-DMA instructions SPREAD over the phases (2 per phase) instead of a
// burst behind the barrier, fragment reads of phase p+1 issued before the MFMAs of phase p, optional s_setprio
// around each MFMA cluster.
template <int SPREAD, int PRIO, int NS>
__global__ __launch_bounds__(512) void v4(GA a) {
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4, BK = 64, NW = 8;
  constexpr int TM = 128, TN = 64;
  constexpr int ROWB = BK * 2, TILEB = (BM + BN) * ROWB;
  constexpr int RA = BM / (NW * 8), RW = BN / (NW * 8), LPT = RA + RW;   // 4 + 4
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN), nwg = tiles_m * tiles_n;
  int id = blockIdx.x;
  { const int q = nwg / 8, r = nwg % 8, xcd = id % 8; id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8; }
  int gm = (int)(sqrtf((float)(nwg > 8 ? nwg / 8 : 1)) + 0.5f); gm = gm < 1 ? 1 : (gm > tiles_m ? tiles_m : gm);
  const int per = gm * tiles_n, g = id / per, first = g * gm, gsz = tiles_m - first < gm ? tiles_m - first : gm;
  const int m0 = (first + (id % per) % gsz) * BM, n0 = ((id % per) / gsz) * BN;
  float4v acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0, 0, 0, 0};
  const int srow = lane >> 3, sp = lane & 7;
  const bf16* asrc[RA]; const bf16* wsrc[RW];
#pragma unroll
  for (int r = 0; r < RA; ++r) { const int row = (r * NW + wave) * 8 + srow; int gmr = m0 + row; gmr = gmr < a.M ? gmr : a.M - 1; asrc[r] = a.A + (int64_t)gmr * a.K + ((sp ^ ((row >> 1) & 7)) << 3); }
#pragma unroll
  for (int r = 0; r < RW; ++r) { const int row = (r * NW + wave) * 8 + srow; int gn = n0 + row; gn = gn < a.N ? gn : a.N - 1; wsrc[r] = a.W + (int64_t)gn * a.K + ((sp ^ ((row >> 1) & 7)) << 3); }
  // DMA instruction q (0..7) of tile t into stage buffer buf: q < 4 -> A round q, else W round q - 4
  auto dma = [&](auto buf_c, auto q_c, int t) {
    constexpr int buf = decltype(buf_c)::value, q = decltype(q_c)::value;
    char* base = smem + buf * TILEB;
    if constexpr (q < RA) __builtin_amdgcn_global_load_lds((gas_ptr)(asrc[q] + t * BK), (las_ptr)(base + (q * NW + wave) * 8 * ROWB), 16, 0, 0);
    else __builtin_amdgcn_global_load_lds((gas_ptr)(wsrc[q - RA] + t * BK), (las_ptr)(base + BM * ROWB + ((q - RA) * NW + wave) * 8 * ROWB), 16, 0, 0);
  };
  auto stage_all = [&](auto buf_c, int t) {
    static_for(std::make_integer_sequence<int, LPT>{}, [&](auto q) { dma(buf_c, q, t); });
  };
  // fragment loads: A rows i*64 .. +64 of the wave's 128 (4 frags x 2 ks), W cols j*32 .. +32 (2 frags x 2 ks)
  auto lda = [&](const char* Ab, int i, short8 (&af)[2][4]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int row = wr * TM + i * 64 + f * 16 + (lane & 15), c = ks * 4 + (lane >> 4);
        af[ks][f] = *reinterpret_cast<const short8*>(Ab + row * ROWB + ((c ^ ((row >> 1) & 7)) << 4));
      }
  };
  auto ldw = [&](const char* Wb, int j, short8 (&bf)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int row = wc * TN + j * 32 + f * 16 + (lane & 15), c = ks * 4 + (lane >> 4);
        bf[ks][f] = *reinterpret_cast<const short8*>(Wb + row * ROWB + ((c ^ ((row >> 1) & 7)) << 4));
      }
  };
  auto mma = [&](auto i_c, auto j_c, const short8 (&af)[2][4], const short8 (&bf)[2][2]) {
    constexpr int i = decltype(i_c)::value, j = decltype(j_c)::value;
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          acc[i * 4 + f][j * 2 + h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ks][h], af[ks][f], acc[i * 4 + f][j * 2 + h], 0, 0, 0);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  const int nt = a.K / BK;
  static_for(std::make_integer_sequence<int, NS - 1>{}, [&](auto b) { if (decltype(b)::value < nt) stage_all(b, decltype(b)::value); });
  for (int t0 = 0; t0 < nt; t0 += NS) {
    static_for(std::make_integer_sequence<int, NS>{}, [&](auto b) {
      constexpr int B = decltype(b)::value;
      const int t = t0 + B;
      if (t < nt) {
        const int behind = nt - 1 - t;
        if (behind >= NS - 2) wait_vm<(NS - 2) * LPT>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        const bool more = t + NS - 1 < nt;
        using NB_ = std::integral_constant<int, (B + NS - 1) % NS>;
        const char* Ab = smem + B * TILEB;
        const char* Wb = Ab + BM * ROWB;
        short8 a0[2][4], a1[2][4], b0[2][2], b1[2][2];
        if (!SPREAD) { if (more) stage_all(NB_(), t + NS - 1); }
        ldw(Wb, 0, b0); lda(Ab, 0, a0);
        if (SPREAD && more) { dma(NB_(), std::integral_constant<int, 0>(), t + NS - 1); dma(NB_(), std::integral_constant<int, 1>(), t + NS - 1); }
        ldw(Wb, 1, b1);                       // next phase's operand, in flight under this phase's MFMAs
        mma(I0(), I0(), a0, b0);
        if (SPREAD && more) { dma(NB_(), std::integral_constant<int, 2>(), t + NS - 1); dma(NB_(), std::integral_constant<int, 3>(), t + NS - 1); }
        lda(Ab, 1, a1);
        mma(I0(), I1(), a0, b1);
        if (SPREAD && more) { dma(NB_(), std::integral_constant<int, 4>(), t + NS - 1); dma(NB_(), std::integral_constant<int, 5>(), t + NS - 1); }
        mma(I1(), I1(), a1, b1);
        if (SPREAD && more) { dma(NB_(), std::integral_constant<int, 6>(), t + NS - 1); dma(NB_(), std::integral_constant<int, 7>(), t + NS - 1); }
        mma(I1(), I0(), a1, b0);
      }
    });
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wr * TM + i * 16 + (lane & 15), n = n0 + wc * TN + j * 16 + 4 * (lane >> 4);
      if (m < a.M && n < a.N) {
        const float4v v = acc[i][j];
        const uint32_t lo = (uint32_t)f2bf(v[0]).x | ((uint32_t)f2bf(v[1]).x << 16), hi = (uint32_t)f2bf(v[2]).x | ((uint32_t)f2bf(v[3]).x << 16);
        *reinterpret_cast<uint2*>(a.C + (int64_t)m * a.N + n) = make_uint2(lo, hi);
      }
    }
}

// V5: PERSISTENT tile loop.  One workgroup per CU walks its share of BM x BN tiles; the LDS-DMA ring (NS stages)
// runs over the flat sequence of (tile, k-tile) steps, so the next tile's first K tiles are already landing while
// this tile's epilogue converts and stores (stores are fire-and-forget): prologue and epilogue of every tile but
// the first / last hide behind the neighbouring tile's main loop.  For K = 1280 (20 K tiles) those are 25 % of v3.
template <int BM, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__(WM * WN * 64) void v5(GA a) {
  constexpr int BK = 64, NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
  constexpr int ROWB = BK * 2, TILEB = (BM + BN) * ROWB;
  constexpr int RA = BM / (NW * 8), RW = BN / (NW * 8), LPT = RA + RW;
  static_assert((NS - 2) * LPT + MI * NI <= 63 && BM % (NW * 8) == 0 && BN % (NW * 8) == 0, "shape / vmcnt range");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN), ntile = tiles_m * tiles_n;
  const int nt = a.K / BK;
  const int srow = lane >> 3, sp = lane & 7;
  // tile id -> (m0, n0): consecutive ids walk down a column of tiles (same W panel), workgroups of one XCD
  // (blockIdx % 8) take a contiguous chunk of ids
  const int G = gridDim.x, per_x = G / 8, xcd = blockIdx.x % 8, wi = blockIdx.x / 8;
  auto tile_of = [&](int i) {  // i-th tile of this workgroup, or -1
    const int chunk = (ntile + 7) / 8;
    const int id = xcd * chunk + i * per_x + wi;
    return (i * per_x + wi < chunk && id < ntile) ? id : -1;
  };
  int my = 0;
  while (tile_of(my) >= 0) ++my;
  if (my == 0) return;
  const int total = my * nt;
  float4v acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = float4v{0, 0, 0, 0};
  auto stage = [&](auto buf_c, int step) {
    constexpr int buf = decltype(buf_c)::value;
    const int id = tile_of(step / nt), t = step % nt;
    const int m0 = (id % tiles_m) * BM, n0 = (id / tiles_m) * BN;
    char* base = smem + buf * TILEB;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int row = (r * NW + wave) * 8 + srow;
      int gm = m0 + row; gm = gm < a.M ? gm : a.M - 1;
      __builtin_amdgcn_global_load_lds((gas_ptr)(a.A + (int64_t)gm * a.K + ((sp ^ ((row >> 1) & 7)) << 3) + t * BK),
                                       (las_ptr)(base + (r * NW + wave) * 8 * ROWB), 16, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int row = (r * NW + wave) * 8 + srow;
      int gn = n0 + row; gn = gn < a.N ? gn : a.N - 1;
      __builtin_amdgcn_global_load_lds((gas_ptr)(a.W + (int64_t)gn * a.K + ((sp ^ ((row >> 1) & 7)) << 3) + t * BK),
                                       (las_ptr)(base + BM * ROWB + (r * NW + wave) * 8 * ROWB), 16, 0, 0);
    }
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
  auto epilogue = [&](int id) {
    const int m0 = (id % tiles_m) * BM, n0 = (id / tiles_m) * BN;
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
        acc[i][j] = float4v{0, 0, 0, 0};
      }
  };
  static_for(std::make_integer_sequence<int, NS - 1>{}, [&](auto b) { if (decltype(b)::value < total) stage(b, decltype(b)::value); });
  for (int s0 = 0; s0 < total; s0 += NS) {
    static_for(std::make_integer_sequence<int, NS>{}, [&](auto b) {
      constexpr int B = decltype(b)::value;
      const int st = s0 + B;
      if (st < total) {
        // stores of an epilogue sit in the same vmcnt queue as the DMAs: count DMAs only by waiting for everything older
        // than the NS-2 youngest TILES' worth -- an epilogue's stores are older than those, so they are waited for too
        // (lab shapes are multiples of the tile: every lane stores, so the MI * NI epilogue stores of the previous
        //  step are an exact count and may stay in flight together with the NS-2 youngest tiles' DMAs)
        const int behind = total - 1 - st;
        if (behind >= NS - 2) {
          if (st % nt == 0 && st > 0) wait_vm<(NS - 2) * LPT + MI * NI>(); else wait_vm<(NS - 2) * LPT>();
        } else {
          wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();
        if (st + NS - 1 < total) stage(std::integral_constant<int, (B + NS - 1) % NS>(), st + NS - 1);
        compute(b);
        if (st % nt == nt - 1) epilogue(tile_of(st / nt));
      }
    });
  }
}

template <typename K>
static float run_grid(K kern, int grid, size_t lds, GA a, int iters, int threads) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, a);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, a);
  CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters * 1e3f;
}

int main() {
  const int shapes[][3] = {{4096, 3840, 1280}, {4096, 5120, 1280}, {4096, 1280, 5120}, {4096, 4096, 4096}, {8192, 8192, 8192}};
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    std::vector<uint16_t> hA((size_t)M * K), hW((size_t)N * K);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f * 2.f - 1.f; };
    for (auto& v : hA) v = f2bf_host(rnd()).x;
    for (auto& v : hW) v = f2bf_host(rnd() * 0.05f).x;
    bf16 *A, *W, *C0, *C1;
    CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&W, hW.size() * 2)); CK(hipMalloc(&C0, (size_t)M * N * 2)); CK(hipMalloc(&C1, (size_t)M * N * 2));
    CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    GA a0{A, W, C0, M, N, K}, a1{A, W, C1, M, N, K};
    const int it = 30;
    const size_t L2 = 2 * 512 * 128;
    float t[8];
    // interleaved rounds in one process (guide rule 24): 3 rounds, keep the minimum
    for (int k = 0; k < 8; ++k) t[k] = 1e9f;
    for (int round = 0; round < 3; ++round) {
      t[0] = fminf(t[0], run(v3<256, 256, 2, 4, 2>, 256, 256, L2, a0, it, 512));
      t[1] = fminf(t[1], run(v4<0, 0, 2>, 256, 256, L2, a1, it, 512));
      t[2] = fminf(t[2], run(v4<1, 0, 2>, 256, 256, L2, a1, it, 512));
      t[3] = fminf(t[3], run(v4<0, 1, 2>, 256, 256, L2, a1, it, 512));
      t[4] = fminf(t[4], run(v4<1, 1, 2>, 256, 256, L2, a1, it, 512));
      t[5] = fminf(t[5], run_grid(v5<128, 256, 2, 4, 3>, 256, 3 * 384 * 128, a1, it, 512));
      t[6] = fminf(t[6], run_grid(v5<256, 256, 2, 4, 2>, 256, 2 * 512 * 128, a1, it, 512));
      t[7] = fminf(t[7], run_grid(v5<128, 128, 2, 4, 4>, 256, 4 * 256 * 128, a1, it, 512));
    }
    std::vector<uint16_t> h0((size_t)M * N), h1((size_t)M * N);
    run(v3<256, 256, 2, 4, 2>, 256, 256, L2, a0, 1, 512);
    run_grid(v5<128, 256, 2, 4, 3>, 256, 3 * 384 * 128, a1, 1, 512);
    CK(hipMemcpy(h0.data(), C0, h0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), C1, h1.size() * 2, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < h0.size(); ++i) bad += h0[i] != h1[i];
    const double fl = 2.0 * M * N * K;
    printf("M=%5d N=%5d K=%5d | v3 %7.1f us %6.0f TF | v4 plain %7.1f  spread %7.1f  prio %7.1f  spread+prio %7.1f | v5 persistent 128x256/3 %7.1f  256x256/2 %7.1f  128x128/4 %7.1f | mismatch(v5) %zu\n",
           M, N, K, t[0], fl / t[0] * 1e-6, t[1], t[2], t[3], t[4], t[5], t[6], t[7], bad);
    fflush(stdout);
    hipFree(A); hipFree(W); hipFree(C0); hipFree(C1);
  }
  return 0;
}

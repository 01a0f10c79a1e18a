// Normalisation, data-movement and elementwise kernels (all HBM-bound; vectorised where the
// layout allows, fp32 math throughout).
#include <algorithm>
#include <cstring>
#include <type_traits>

#include "kernels.h"

namespace anyref {

thread_local Profiler* g_prof = nullptr;

Profiler::~Profiler() {
  for (auto& p : pool_) {
    (void)hipEventDestroy(p.first);
    (void)hipEventDestroy(p.second);
  }
}
bool Profiler::begin(const char* tag, double flops, double bytes, hipStream_t s) {
  if (!filter.empty() && strncmp(tag, filter.c_str(), filter.size()) != 0) return false;  // prefix match
  int t = -1;
  for (size_t i = 0; i < tags_.size(); ++i)
    if (tags_[i] == tag) t = (int)i;
  if (t < 0) {
    tags_.push_back(tag);
    stats_.push_back(ProfStat());
    seen_.push_back(0);
    t = (int)tags_.size() - 1;
  }
  if (sample_every > 1 && (seen_[t]++ % sample_every) != 0) return false;
  if (used_ == pool_.size()) {
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    pool_.push_back({a, b});
  }
  Rec r{t, pool_[used_].first, pool_[used_].second, flops, bytes};
  ++used_;
  HIP_TRY(hipEventRecord(r.a, s));
  recs_.push_back(r);
  return true;
}
void Profiler::end(hipStream_t s) { HIP_TRY(hipEventRecord(recs_.back().b, s)); }
// calibration kernel: every wave spins on the 100 MHz wall clock for ~10 us (a duration of the order of
// the kernels being timed; a near-empty kernel over-states the bracket cost of a real one by 2-3 us)
__global__ void calib_spin_kernel(int ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
}
void Profiler::calibrate(hipStream_t s) {
  // What a bracket adds to the kernel inside it, in rocprofv3's accounting (back-to-back kernels of a
  // stream abut): time [a K b] and [a K K b]; the second K costs E2 - E1, so the bracket costs 2 E1 - E2.
  constexpr int N = 48;
  std::vector<hipEvent_t> ev(4 * N);
  for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
  auto K = [&]() { hipLaunchKernelGGL(calib_spin_kernel, dim3(512), dim3(256), 0, s, 1000); };
  for (int i = 0; i < N; ++i) {
    HIP_TRY(hipEventRecord(ev[4 * i], s));
    K();
    HIP_TRY(hipEventRecord(ev[4 * i + 1], s));
    HIP_TRY(hipEventRecord(ev[4 * i + 2], s));
    K();
    K();
    HIP_TRY(hipEventRecord(ev[4 * i + 3], s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  std::vector<float> e1, e2;
  for (int i = 0; i < N; ++i) {
    float a = 0.f, b = 0.f;
    if (hipEventElapsedTime(&a, ev[4 * i], ev[4 * i + 1]) == hipSuccess &&
        hipEventElapsedTime(&b, ev[4 * i + 2], ev[4 * i + 3]) == hipSuccess) {
      e1.push_back(a);
      e2.push_back(b);
    }
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  null_ms = 0.0;
  if (!e1.empty()) {
    std::sort(e1.begin(), e1.end());
    std::sort(e2.begin(), e2.end());
    const double m1 = e1[e1.size() / 2], m2 = e2[e2.size() / 2];
    null_ms = std::max(0.0, 2.0 * m1 - m2);
  }
}
void Profiler::collect() {
  for (auto& r : recs_) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    ms = ms > (float)null_ms ? ms - (float)null_ms : 0.f;
    ProfStat& st = stats_[r.tag];
    st.ms += ms;
    st.count += 1;
    st.flops += r.flops;
    st.bytes += r.bytes;
  }
  recs_.clear();
  used_ = 0;
}
void Profiler::reset() {
  recs_.clear();
  used_ = 0;
  for (auto& s : stats_) s = ProfStat();
  for (auto& n : seen_) n = 0;
}
std::vector<std::pair<std::string, ProfStat>> Profiler::stats() const {
  std::vector<std::pair<std::string, ProfStat>> v;
  for (size_t i = 0; i < tags_.size(); ++i) v.push_back({tags_[i], stats_[i]});
  return v;
}

// ---- kernel-side timestamps (kernels.h: Stamper) ----------------------------------------------------
thread_local Stamper* g_stamp = nullptr;
namespace {
constexpr size_t kStampEager = (size_t)4096 * 512 * 2;   // u64: 4096 eager launches of 512 workgroups
constexpr size_t kStampStride = (size_t)256 * 512 * 2;   // u64 per replay: 256 launches of 512 workgroups
constexpr int kStampEpochs = 48;
__global__ void stamp_bump_kernel(int* epoch) { *epoch += 1; }
}  // namespace
Stamper::~Stamper() {
  if (buf_) (void)hipFree(buf_);
  if (ctl_) (void)hipFree(ctl_);
}
void Stamper::enable(bool want) {
  if (want && !buf_) {
    eager_cap_ = kStampEager;
    graph_off_ = eager_cap_;
    graph_stride_ = kStampStride;
    max_epoch_ = kStampEpochs;
    const size_t n = eager_cap_ + graph_stride_ * max_epoch_;
    HIP_TRY(hipMalloc((void**)&buf_, n * 8));
    HIP_TRY(hipMemset(buf_, 0, n * 8));
    HIP_TRY(hipMalloc((void**)&ctl_, 2 * sizeof(int)));
    HIP_TRY(hipMemset(ctl_, 0, 2 * sizeof(int)));
  }
  if (want) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(ctl_, 0, 2 * sizeof(int)));
    eager_.clear();
    epoch_keys_.clear();
    eager_used_ = 0;
    dropped_ = 0;
  }
  on = want;
}
StampArgs Stamper::slot(const char* tag, double bytes, int grid) {
  StampArgs a;
  if (!on || !buf_) return a;
  const size_t need = (size_t)grid * 2;
  if (capturing_ >= 0) {
    if (cap_used_ + need > graph_stride_) {  // step larger than the record: this launch goes unstamped (and is counted)
      ++dropped_;
      return a;
    }
    graphs_[capturing_].push_back({tag, bytes, grid, cap_used_});
    a.base = buf_ + graph_off_ + cap_used_;
    a.epoch = ctl_;
    a.stride = (unsigned)graph_stride_;
    a.max_epoch = max_epoch_;
    cap_used_ += need;
  } else {
    if (eager_used_ + need > eager_cap_) {
      ++dropped_;
      return a;
    }
    eager_.push_back({tag, bytes, grid, eager_used_});
    a.base = buf_ + eager_used_;
    a.epoch = ctl_ + 1;
    a.stride = 0;
    a.max_epoch = 1;
    eager_used_ += need;
  }
  return a;
}
void Stamper::graph_begin(int key) {
  capturing_ = key;
  cap_used_ = 0;
  graphs_[key].clear();
}
void Stamper::graph_end(hipStream_t cap) {
  hipLaunchKernelGGL(stamp_bump_kernel, dim3(1), dim3(1), 0, cap, ctl_);
  capturing_ = -1;
}
void Stamper::graph_abort() {
  if (capturing_ >= 0) graphs_.erase(capturing_);
  capturing_ = -1;
  cap_used_ = 0;
}
std::vector<StampRow> Stamper::collect() {
  std::vector<StampRow> rows;
  if (!buf_) return rows;
  const int E = std::min((int)epoch_keys_.size(), max_epoch_);
  std::vector<unsigned long long> h(graph_off_ + graph_stride_ * (size_t)E);
  if (eager_used_) HIP_TRY(hipMemcpy(h.data(), buf_, eager_used_ * 8, hipMemcpyDeviceToHost));
  if (E) HIP_TRY(hipMemcpy(h.data() + graph_off_, buf_ + graph_off_, graph_stride_ * (size_t)E * 8, hipMemcpyDeviceToHost));
  struct Raw { const Rec* r; unsigned long long t0, t1; int epoch; unsigned long long s1, e0; double med; };
  std::vector<Raw> raw;
  auto reduce = [&](const Rec& r, size_t base, int epoch) {
    unsigned long long t0 = ~0ull, t1 = 0, s1 = 0, e0 = ~0ull;
    std::vector<unsigned long long> spans;
    for (int g = 0; g < r.grid; ++g) {
      const unsigned long long a = h[base + (size_t)g * 2], b = h[base + (size_t)g * 2 + 1];
      if (a == 0 || b == 0) continue;  // a workgroup slot that was never written
      t0 = std::min(t0, a);
      t1 = std::max(t1, b);
      s1 = std::max(s1, a);
      e0 = std::min(e0, b);
      spans.push_back(b - a);
    }
    if (t1 > 0 && t0 != ~0ull) {
      std::sort(spans.begin(), spans.end());
      raw.push_back({&r, t0, t1, epoch, s1, e0, (double)spans[spans.size() / 2] * 0.01});
    }
  };
  for (const Rec& r : eager_) reduce(r, r.off, -1);
  for (int e = 0; e < E; ++e)
    for (const Rec& r : graphs_[epoch_keys_[e]]) reduce(r, graph_off_ + graph_stride_ * (size_t)e + r.off, e);
  std::sort(raw.begin(), raw.end(), [](const Raw& a, const Raw& b) { return a.t0 < b.t0; });
  const unsigned long long origin = raw.empty() ? 0 : raw.front().t0;
  for (const Raw& x : raw) {
    StampRow row;
    row.tag = x.r->tag;
    row.bytes = x.r->bytes;
    row.t0_us = (double)(x.t0 - origin) * 0.01;  // 100 MHz wall clock
    row.t1_us = (double)(x.t1 - origin) * 0.01;
    row.epoch = x.epoch;
    row.start_spread_us = (double)(x.s1 - x.t0) * 0.01;
    row.end_spread_us = (double)(x.t1 - x.e0) * 0.01;
    row.wg_median_us = x.med;
    rows.push_back(row);
  }
  // ready for the next pass: slots back to "never written", counters to zero
  HIP_TRY(hipMemset(buf_, 0, (graph_off_ + graph_stride_ * (size_t)max_epoch_) * 8));
  HIP_TRY(hipMemset(ctl_, 0, 2 * sizeof(int)));
  eager_.clear();
  epoch_keys_.clear();
  eager_used_ = 0;
  dropped_ = 0;
  return rows;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm / RMSNorm: one wave per row, two passes over registers-cached data.
// ---------------------------------------------------------------------------------------------
template <typename T, int MAXV>  // MAXV = ceil(D / 64) values per lane kept in registers
__global__ __launch_bounds__(256) void norm_kernel(NormArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * 4 + wave;
  if (m >= a.M) return;
  const int dm = a.row_map ? a.row_map[m] : m;
  if (dm < 0) return;
  const float* x = a.x + (int64_t)m * a.ldx;
  float v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int d = lane + i * 64;
    v[i] = d < a.D ? x[d] : 0.f;
    s += v[i];
  }
  float mean = 0.f;
  if (!a.rms) mean = wave_sum(s) / (float)a.D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int d = lane + i * 64;
    const float c = d < a.D ? v[i] - mean : 0.f;
    ss += c * c;
  }
  const float var = wave_sum(ss) / (float)a.D;
  const float inv = a.rms ? rsqrtf(var + a.eps) : 1.f / sqrtf(var + a.eps);
  float* yf = reinterpret_cast<float*>(a.y) + (int64_t)dm * a.ldy;
  T* yt = reinterpret_cast<T*>(a.y) + (int64_t)dm * a.ldy;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int d = lane + i * 64;
    if (d >= a.D) continue;
    float o = (v[i] - mean) * inv * a.gain[d];
    if (a.bias) o += a.bias[d];
    o = apply_act(o, a.act);
    if (a.y_f32)
      yf[d] = o;
    else
      st1<T>(yt, d, o);
  }
}

// Wide rows (D >= 1024, 16-byte aligned): one 256-thread workgroup per row, float4 loads, the row
// stays in registers between the statistics and the normalise pass.
template <typename T, int NV>  // NV float4 per thread: D <= 256*4*NV
__global__ __launch_bounds__(256) void norm_wide_kernel(NormArgs a) {
  if ((int)blockIdx.x >= a.M) {  // the side job: one bias row per extra workgroup (fill_N % 4 == 0, 8-byte aligned rows)
    // (split-pair mode: the q/k/v rows the fill writes feed the attention kernels, which read f32)
    using FT = std::conditional_t<is_split<T>::value, float, T>;
    FT* d = reinterpret_cast<FT*>(a.fill_dst) + (int64_t)a.fill_rows[blockIdx.x - a.M] * a.fill_ld;
    for (int n = threadIdx.x * 4; n < a.fill_N; n += 256 * 4) {
      const float4v b = *reinterpret_cast<const float4v*>(a.fill_bias + n);
      store4_from_f32<FT>(d + n, b[0], b[1], b[2], b[3]);
    }
    return;
  }
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dm = a.row_map ? a.row_map[m] : m;
  if (dm < 0) return;
  __shared__ float red[8];
  const float4v* x4 = reinterpret_cast<const float4v*>(a.x + (int64_t)m * a.ldx);
  const int n4 = a.D / 4;
  float4v v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = tid + i * 256;
    v[i] = j < n4 ? x4[j] : float4v{0.f, 0.f, 0.f, 0.f};
    s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
  float mean = 0.f;
  if (!a.rms) {
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    mean = (red[0] + red[1] + red[2] + red[3]) / (float)a.D;
  }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = tid + i * 256;
    if (j < n4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float c = v[i][e] - mean;
        ss += c * c;
      }
    }
  }
  ss = wave_sum(ss);
  if (lane == 0) red[4 + wave] = ss;
  __syncthreads();
  const float var = (red[4] + red[5] + red[6] + red[7]) / (float)a.D;
  const float inv = a.rms ? rsqrtf(var + a.eps) : 1.f / sqrtf(var + a.eps);
  const float4v* g4 = reinterpret_cast<const float4v*>(a.gain);
  const float4v* b4 = reinterpret_cast<const float4v*>(a.bias);
  float* yf = reinterpret_cast<float*>(a.y) + (int64_t)dm * a.ldy;
  T* yt = reinterpret_cast<T*>(a.y) + (int64_t)dm * a.ldy;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = tid + i * 256;
    if (j >= n4) continue;
    const float4v g = g4[j];
    float4v o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * inv * g[e];
    if (a.bias) {
      const float4v bb = b4[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] += bb[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = apply_act(o[e], a.act);
    if (a.y_f32) {
      *reinterpret_cast<float4v*>(yf + 4 * j) = o;
    } else {
      st4<T>(yt, 4 * j, o[0], o[1], o[2], o[3]);
    }
  }
}

template <typename T>
void launch_norm(const NormArgs& a, hipStream_t s) {
  if (a.M <= 0) return;
  if (a.D >= 1024 && a.D % 4 == 0 && a.ldx % 4 == 0 && a.ldy % 4 == 0 && !((uintptr_t)a.x & 15) &&
      !((uintptr_t)a.y & 15) && !((uintptr_t)a.gain & 15) && !((uintptr_t)a.bias & 15) && a.D <= 8192) {
    int extra = 0;
    if (a.fill_done) *a.fill_done = false;
    if (a.fill_n > 0 && a.fill_dst && a.fill_bias && a.fill_N % 4 == 0 && a.fill_ld % 4 == 0 && !((uintptr_t)a.fill_dst & 15) &&
        !((uintptr_t)a.fill_bias & 15)) {
      extra = a.fill_n;
      if (a.fill_done) *a.fill_done = true;
    }
    dim3 g(a.M + extra), b(256);
    if (a.D <= 2048)
      hipLaunchKernelGGL((norm_wide_kernel<T, 2>), g, b, 0, s, a);
    else if (a.D <= 4096)
      hipLaunchKernelGGL((norm_wide_kernel<T, 4>), g, b, 0, s, a);
    else
      hipLaunchKernelGGL((norm_wide_kernel<T, 8>), g, b, 0, s, a);
    return;
  }
  dim3 grid(cdiv(a.M, 4)), block(256);
  const int nv = cdiv(a.D, 64);
  if (nv <= 1)
    hipLaunchKernelGGL((norm_kernel<T, 1>), grid, block, 0, s, a);
  else if (nv <= 4)
    hipLaunchKernelGGL((norm_kernel<T, 4>), grid, block, 0, s, a);
  else if (nv <= 16)
    hipLaunchKernelGGL((norm_kernel<T, 16>), grid, block, 0, s, a);
  else if (nv <= 32)
    hipLaunchKernelGGL((norm_kernel<T, 32>), grid, block, 0, s, a);
  else if (nv <= 80)
    hipLaunchKernelGGL((norm_kernel<T, 80>), grid, block, 0, s, a);
  else
    throw std::runtime_error("norm: D > 5120 not supported");
}
template void launch_norm<float>(const NormArgs&, hipStream_t);
template void launch_norm<bf16>(const NormArgs&, hipStream_t);
template void launch_norm<f16>(const NormArgs&, hipStream_t);
template void launch_norm<sp16>(const NormArgs&, hipStream_t);

// ---------------------------------------------------------------------------------------------
// im2col
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void im2col_patch_kernel(const float* __restrict__ img, int B, int S, int p, T* __restrict__ out,
                                    int Kp) {
  const int g = S / p;
  const int64_t total = (int64_t)B * g * g * Kp;
  const int K = 3 * p * p;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kp);
    const int64_t row = i / Kp;
    float v = 0.f;
    if (k < K) {
      const int px = (int)(row % g), py = (int)((row / g) % g), b = (int)(row / ((int64_t)g * g));
      const int kx = k % p, ky = (k / p) % p, c = k / (p * p);
      v = img[(((int64_t)b * 3 + c) * S + (py * p + ky)) * S + px * p + kx];
    }
    st1<T>(out + row * Kp, k, v);
  }
}
template <typename T>
void launch_im2col_patch(const float* img, int B, int S, int p, void* out, int Kp, hipStream_t s) {
  const int g = S / p;
  const int64_t total = (int64_t)B * g * g * Kp;
  const int grid = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
  hipLaunchKernelGGL((im2col_patch_kernel<T>), dim3(grid), dim3(256), 0, s, img, B, S, p,
                     reinterpret_cast<T*>(out), Kp);
}
template void launch_im2col_patch<float>(const float*, int, int, int, void*, int, hipStream_t);
template void launch_im2col_patch<bf16>(const float*, int, int, int, void*, int, hipStream_t);
template void launch_im2col_patch<f16>(const float*, int, int, int, void*, int, hipStream_t);
template void launch_im2col_patch<sp16>(const float*, int, int, int, void*, int, hipStream_t);

template <typename T>
__global__ void im2col_3x3_kernel(const T* __restrict__ in, int B, int g, int C, T* __restrict__ out) {
  // one thread per (row, tap, c)
  const int64_t total = (int64_t)B * g * g * 9 * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int tap = (int)((i / C) % 9);
    const int64_t row = i / (9 * (int64_t)C);
    const int x = (int)(row % g), y = (int)((row / g) % g), b = (int)(row / ((int64_t)g * g));
    const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
    const bool in_img = yy >= 0 && yy < g && xx >= 0 && xx < g;
    if constexpr (is_split<T>::value) {  // both terms of the pair travel as they are
      const uint16_t* ip = reinterpret_cast<const uint16_t*>(in + (((int64_t)b * g + yy) * g + xx) * C) + sp_col(c);
      uint16_t* op = reinterpret_cast<uint16_t*>(out + row * 9 * C) + sp_col(tap * C + c);
      op[0] = in_img ? ip[0] : (uint16_t)0;
      op[64] = in_img ? ip[64] : (uint16_t)0;
    } else {
      T v = from_f32<T>(0.f);
      if (in_img) v = in[(((int64_t)b * g + yy) * g + xx) * C + c];
      out[i] = v;
    }
  }
}
template <typename T>
void launch_im2col_3x3(const void* in, int B, int g, int C, void* out, hipStream_t s) {
  const int64_t total = (int64_t)B * g * g * 9 * C;
  const int grid = (int)(cdiv64(total, 256) < 8192 ? cdiv64(total, 256) : 8192);
  hipLaunchKernelGGL((im2col_3x3_kernel<T>), dim3(grid), dim3(256), 0, s, reinterpret_cast<const T*>(in), B,
                     g, C, reinterpret_cast<T*>(out));
}
template void launch_im2col_3x3<float>(const void*, int, int, int, void*, hipStream_t);
template void launch_im2col_3x3<bf16>(const void*, int, int, int, void*, hipStream_t);
template void launch_im2col_3x3<f16>(const void*, int, int, int, void*, hipStream_t);
template void launch_im2col_3x3<sp16>(const void*, int, int, int, void*, hipStream_t);

// ---------------------------------------------------------------------------------------------
// converts / adds
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void convert_kernel(const float* __restrict__ in, int64_t ld_in, T* __restrict__ out,
                               int64_t ld_out, int rows, int cols) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i % cols;
    st1<T>(out + r * ld_out, (int)c, in[r * ld_in + c]);
  }
}
template <typename T>
void launch_convert(const float* in, int64_t ld_in, void* out, int64_t ld_out, int rows, int cols,
                    hipStream_t s) {
  const int64_t total = (int64_t)rows * cols;
  if (total <= 0) return;
  const int grid = (int)(cdiv64(total, 256) < 8192 ? cdiv64(total, 256) : 8192);
  hipLaunchKernelGGL((convert_kernel<T>), dim3(grid), dim3(256), 0, s, in, ld_in, reinterpret_cast<T*>(out),
                     ld_out, rows, cols);
}
template void launch_convert<float>(const float*, int64_t, void*, int64_t, int, int, hipStream_t);
template void launch_convert<bf16>(const float*, int64_t, void*, int64_t, int, int, hipStream_t);
template void launch_convert<f16>(const float*, int64_t, void*, int64_t, int, int, hipStream_t);
template void launch_convert<sp16>(const float*, int64_t, void*, int64_t, int, int, hipStream_t);

// split-pair rows back to f32 (tests, and wherever an f32 view of an sp16 matrix is needed): out[r, c] = hi + lo
__global__ void unsplit_kernel(const sp16* __restrict__ in, int64_t ld_in, float* __restrict__ out, int64_t ld_out,
                               int rows, int cols) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols, c = i % cols;
    out[r * ld_out + c] = sp_load(in + r * ld_in, (int)c);
  }
}
void launch_unsplit(const void* in, int64_t ld_in, float* out, int64_t ld_out, int rows, int cols, hipStream_t s) {
  const int64_t total = (int64_t)rows * cols;
  if (total <= 0) return;
  const int grid = (int)(cdiv64(total, 256) < 8192 ? cdiv64(total, 256) : 8192);
  hipLaunchKernelGGL(unsplit_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const sp16*>(in), ld_in, out, ld_out, rows,
                     cols);
}

template <typename T>
__global__ void add_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, int bmod,
                                T* __restrict__ out, int M, int D) {
  const int64_t total = (int64_t)M * D;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / D;
    const int d = (int)(i % D);
    out[i] = from_f32<T>(a[i] + b[(m % bmod) * D + d]);
  }
}
void launch_add_rows(const float* a, const float* b, int bmod, float* out, int M, int D, hipStream_t s) {
  const int64_t total = (int64_t)M * D;
  if (total <= 0) return;
  const int grid = (int)(cdiv64(total, 256) < 8192 ? cdiv64(total, 256) : 8192);
  hipLaunchKernelGGL((add_rows_kernel<float>), dim3(grid), dim3(256), 0, s, a, b, bmod, out, M, D);
}
template <typename T>
void launch_add_rows_to(const float* a, const float* b, int bmod, void* out, int M, int D, hipStream_t s) {
  const int64_t total = (int64_t)M * D;
  if (total <= 0) return;
  const int grid = (int)(cdiv64(total, 256) < 8192 ? cdiv64(total, 256) : 8192);
  hipLaunchKernelGGL((add_rows_kernel<T>), dim3(grid), dim3(256), 0, s, a, b, bmod, reinterpret_cast<T*>(out),
                     M, D);
}
template void launch_add_rows_to<float>(const float*, const float*, int, void*, int, int, hipStream_t);
template void launch_add_rows_to<bf16>(const float*, const float*, int, void*, int, int, hipStream_t);

__global__ void add_vec_kernel(const float* __restrict__ a, const float* __restrict__ v, float* __restrict__ out,
                               int M, int D) {
  const int64_t total = (int64_t)M * D;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = a[i] + v[i % D];
}
void launch_add_vec(const float* a, const float* v, float* out, int M, int D, hipStream_t s) {
  const int64_t total = (int64_t)M * D;
  if (total <= 0) return;
  const int grid = (int)(cdiv64(total, 256) < 8192 ? cdiv64(total, 256) : 8192);
  hipLaunchKernelGGL(add_vec_kernel, dim3(grid), dim3(256), 0, s, a, v, out, M, D);
}

// x[b, t, :] = (t == 0 ? cls : patch[b, t-1, :]) + pos[t, :] for t <= n; rows n+1 .. rows-1 of an item are zeroed
// (the audio trunk keeps one spare row per clip for the `add_bias_kv` key / value)
__global__ void clip_assemble_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                     const float* __restrict__ pos, float* __restrict__ x, int B, int n, int D,
                                     int rows) {
  const int64_t total = (int64_t)B * rows * D;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int t = (int)((i / D) % rows);
    const int b = (int)(i / ((int64_t)D * rows));
    if (t > n) {
      x[i] = 0.f;
      continue;
    }
    const float v = t == 0 ? cls[d] : patch[((int64_t)b * n + (t - 1)) * D + d];
    x[i] = v + pos[(int64_t)t * D + d];
  }
}
void launch_clip_assemble(const float* patch, const float* cls, const float* pos, float* x, int B, int n, int D,
                          hipStream_t s, int rows) {
  if (rows <= 0) rows = n + 1;
  const int64_t total = (int64_t)B * rows * D;
  const int grid = (int)(cdiv64(total, 256) < 8192 ? cdiv64(total, 256) : 8192);
  hipLaunchKernelGGL(clip_assemble_kernel, dim3(grid), dim3(256), 0, s, patch, cls, pos, x, B, n, D, rows);
}

// f-4: ImageBind audio stem (imagebind_model.py:175-192: Conv2d(1, D, k, stride < k, bias=False) on the mel
// spectrogram) as im2col + GEMM: img f32 [n, 1, Hh, Ww] -> out T [n * gh * gw, k * k], patches overlap
template <typename T>
__global__ void im2col_conv1_kernel(const float* __restrict__ img, int n, int Hh, int Ww, int k, int st, int gh, int gw,
                                    T* __restrict__ out) {
  const int KK = k * k;
  const int64_t total = (int64_t)n * gh * gw * KK;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % KK);
    const int64_t row = i / KK;
    const int px = (int)(row % gw), py = (int)((row / gw) % gh), b = (int)(row / ((int64_t)gh * gw));
    st1<T>(out + row * KK, e, img[((int64_t)b * Hh + py * st + e / k) * Ww + px * st + e % k]);
  }
}
template <typename T>
void launch_im2col_conv1(const float* img, int n, int Hh, int Ww, int k, int st, void* out, hipStream_t s) {
  const int gh = (Hh - k) / st + 1, gw = (Ww - k) / st + 1;
  const int64_t total = (int64_t)n * gh * gw * k * k;
  const int grid = (int)(cdiv64(total, 256) < 4096 ? cdiv64(total, 256) : 4096);
  hipLaunchKernelGGL((im2col_conv1_kernel<T>), dim3(grid), dim3(256), 0, s, img, n, Hh, Ww, k, st, gh, gw,
                     reinterpret_cast<T*>(out));
}
template void launch_im2col_conv1<float>(const float*, int, int, int, int, int, void*, hipStream_t);
template void launch_im2col_conv1<bf16>(const float*, int, int, int, int, int, void*, hipStream_t);
template void launch_im2col_conv1<sp16>(const float*, int, int, int, int, int, void*, hipStream_t);

// f-4: ImageBind audio head tail (imagebind_model.py:425-428): y = x / max(||x||_2, 1e-12) * scale, one block per row
__global__ __launch_bounds__(256) void l2norm_scale_kernel(const float* __restrict__ x, int D, float scale,
                                                           float* __restrict__ y) {
  const float* r = x + (int64_t)blockIdx.x * D;
  float ss = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) ss += r[d] * r[d];
  ss = wave_sum(ss);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float nrm = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);
  for (int d = threadIdx.x; d < D; d += 256) y[(int64_t)blockIdx.x * D + d] = r[d] / nrm * scale;
}
void launch_l2norm_scale(const float* x, int rows, int D, float scale, float* y, hipStream_t s) {
  if (rows > 0) hipLaunchKernelGGL(l2norm_scale_kernel, dim3(rows), dim3(256), 0, s, x, D, scale, y);
}

// ---------------------------------------------------------------------------------------------
// token embedding gather + image splice.  One block per output row.
// ---------------------------------------------------------------------------------------------
__global__ void embed_splice_kernel(const int64_t* __restrict__ ids, const int* __restrict__ lens, int B,
                                    int Lmax, const void* __restrict__ table, int is_bf16, int vocab,
                                    const float* __restrict__ img_feat, int n_img, float* __restrict__ x,
                                    int Smax, int D, int* __restrict__ out_len) {
  const int b = blockIdx.y, srow = blockIdx.x;
  const int L = lens[b];
  const int64_t* row = ids + (int64_t)b * Lmax;
  // position of the (single) image placeholder, -1 if none
  __shared__ int ip_s;
  if (threadIdx.x == 0) {
    int ip = -1;
    for (int i = 0; i < L; ++i)
      if (row[i] == -200) {
        ip = i;
        break;
      }
    ip_s = ip;
    if (srow == 0) out_len[b] = ip >= 0 ? L + n_img - 1 : L;
  }
  __syncthreads();
  const int ip = ip_s;
  const int S = ip >= 0 ? L + n_img - 1 : L;
  float* dst = x + ((int64_t)b * Smax + srow) * D;
  if (srow >= S) return;
  if (ip >= 0 && srow >= ip && srow < ip + n_img) {
    const float* src = img_feat + ((int64_t)b * n_img + (srow - ip)) * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) dst[d] = src[d];
    return;
  }
  const int tpos = (ip >= 0 && srow >= ip + n_img) ? srow - n_img + 1 : srow;
  int64_t id = row[tpos];
  if (id < 0 || id >= vocab) {  // other placeholders: rows are overwritten by scatter_rows
    for (int d = threadIdx.x; d < D; d += blockDim.x) dst[d] = 0.f;
    return;
  }
  if (is_bf16) {
    const bf16* src = reinterpret_cast<const bf16*>(table) + id * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) dst[d] = bf2f(src[d]);
  } else {
    const float* src = reinterpret_cast<const float*>(table) + id * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) dst[d] = src[d];
  }
}
void launch_embed_splice(const int64_t* ids, const int* lens, int B, int Lmax, const void* emb_table,
                         int emb_is_bf16, int vocab, const float* img_feat, int n_img, float* x, int Smax, int D,
                         int* out_len, hipStream_t s) {
  dim3 grid(Lmax + n_img, B);
  if ((int)grid.x > Smax) grid.x = Smax;
  hipLaunchKernelGGL(embed_splice_kernel, grid, dim3(256), 0, s, ids, lens, B, Lmax, emb_table, emb_is_bf16,
                     vocab, img_feat, n_img, x, Smax, D, out_len);
}

__global__ void scatter_rows_kernel(const float* __restrict__ rows, const int* __restrict__ db,
                                    const int* __restrict__ dp, float* __restrict__ x, int Smax, int D) {
  const int i = blockIdx.x;
  float* dst = x + ((int64_t)db[i] * Smax + dp[i]) * D;
  const float* src = rows + (int64_t)i * D;
  for (int d = threadIdx.x; d < D; d += blockDim.x) dst[d] = src[d];
}
void launch_scatter_rows(const float* rows, const int* dst_b, const int* dst_pos, int n, float* x, int Smax,
                         int D, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(n), dim3(256), 0, s, rows, dst_b, dst_pos, x, Smax, D);
}

__global__ void gather_rows_kernel(const float* __restrict__ x, int Smax, int D, const int* __restrict__ b,
                                   const int* __restrict__ pos, float* __restrict__ out) {
  const int i = blockIdx.x;
  const float* src = x + ((int64_t)b[i] * Smax + pos[i]) * D;
  for (int d = threadIdx.x; d < D; d += blockDim.x) out[(int64_t)i * D + d] = src[d];
}
void launch_gather_rows(const float* x, int Smax, int D, const int* b, const int* pos, int n, float* out,
                        hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, s, x, Smax, D, b, pos, out);
}

// ---------------------------------------------------------------------------------------------
// RoPE (HF rotate_half form) + KV-cache append
// ---------------------------------------------------------------------------------------------
template <typename T, typename TIN>
__global__ void rope_cache_kernel(const TIN* __restrict__ qkv, int B, int S, int H, int hd,
                                  const int* __restrict__ pos0, const int* __restrict__ lens,
                                  const float* __restrict__ cs_tab, T* __restrict__ q_out, T* __restrict__ kc,
                                  T* __restrict__ vc, int maxS, T* __restrict__ q_keep) {
  // one block per (row, b); threads over (h, d < hd/2)
  const int srow = blockIdx.x, b = blockIdx.y;
  if (lens && srow >= lens[b]) return;
  const int pos = (pos0 ? pos0[b] : 0) + srow;
  const int half = hd / 2;
  const TIN* base = qkv + ((int64_t)b * S + srow) * 3 * H * hd;
  for (int i = threadIdx.x; i < H * half; i += blockDim.x) {
    const int h = i / half, d = i % half;
    // host-built table [maxS][2][hd/2] (cos | sin), same fp32 op order as HF's rotary embedding
    const float cs = cs_tab[((int64_t)pos * 2) * half + d], sn = cs_tab[((int64_t)pos * 2 + 1) * half + d];
    const float q1 = to_f32<TIN>(base[h * hd + d]), q2 = to_f32<TIN>(base[h * hd + d + half]);
    const float k1 = to_f32<TIN>(base[(H + h) * hd + d]), k2 = to_f32<TIN>(base[(H + h) * hd + d + half]);
    const float v1 = to_f32<TIN>(base[(2 * H + h) * hd + d]),
                v2 = to_f32<TIN>(base[(2 * H + h) * hd + d + half]);
    T* qo = q_out + (((int64_t)b * S + srow) * H + h) * hd;
    qo[d] = from_f32<T>(q1 * cs - q2 * sn);
    qo[d + half] = from_f32<T>(q2 * cs + q1 * sn);
    const int64_t co = (((int64_t)b * maxS + pos) * H + h) * hd;
    if (q_keep) {
      q_keep[co + d] = qo[d];
      q_keep[co + d + half] = qo[d + half];
    }
    kc[co + d] = from_f32<T>(k1 * cs - k2 * sn);
    kc[co + d + half] = from_f32<T>(k2 * cs + k1 * sn);
    vc[co + d] = from_f32<T>(v1);
    vc[co + d + half] = from_f32<T>(v2);
  }
}
// bf16 rows, 16-byte accesses: a lane takes 8 consecutive d of the first half and the 8 partners of the second half
// (six 16-byte loads, six stores) -- the same f32 operations as the scalar kernel above, bit-identical results.
template <bool SLABS>
__global__ __launch_bounds__(256) void rope_cache_vec_kernel(const bf16* __restrict__ qkv, const float* __restrict__ slab0,
                                                             const float* __restrict__ slab1, int S, int H, int hd,
                                                             const int* __restrict__ pos0, const int* __restrict__ lens,
                                                             const float* __restrict__ cs_tab, bf16* __restrict__ q_out,
                                                             bf16* __restrict__ kc, bf16* __restrict__ vc, int maxS,
                                                             bf16* __restrict__ q_keep) {
  const int srow = blockIdx.x, b = blockIdx.y;
  if (lens && srow >= lens[b]) return;
  const int pos = (pos0 ? pos0[b] : 0) + srow;
  const int half = hd / 2, per = half / 8;  // lanes per head
  const int64_t roff = ((int64_t)b * S + srow) * 3 * H * hd;
  const bf16* base = qkv + roff;
  // 8 values at column c of this row: the GEMM's bf16 output, or the sum of its two f32 K slices rounded the same way
  auto ld8 = [&](int c, float (&o)[8]) {
    if constexpr (SLABS) {
      const float4v a0 = *reinterpret_cast<const float4v*>(slab0 + roff + c), a1 = *reinterpret_cast<const float4v*>(slab0 + roff + c + 4);
      const float4v b0 = *reinterpret_cast<const float4v*>(slab1 + roff + c), b1 = *reinterpret_cast<const float4v*>(slab1 + roff + c + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = bf2f(f2bf(a0[e] + b0[e]));
        o[4 + e] = bf2f(f2bf(a1[e] + b1[e]));
      }
    } else {
      Vec16<bf16>::unpack(*reinterpret_cast<const uint4v*>(base + c), o);
    }
  };
  for (int i = threadIdx.x; i < H * per; i += blockDim.x) {
    const int h = i / per, d = (i % per) * 8;
    float cs[8], sn[8];
    *reinterpret_cast<float4v*>(cs) = *reinterpret_cast<const float4v*>(cs_tab + ((int64_t)pos * 2) * half + d);
    *reinterpret_cast<float4v*>(cs + 4) = *reinterpret_cast<const float4v*>(cs_tab + ((int64_t)pos * 2) * half + d + 4);
    *reinterpret_cast<float4v*>(sn) = *reinterpret_cast<const float4v*>(cs_tab + ((int64_t)pos * 2 + 1) * half + d);
    *reinterpret_cast<float4v*>(sn + 4) = *reinterpret_cast<const float4v*>(cs_tab + ((int64_t)pos * 2 + 1) * half + d + 4);
    float q1[8], q2[8], k1[8], k2[8], va[8], vb[8];
    ld8(h * hd + d, q1);
    ld8(h * hd + d + half, q2);
    ld8((H + h) * hd + d, k1);
    ld8((H + h) * hd + d + half, k2);
    ld8((2 * H + h) * hd + d, va);
    ld8((2 * H + h) * hd + d + half, vb);
    bf16 qa[8], qb[8], ka[8], kb[8], v1[8], v2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v1[e] = from_f32<bf16>(va[e]);   // exact: va / vb are bf16 values
      v2[e] = from_f32<bf16>(vb[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      qa[e] = from_f32<bf16>(q1[e] * cs[e] - q2[e] * sn[e]);
      qb[e] = from_f32<bf16>(q2[e] * cs[e] + q1[e] * sn[e]);
      ka[e] = from_f32<bf16>(k1[e] * cs[e] - k2[e] * sn[e]);
      kb[e] = from_f32<bf16>(k2[e] * cs[e] + k1[e] * sn[e]);
    }
    bf16* qo = q_out + (((int64_t)b * S + srow) * H + h) * hd;
    *reinterpret_cast<uint4v*>(qo + d) = *reinterpret_cast<const uint4v*>(qa);
    *reinterpret_cast<uint4v*>(qo + d + half) = *reinterpret_cast<const uint4v*>(qb);
    const int64_t co = (((int64_t)b * maxS + pos) * H + h) * hd;
    if (q_keep) {
      *reinterpret_cast<uint4v*>(q_keep + co + d) = *reinterpret_cast<const uint4v*>(qa);
      *reinterpret_cast<uint4v*>(q_keep + co + d + half) = *reinterpret_cast<const uint4v*>(qb);
    }
    *reinterpret_cast<uint4v*>(kc + co + d) = *reinterpret_cast<const uint4v*>(ka);
    *reinterpret_cast<uint4v*>(kc + co + d + half) = *reinterpret_cast<const uint4v*>(kb);
    *reinterpret_cast<uint4v*>(vc + co + d) = *reinterpret_cast<const uint4v*>(v1);
    *reinterpret_cast<uint4v*>(vc + co + d + half) = *reinterpret_cast<const uint4v*>(v2);
  }
}
template <typename T>
void launch_rope_cache(const void* qkv, int B, int S, int H, int hd, const int* pos0, const int* lens,
                       const float* cs_tab, void* q_out, void* kc, void* vc, int maxS, void* q_keep,
                       hipStream_t s) {
  if constexpr (sizeof(T) == 2) {
    const bool al = !(((uintptr_t)qkv | (uintptr_t)q_out | (uintptr_t)kc | (uintptr_t)vc | (uintptr_t)q_keep |
                       (uintptr_t)cs_tab) & 15);
    if (hd % 16 == 0 && al) {
      hipLaunchKernelGGL(rope_cache_vec_kernel<false>, dim3(S, B), dim3(256), 0, s, reinterpret_cast<const bf16*>(qkv),
                         (const float*)nullptr, (const float*)nullptr, S, H, hd,
                         pos0, lens, cs_tab, reinterpret_cast<bf16*>(q_out), reinterpret_cast<bf16*>(kc),
                         reinterpret_cast<bf16*>(vc), maxS, reinterpret_cast<bf16*>(q_keep));
      return;
    }
  }
  hipLaunchKernelGGL((rope_cache_kernel<T, T>), dim3(S, B), dim3(256), 0, s, reinterpret_cast<const T*>(qkv), B,
                     S, H, hd, pos0, lens, cs_tab, reinterpret_cast<T*>(q_out), reinterpret_cast<T*>(kc),
                     reinterpret_cast<T*>(vc), maxS, reinterpret_cast<T*>(q_keep));
}
// f32 q / k / v (ANYREF_MODE_PARITY16: f32 attention operands and KV cache) from the two f32 K slices of the projection: a lane
// takes 4 consecutive d of the first half and their partners, 16-byte accesses; the rotation in the scalar kernel's f32 order
__global__ __launch_bounds__(256) void rope_cache_slabs_f32_kernel(const float* __restrict__ slab0, const float* __restrict__ slab1,
                                                                   int S, int H, int hd, const int* __restrict__ pos0,
                                                                   const int* __restrict__ lens, const float* __restrict__ cs_tab,
                                                                   float* __restrict__ q_out, float* __restrict__ kc,
                                                                   float* __restrict__ vc, int maxS, float* __restrict__ q_keep) {
  const int srow = blockIdx.x, b = blockIdx.y;
  if (lens && srow >= lens[b]) return;
  const int pos = (pos0 ? pos0[b] : 0) + srow;
  const int half = hd / 2, per = half / 4;  // lanes per head
  const int64_t roff = ((int64_t)b * S + srow) * 3 * H * hd;
  auto ld4 = [&](int c) { return *reinterpret_cast<const float4v*>(slab0 + roff + c) + *reinterpret_cast<const float4v*>(slab1 + roff + c); };
  for (int i = threadIdx.x; i < H * per; i += blockDim.x) {
    const int h = i / per, d = (i % per) * 4;
    const float4v cs = *reinterpret_cast<const float4v*>(cs_tab + ((int64_t)pos * 2) * half + d);
    const float4v sn = *reinterpret_cast<const float4v*>(cs_tab + ((int64_t)pos * 2 + 1) * half + d);
    const float4v q1 = ld4(h * hd + d), q2 = ld4(h * hd + d + half);
    const float4v k1 = ld4((H + h) * hd + d), k2 = ld4((H + h) * hd + d + half);
    const float4v v1 = ld4((2 * H + h) * hd + d), v2 = ld4((2 * H + h) * hd + d + half);
    float4v qa, qb, ka, kb;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qa[e] = q1[e] * cs[e] - q2[e] * sn[e];
      qb[e] = q2[e] * cs[e] + q1[e] * sn[e];
      ka[e] = k1[e] * cs[e] - k2[e] * sn[e];
      kb[e] = k2[e] * cs[e] + k1[e] * sn[e];
    }
    float* qo = q_out + (((int64_t)b * S + srow) * H + h) * hd;
    *reinterpret_cast<float4v*>(qo + d) = qa;
    *reinterpret_cast<float4v*>(qo + d + half) = qb;
    const int64_t co = (((int64_t)b * maxS + pos) * H + h) * hd;
    if (q_keep) {
      *reinterpret_cast<float4v*>(q_keep + co + d) = qa;
      *reinterpret_cast<float4v*>(q_keep + co + d + half) = qb;
    }
    *reinterpret_cast<float4v*>(kc + co + d) = ka;
    *reinterpret_cast<float4v*>(kc + co + d + half) = kb;
    *reinterpret_cast<float4v*>(vc + co + d) = v1;
    *reinterpret_cast<float4v*>(vc + co + d + half) = v2;
  }
}
void launch_rope_cache_slabs(const float* slab0, const float* slab1, int B, int S, int H, int hd, const int* pos0,
                             const int* lens, const float* cs_tab, void* q_out, void* kc, void* vc, int maxS, void* q_keep,
                             hipStream_t s, bool out_f32) {
  if (hd % 16 || (((uintptr_t)slab0 | (uintptr_t)slab1 | (uintptr_t)q_out | (uintptr_t)kc | (uintptr_t)vc | (uintptr_t)q_keep |
                   (uintptr_t)cs_tab) & 15))
    throw std::runtime_error("rope_cache_slabs: head dim % 16 and 16-byte aligned buffers");
  if (out_f32) {
    hipLaunchKernelGGL(rope_cache_slabs_f32_kernel, dim3(S, B), dim3(256), 0, s, slab0, slab1, S, H, hd, pos0, lens, cs_tab,
                       reinterpret_cast<float*>(q_out), reinterpret_cast<float*>(kc), reinterpret_cast<float*>(vc), maxS,
                       reinterpret_cast<float*>(q_keep));
    return;
  }
  hipLaunchKernelGGL(rope_cache_vec_kernel<true>, dim3(S, B), dim3(256), 0, s, (const bf16*)nullptr, slab0, slab1, S, H, hd,
                     pos0, lens, cs_tab, reinterpret_cast<bf16*>(q_out), reinterpret_cast<bf16*>(kc),
                     reinterpret_cast<bf16*>(vc), maxS, reinterpret_cast<bf16*>(q_keep));
}
template void launch_rope_cache<float>(const void*, int, int, int, int, const int*, const int*, const float*,
                                       void*, void*, void*, int, void*, hipStream_t);
template void launch_rope_cache<bf16>(const void*, int, int, int, int, const int*, const int*, const float*,
                                      void*, void*, void*, int, void*, hipStream_t);
template <typename T>
void launch_rope_cache_f32(const float* qkv, int B, int H, int hd, const int* pos, const float* cs_tab,
                           void* q_out, void* kc, void* vc, int maxS, void* q_keep, hipStream_t s) {
  hipLaunchKernelGGL((rope_cache_kernel<T, float>), dim3(1, B), dim3(256), 0, s, qkv, B, 1, H, hd, pos, nullptr,
                     cs_tab, reinterpret_cast<T*>(q_out), reinterpret_cast<T*>(kc), reinterpret_cast<T*>(vc),
                     maxS, reinterpret_cast<T*>(q_keep));
}
template void launch_rope_cache_f32<float>(const float*, int, int, int, const int*, const float*, void*, void*,
                                           void*, int, void*, hipStream_t);
template void launch_rope_cache_f32<bf16>(const float*, int, int, int, const int*, const float*, void*, void*,
                                          void*, int, void*, hipStream_t);

__global__ void embed_rows_kernel(const int64_t* __restrict__ ids, const void* __restrict__ table, int is_bf16,
                                  int D, float* __restrict__ x) {
  const int b = blockIdx.x;
  const int64_t id = ids[b];
  if (is_bf16) {
    const bf16* src = reinterpret_cast<const bf16*>(table) + id * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) x[(int64_t)b * D + d] = bf2f(src[d]);
  } else {
    const float* src = reinterpret_cast<const float*>(table) + id * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) x[(int64_t)b * D + d] = src[d];
  }
}
void launch_embed_rows(const int64_t* ids, int B, const void* table, int is_bf16, int D, float* x,
                       hipStream_t s) {
  hipLaunchKernelGGL(embed_rows_kernel, dim3(B), dim3(256), 0, s, ids, table, is_bf16, D, x);
}
__global__ void decode_index_kernel(const int* __restrict__ pos, int B, int maxS, int* __restrict__ row_map,
                                    int* __restrict__ kvlen) {
  const int b = threadIdx.x;
  if (b < B) {
    row_map[b] = b * maxS + pos[b];
    kvlen[b] = pos[b] + 1;
  }
}
void launch_decode_index(const int* pos, int B, int maxS, int* row_map, int* kvlen, hipStream_t s) {
  hipLaunchKernelGGL(decode_index_kernel, dim3(1), dim3(64), 0, s, pos, B, maxS, row_map, kvlen);
}
__global__ void build_tokens_kernel(const float* __restrict__ out_tokens, int n_out, const float* __restrict__ pred,
                                    int C, float* __restrict__ tokens) {
  const int i = blockIdx.y, t = blockIdx.x;
  const float* src = t < n_out ? out_tokens + (int64_t)t * C : pred + (int64_t)i * C;
  float* dst = tokens + ((int64_t)i * (n_out + 1) + t) * C;
  for (int d = threadIdx.x; d < C; d += blockDim.x) dst[d] = src[d];
}
void launch_build_tokens(const float* out_tokens, int n_out, const float* pred, int n, int C, float* tokens,
                         hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(build_tokens_kernel, dim3(n_out + 1, n), dim3(256), 0, s, out_tokens, n_out, pred, C, tokens);
}

template <typename T>
__global__ void swiglu_kernel(const T* __restrict__ gu, int M, int F, T* __restrict__ out) {
  const int64_t total = (int64_t)M * F;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / F;
    const int j = (int)(i % F);
    const float g = to_f32<T>(gu[m * 2 * F + j]), u = to_f32<T>(gu[m * 2 * F + F + j]);
    out[i] = from_f32<T>(apply_act(g, ACT_SILU) * u);
  }
}
template <typename T>
void launch_swiglu(const void* gu, int M, int F, void* out, hipStream_t s) {
  const int64_t total = (int64_t)M * F;
  if (total <= 0) return;
  const int grid = (int)(cdiv64(total, 256) < 8192 ? cdiv64(total, 256) : 8192);
  hipLaunchKernelGGL((swiglu_kernel<T>), dim3(grid), dim3(256), 0, s, reinterpret_cast<const T*>(gu), M, F,
                     reinterpret_cast<T*>(out));
}
template void launch_swiglu<float>(const void*, int, int, void*, hipStream_t);
template void launch_swiglu<bf16>(const void*, int, int, void*, hipStream_t);

// argmax, first index on ties (torch.argmax on CPU returns the first maximal index).  One 1024-thread
// workgroup per row, 16-byte loads all issued before the first compare (the row is read once, the
// kernel is pure latency); `bump` (optional) is a per-row counter incremented by one -- the decode
// step's position -- so the step needs no separate increment launch.
__device__ __forceinline__ void argmax_take(float v, int i, float& best, int& bi) {
  if (v > best || (v == best && i < bi)) {
    best = v;
    bi = i;
  }
}
// `nx` (optional): the decode step's bookkeeping for the NEXT step rides in the same launch -- embedding row of
// the chosen token -> nx.x[b], KV-cache row index and key count of the (bumped) position -- instead of two more
// one-workgroup launches at the head of every step.
struct ArgmaxNext {
  const void* table = nullptr;  // [vocab, D] bf16 or f32
  int is_bf16 = 0, D = 0, maxS = 0;
  float* x = nullptr;           // [B, D]
  int* row_map = nullptr;       // [B] b * maxS + pos
  int* kvlen = nullptr;         // [B] pos + 1
};
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ x, int N, int ldx,
                                                      int64_t* __restrict__ out, int* __restrict__ bump,
                                                      ArgmaxNext nx) {
  constexpr int NT = 1024, U = 8;  // 8 x 1024 float4: a 32k-entry row in one round trip
  const float* row = x + (int64_t)blockIdx.x * ldx;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  if ((ldx & 3) == 0 && ((uintptr_t)x & 15) == 0) {
    const int n4 = N >> 2;
    for (int i0 = 0; i0 < n4; i0 += NT * U) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * NT + tid;
        v[u] = i < n4 ? reinterpret_cast<const float4*>(row)[i] : float4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = (i0 + u * NT + tid) * 4;
        argmax_take(v[u].x, i, best, bi);
        argmax_take(v[u].y, i + 1, best, bi);
        argmax_take(v[u].z, i + 2, best, bi);
        argmax_take(v[u].w, i + 3, best, bi);
      }
    }
    for (int i = n4 * 4 + tid; i < N; i += NT) argmax_take(row[i], i, best, bi);
  } else {
    for (int i = tid; i < N; i += NT) argmax_take(row[i], i, best, bi);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v = __shfl_xor(best, o, 64);
    const int i = __shfl_xor(bi, o, 64);
    argmax_take(v, i, best, bi);
  }
  __shared__ float sv[NT / 64];
  __shared__ int si[NT / 64];
  if (lane == 0) {
    sv[wave] = best;
    si[wave] = bi;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < NT / 64; ++w) argmax_take(sv[w], si[w], best, bi);
    out[blockIdx.x] = bi;
    si[0] = bi;
    if (bump) {
      const int p = bump[blockIdx.x] + 1;
      bump[blockIdx.x] = p;
      if (nx.x) {
        nx.row_map[blockIdx.x] = blockIdx.x * nx.maxS + p;
        nx.kvlen[blockIdx.x] = p + 1;
      }
    }
  }
  if (!nx.x) return;
  __syncthreads();
  const int64_t id = si[0];
  float* dst = nx.x + (int64_t)blockIdx.x * nx.D;
  if (nx.is_bf16) {
    const bf16* src = reinterpret_cast<const bf16*>(nx.table) + id * nx.D;
    for (int d = tid; d < nx.D; d += NT) dst[d] = bf2f(src[d]);
  } else {
    const float* src = reinterpret_cast<const float*>(nx.table) + id * nx.D;
    for (int d = tid; d < nx.D; d += NT) dst[d] = src[d];
  }
}
void launch_argmax(const float* x, int M, int N, int ldx, int64_t* out, hipStream_t s, int* bump) {
  if (M <= 0) return;
  hipLaunchKernelGGL(argmax_kernel, dim3(M), dim3(1024), 0, s, x, N, ldx, out, bump, ArgmaxNext{});
}
void launch_argmax_next(const float* x, int M, int N, int ldx, int64_t* out, int* pos, const void* table, int is_bf16,
                        int D, int maxS, float* x_next, int* row_map, int* kvlen, hipStream_t s) {
  if (M <= 0) return;
  ArgmaxNext nx;
  nx.table = table; nx.is_bf16 = is_bf16; nx.D = D; nx.maxS = maxS; nx.x = x_next; nx.row_map = row_map; nx.kvlen = kvlen;
  hipLaunchKernelGGL(argmax_kernel, dim3(M), dim3(1024), 0, s, x, N, ldx, out, pos, nx);
}

// ---------------------------------------------------------------------------------------------
// mask-decoder upscaler tails
// ---------------------------------------------------------------------------------------------
// One wave per output pixel: un-shuffle ConvT(k2,s2) GEMM output, LayerNorm2d over C, GELU.
template <typename T>
__global__ __launch_bounds__(256) void upscale1_kernel(const float* __restrict__ tmp, int n, int g, int C,
                                                       const float* __restrict__ ln_g,
                                                       const float* __restrict__ ln_b, float eps,
                                                       T* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t pix = (int64_t)blockIdx.x * 4 + wave;  // over n * 2g * 2g
  const int G = 2 * g;
  if (pix >= (int64_t)n * G * G) return;
  const int X = (int)(pix % G), Y = (int)((pix / G) % G), i = (int)(pix / ((int64_t)G * G));
  const float* src = tmp + (((int64_t)i * g + Y / 2) * g + X / 2) * 4 * C + ((Y & 1) * 2 + (X & 1)) * C;
  // C <= 128: up to two values per lane
  float v0 = lane < C ? src[lane] : 0.f, v1 = lane + 64 < C ? src[lane + 64] : 0.f;
  const float mean = wave_sum(v0 + v1) / (float)C;
  const float c0 = lane < C ? v0 - mean : 0.f, c1 = lane + 64 < C ? v1 - mean : 0.f;
  const float var = wave_sum(c0 * c0 + c1 * c1) / (float)C;
  const float inv = 1.f / sqrtf(var + eps);
  T* dst = out + pix * C;
  if (lane < C) dst[lane] = from_f32<T>(apply_act(c0 * inv * ln_g[lane] + ln_b[lane], ACT_GELU));
  if (lane + 64 < C) dst[lane + 64] = from_f32<T>(apply_act(c1 * inv * ln_g[lane + 64] + ln_b[lane + 64], ACT_GELU));
}
template <typename T>
void launch_upscale1(const float* tmp, int n, int g, int C, const float* ln_g, const float* ln_b, float eps,
                     void* out, hipStream_t s) {
  if (C > 128) throw std::runtime_error("upscale1: C > 128");
  const int64_t pix = (int64_t)n * 4 * g * g;
  hipLaunchKernelGGL((upscale1_kernel<T>), dim3((unsigned)cdiv64(pix, 4)), dim3(256), 0, s, tmp, n, g, C, ln_g,
                     ln_b, eps, reinterpret_cast<T*>(out));
}
template void launch_upscale1<float>(const float*, int, int, int, const float*, const float*, float, void*,
                                     hipStream_t);
template void launch_upscale1<bf16>(const float*, int, int, int, const float*, const float*, float, void*,
                                    hipStream_t);

// masks[i,t,Y,X] = sum_c hyper[i,t,c] * gelu(tmp[i, (Y/2,X/2), (Y%2*2+X%2)*C + c]); thread per pixel.
__global__ void upscale2_masks_kernel(const float* __restrict__ tmp, const float* __restrict__ hyper, int n,
                                      int ntok, int g2, int C, float* __restrict__ masks) {
  extern __shared__ float hs[];  // [ntok*C] of prompt i
  const int i = blockIdx.y;
  for (int j = threadIdx.x; j < ntok * C; j += blockDim.x) hs[j] = hyper[(int64_t)i * ntok * C + j];
  __syncthreads();
  const int G = 2 * g2;
  const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (int64_t)G * G) return;
  const int X = (int)(pix % G), Y = (int)(pix / G);
  const float* src = tmp + (((int64_t)i * g2 + Y / 2) * g2 + X / 2) * 4 * C + ((Y & 1) * 2 + (X & 1)) * C;
  float acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = 0.f;
  for (int c = 0; c < C; ++c) {
    const float u = apply_act(src[c], ACT_GELU);
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (t < ntok) acc[t] = fmaf(hs[t * C + c], u, acc[t]);
  }
  for (int t = 0; t < ntok; ++t) masks[(((int64_t)i * ntok + t) * G + Y) * G + X] = acc[t];
}
void launch_upscale2_masks(const float* tmp, const float* hyper, int n, int ntok, int g2, int C, float* masks,
                           hipStream_t s) {
  if (ntok > 8) throw std::runtime_error("upscale2: more than 8 mask tokens");
  const int G = 2 * g2;
  dim3 grid((unsigned)cdiv64((int64_t)G * G, 256), n);
  hipLaunchKernelGGL(upscale2_masks_kernel, grid, dim3(256), ntok * C * sizeof(float), s, tmp, hyper, n, ntok,
                     g2, C, masks);
}

// ---------------------------------------------------------------------------------------------
// Sam.postprocess_masks: bilinear(lh x lw -> S x S), crop [:rh,:rw], bilinear -> (H,W).
// PyTorch upsample_bilinear2d, align_corners=False: src = max(scale*(dst+0.5)-0.5, 0),
// i0 = floor(src), i1 = min(i0+1, in-1), lambda = src - i0; scale = in/out in fp32.
// ---------------------------------------------------------------------------------------------
__device__ inline void bil_idx(int dst, float scale, int in, int& i0, int& i1, float& l1) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 < in - 1 ? i0 + 1 : i0;
  l1 = src - (float)i0;
}
__device__ inline float bil_stage1(const float* __restrict__ low, int lh, int lw, float sy, float sx, int y,
                                   int x) {
  int y0, y1, x0, x1;
  float ly, lx;
  bil_idx(y, sy, lh, y0, y1, ly);
  bil_idx(x, sx, lw, x0, x1, lx);
  const float a = low[y0 * lw + x0], b = low[y0 * lw + x1], c = low[y1 * lw + x0], d = low[y1 * lw + x1];
  return (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * c + lx * d);
}
__global__ void postprocess_kernel(const float* __restrict__ low, int64_t lstride, int lh, int lw, int S, int rh,
                                   int rw, int H, int W, float* __restrict__ out) {
  const int i = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (int64_t)H * W) return;
  const int X = (int)(p % W), Y = (int)(p / W);
  const float* l = low + (int64_t)i * lstride;
  const float s1y = (float)lh / (float)S, s1x = (float)lw / (float)S;
  const float s2y = (float)rh / (float)H, s2x = (float)rw / (float)W;
  int y0, y1, x0, x1;
  float ly, lx;
  bil_idx(Y, s2y, rh, y0, y1, ly);
  bil_idx(X, s2x, rw, x0, x1, lx);
  const float a = bil_stage1(l, lh, lw, s1y, s1x, y0, x0), b = bil_stage1(l, lh, lw, s1y, s1x, y0, x1);
  const float c = bil_stage1(l, lh, lw, s1y, s1x, y1, x0), d = bil_stage1(l, lh, lw, s1y, s1x, y1, x1);
  out[(int64_t)i * H * W + p] = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * c + lx * d);
}
void launch_postprocess(const float* low, int64_t lstride, int n, int lh, int lw, int S, int rh, int rw, int H,
                        int W, float* out, hipStream_t s) {
  if (n <= 0) return;
  dim3 grid((unsigned)cdiv64((int64_t)H * W, 256), n);
  hipLaunchKernelGGL(postprocess_kernel, grid, dim3(256), 0, s, low, lstride, lh, lw, S, rh, rw, H, W, out);
}

// dense PE: out[(y*g+x), f] = sin(2pi*(cx*G[0,f] + cy*G[1,f])), out[.., F+f] = cos(..); c = 2*((i+.5)/g)-1
__global__ void dense_pe_kernel(const float* __restrict__ gauss, int g, int F, float* __restrict__ out) {
  const int64_t total = (int64_t)g * g * F;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % F);
    const int x = (int)((i / F) % g), y = (int)(i / ((int64_t)F * g));
    const float cx = 2.f * (((float)x + 0.5f) / (float)g) - 1.f, cy = 2.f * (((float)y + 0.5f) / (float)g) - 1.f;
    const float v = 6.283185307179586f * (cx * gauss[f] + cy * gauss[F + f]);
    out[((int64_t)y * g + x) * 2 * F + f] = sinf(v);
    out[((int64_t)y * g + x) * 2 * F + F + f] = cosf(v);
  }
}
void launch_dense_pe(const float* gauss, int g, int F, float* out, hipStream_t s) {
  const int64_t total = (int64_t)g * g * F;
  hipLaunchKernelGGL(dense_pe_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, s, gauss, g, F, out);
}

__global__ void fill_i32_kernel(int* p, int v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
void launch_fill_i32(int* p, int v, int n, hipStream_t s) {
  hipLaunchKernelGGL(fill_i32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, p, v, n);
}
__global__ void add_i32_kernel(int* p, int v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += v;
}
void launch_add_i32(int* p, int v, int n, hipStream_t s) {
  hipLaunchKernelGGL(add_i32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, p, v, n);
}

// y[:] += w * sum_{j in [s0,e0)} (p[j]/sum p) * X[j,:]
__global__ void rephrase_kernel(const float* __restrict__ hidden, int D, const float* __restrict__ attn, int s0,
                                int e0, float weight, float* __restrict__ y) {
  float tot = 0.f;
  for (int j = s0; j < e0; ++j) tot += attn[j];
  for (int d = blockIdx.x * blockDim.x + threadIdx.x; d < D; d += gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int j = s0; j < e0; ++j) acc += hidden[(int64_t)j * D + d] * (attn[j] / tot);
    y[d] += weight * acc;
  }
}
void launch_rephrase(const float* hidden_b, int D, const float* attn_row, int s0, int e0, float weight, float* y,
                     hipStream_t s) {
  hipLaunchKernelGGL(rephrase_kernel, dim3(cdiv(D, 256)), dim3(256), 0, s, hidden_b, D, attn_row, s0, e0, weight,
                     y);
}

__global__ void to_f32_kernel(const void* __restrict__ in, int dtype, float* __restrict__ out, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v;
    if (dtype == 0)
      v = reinterpret_cast<const float*>(in)[i];
    else if (dtype == 1)
      v = bf2f(reinterpret_cast<const bf16*>(in)[i]);
    else
      v = (float)reinterpret_cast<const _Float16*>(in)[i];
    out[i] = v;
  }
}
void launch_to_f32(const void* in, int dtype, float* out, int64_t n, hipStream_t s) {
  if (n <= 0) return;
  const int grid = (int)(cdiv64(n, 256) < 16384 ? cdiv64(n, 256) : 16384);
  hipLaunchKernelGGL(to_f32_kernel, dim3(grid), dim3(256), 0, s, in, dtype, out, n);
}

// ---------------------------------------------------------------------------------------------
// The steps either side of the path, integer / exact-f32 work, HBM-bound.
// ---------------------------------------------------------------------------------------------
// f-2: `(sigmoid(logit) > 0.5).int()` + intersectionAndUnionGPU(pred, gt, K=2, ignore_index=255)
// (eval_referseg.py:189-208, utils/utils.py:79-91) fused: the full-resolution logits are read once
// and never leave HBM.  counts[mask] = {I0, I1, O0, O1, T0, T1}: I_c = #{pred == gt == c},
// O_c = #{pred == c, gt != 255}, T_c = #{gt == c}; union_c = O_c + T_c - I_c.  sigmoid(x) > 0.5 is
// evaluated as x > 0 (identical except for 0 < x < ~1.2e-7, where the fp32 sigmoid rounds to 0.5).
__global__ __launch_bounds__(256) void iou_counts_kernel(const float* __restrict__ logits,
                                                         const uint8_t* __restrict__ target, int64_t hw,
                                                         unsigned long long* __restrict__ counts) {
  const int m = blockIdx.y;
  const float* x = logits + (int64_t)m * hw;
  const uint8_t* t = target + (int64_t)m * hw;
  unsigned c[6] = {0, 0, 0, 0, 0, 0};
  auto take = [&](float v, unsigned g) {
    const unsigned p = v > 0.f ? 1u : 0u;
    if (g != 255u) {
      c[2 + p] += 1;
      if (g < 2u) {
        c[4 + g] += 1;
        if (g == p) c[p] += 1;
      }
    }
  };
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if ((hw & 3) == 0 && (((uintptr_t)x | (uintptr_t)t) & 15) == 0) {  // 4 pixels per step: 16 B of logits, 4 B of labels
    for (int64_t i = i0; i < hw / 4; i += stride) {
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      const uint32_t g = reinterpret_cast<const uint32_t*>(t)[i];
      take(v.x, g & 255u); take(v.y, (g >> 8) & 255u); take(v.z, (g >> 16) & 255u); take(v.w, g >> 24);
    }
  } else {
    for (int64_t i = i0; i < hw; i += stride) take(x[i], t[i]);
  }
  __shared__ unsigned red[4][6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    unsigned v = c[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6) {  // integer atomics: the result does not depend on the order of arrival
    const unsigned v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (v) atomicAdd(&counts[(int64_t)m * 6 + threadIdx.x], (unsigned long long)v);
  }
}
void launch_iou_counts(const float* logits, const uint8_t* target, int n, int64_t hw, int64_t* counts, hipStream_t s) {
  if (n <= 0) return;
  HIP_TRY(hipMemsetAsync(counts, 0, (size_t)n * 6 * sizeof(int64_t), s));
  if (hw <= 0) return;
  int64_t blocks = cdiv64(hw, 256 * 16);
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(iou_counts_kernel, dim3((unsigned)blocks, n), dim3(256), 0, s, logits, target, hw,
                     reinterpret_cast<unsigned long long*>(counts));
}

// f-2 (AVS): the per-pixel part of `mask_iou` and `Eval_Fmeasure` / `_eval_pr` (utils/pyutils.py:163-236) in one
// pass over the logits.  The reference thresholds sigmoid(x) at 0.5 (strictly) and at pr_num = 255 values of
// linspace(0, 1 - 1e-10); sigmoid is monotone, so every test `sigmoid(x) >= th_i` is `x >= cut_i` with
// cut_i = the smallest f32 whose f32 sigmoid reaches th_i.  The host finds the cuts by bisection against the same
// sigmoid the reference calls (anyref_amd/evalops.py), so the counts are exact, not "close".
//   conf[m]    = {n00, n01, n10, n11}  (index 2 * pred + gt, pred = x >= cut_pred, gt in {0, 1})
//   hist[m][b][g], b = #{i : x >= cut_i} in [0, nth]: pixels of label g whose sigmoid passes exactly the first b
//   thresholds; tp_i = sum_{b > i} hist[m][b][1], #{y_temp}_i = sum_{b > i} (hist[m][b][0] + hist[m][b][1]).
constexpr int AVS_MAX_TH = 255;
__global__ __launch_bounds__(256) void avs_counts_kernel(const float* __restrict__ logits,
                                                         const uint8_t* __restrict__ target, int64_t hw,
                                                         const float* __restrict__ cuts, int nth, float cut_pred,
                                                         unsigned long long* __restrict__ conf,
                                                         unsigned long long* __restrict__ hist) {
  __shared__ float cut_s[AVS_MAX_TH + 1];
  __shared__ unsigned h_s[(AVS_MAX_TH + 1) * 2];
  __shared__ unsigned c_s[4];
  const int m = blockIdx.y;
  for (int i = threadIdx.x; i <= AVS_MAX_TH; i += 256) cut_s[i] = i < nth ? cuts[i] : INFINITY;
  for (int i = threadIdx.x; i < (AVS_MAX_TH + 1) * 2; i += 256) h_s[i] = 0;
  if (threadIdx.x < 4) c_s[threadIdx.x] = 0;
  __syncthreads();
  const float* x = logits + (int64_t)m * hw;
  const uint8_t* t = target + (int64_t)m * hw;
  unsigned c[4] = {0, 0, 0, 0};
  auto take = [&](float v, unsigned g) {
    g = g ? 1u : 0u;
    c[(v >= cut_pred ? 2u : 0u) + g] += 1;
    // upper bound over the ascending cuts: b = #{i < nth : cut_i <= v} (NaN compares false everywhere -> 0,
    // as (NaN >= th) is False in the reference); cut_s is +inf from nth on, 256 entries -> 8 steps
    int b = 0;
#pragma unroll
    for (int step = (AVS_MAX_TH + 1) / 2; step > 0; step >>= 1)
      if (v >= cut_s[b + step - 1]) b += step;
    atomicAdd(&h_s[b * 2 + g], 1u);
  };
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if ((hw & 3) == 0 && (((uintptr_t)x | (uintptr_t)t) & 15) == 0) {
    for (int64_t i = i0; i < hw / 4; i += stride) {
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      const uint32_t g = reinterpret_cast<const uint32_t*>(t)[i];
      take(v.x, g & 255u); take(v.y, (g >> 8) & 255u); take(v.z, (g >> 16) & 255u); take(v.w, g >> 24);
    }
  } else {
    for (int64_t i = i0; i < hw; i += stride) take(x[i], t[i]);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned v = c[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&c_s[k], v);
  }
  __syncthreads();
  // integer atomics: the totals do not depend on the order of arrival
  if (threadIdx.x < 4 && c_s[threadIdx.x]) atomicAdd(&conf[(int64_t)m * 4 + threadIdx.x], (unsigned long long)c_s[threadIdx.x]);
  for (int i = threadIdx.x; i < (nth + 1) * 2; i += 256)
    if (h_s[i]) atomicAdd(&hist[(int64_t)m * (nth + 1) * 2 + i], (unsigned long long)h_s[i]);
}
void launch_avs_counts(const float* logits, const uint8_t* target, int n, int64_t hw, const float* cuts, int nth,
                       float cut_pred, int64_t* conf, int64_t* hist, hipStream_t s) {
  if (nth < 1 || nth > AVS_MAX_TH) throw std::runtime_error("avs_counts: 1 <= number of thresholds <= 255");
  if (n <= 0) return;
  HIP_TRY(hipMemsetAsync(conf, 0, (size_t)n * 4 * sizeof(int64_t), s));
  HIP_TRY(hipMemsetAsync(hist, 0, (size_t)n * (nth + 1) * 2 * sizeof(int64_t), s));
  if (hw <= 0) return;
  int64_t blocks = cdiv64(hw, 256 * 32);
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(avs_counts_kernel, dim3((unsigned)blocks, n), dim3(256), 0, s, logits, target, hw, cuts, nth,
                     cut_pred, reinterpret_cast<unsigned long long*>(conf), reinterpret_cast<unsigned long long*>(hist));
}

// f-1: `sam_preprocess` (utils/refer_seg.py:560-570) on the resized uint8 HWC image: (x - mean) / std per
// channel in f32 (IEEE division, bit-identical to the torch expression), CHW output zero-padded to S x S.
__global__ __launch_bounds__(256) void sam_preprocess_kernel(const uint8_t* __restrict__ img, int h, int w, int S,
                                                             float m0, float m1, float m2, float s0, float s1,
                                                             float s2, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)S * S) return;
  const int y = (int)(i / S), x = (int)(i % S);
  float r = 0.f, g = 0.f, b = 0.f;  // F.pad pads the NORMALISED image with zeros
  if (y < h && x < w) {
    const uint8_t* p = img + ((int64_t)y * w + x) * 3;
    r = ((float)p[0] - m0) / s0;
    g = ((float)p[1] - m1) / s1;
    b = ((float)p[2] - m2) / s2;
  }
  out[i] = r;
  out[(int64_t)S * S + i] = g;
  out[2 * (int64_t)S * S + i] = b;
}
void launch_sam_preprocess(const uint8_t* img, int h, int w, int S, const float* mean, const float* std_, float* out,
                           hipStream_t s) {
  if (h > S || w > S || h <= 0 || w <= 0) throw std::runtime_error("sam_preprocess: image larger than the SAM input");
  hipLaunchKernelGGL(sam_preprocess_kernel, dim3((unsigned)cdiv64((int64_t)S * S, 256)), dim3(256), 0, s, img, h, w, S,
                     mean[0], mean[1], mean[2], std_[0], std_[1], std_[2], out);
}

// f-1: `ResizeLongestSide.apply_image` (segment_anything/utils/transforms.py:27-34) = torchvision `resize` of a PIL
// image = Pillow `Image.resize(..., BILINEAR)`, and the bicubic shortest-edge resize inside `CLIPImageProcessor`
// (utils/refer_seg.py:578-580).  Pillow (un-vendored dependency; algorithm of src/libImaging/Resample.c, 12.2.0
// here) resamples 8-bit images separably in FIXED POINT: per output index a window [xmin, xmin + n) of input pixels
// and int32 coefficients round(k * 2^22) (normalised filter taps; the support widens with the downscale factor =
// antialiasing); out = clip8((2^21 + sum pixel * coeff) >> 22); horizontal pass first, its uint8 result feeds the
// vertical pass.  The (tiny) coefficient tables are built on the host in double exactly as precompute_coeffs /
// normalize_coeffs_8bpc do (anyref_amd/preprocess.py); the two integer passes below are bit-exact.
__global__ __launch_bounds__(256) void resample_h_u8_kernel(const uint8_t* __restrict__ in, int H, int W, int C,
                                                            uint8_t* __restrict__ out, int ow,
                                                            const int* __restrict__ bounds, const int* __restrict__ kk,
                                                            int ksize) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)H * ow * C) return;
  const int c = (int)(i % C), xx = (int)((i / C) % ow);
  const int64_t y = i / ((int64_t)C * ow);
  const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
  const int* k = kk + (int64_t)xx * ksize;
  const uint8_t* row = in + (y * W + xmin) * C + c;
  int ss = 1 << 21;
  for (int x = 0; x < n; ++x) ss += (int)row[(int64_t)x * C] * k[x];
  ss >>= 22;
  out[i] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
}
__global__ __launch_bounds__(256) void resample_v_u8_kernel(const uint8_t* __restrict__ in, int H, int64_t WC,
                                                            uint8_t* __restrict__ out, int oh,
                                                            const int* __restrict__ bounds, const int* __restrict__ kk,
                                                            int ksize) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)oh * WC) return;
  const int64_t col = i % WC;
  const int yy = (int)(i / WC);
  const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
  const int* k = kk + (int64_t)yy * ksize;
  const uint8_t* p = in + (int64_t)ymin * WC + col;
  int ss = 1 << 21;
  for (int y = 0; y < n; ++y) ss += (int)p[(int64_t)y * WC] * k[y];
  ss >>= 22;
  out[i] = (uint8_t)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
}
void launch_pil_resample_u8(const uint8_t* in, int H, int W, int C, uint8_t* tmp, uint8_t* out, int ow, int oh,
                            const int* xbounds, const int* xk, int kx, const int* ybounds, const int* yk, int ky,
                            hipStream_t s) {
  if (H <= 0 || W <= 0 || C <= 0 || ow <= 0 || oh <= 0) throw std::runtime_error("pil_resample: empty image");
  // Pillow skips a pass whose size does not change (ImagingResample: need_horizontal / need_vertical)
  const bool need_h = ow != W, need_v = oh != H;
  const uint8_t* src = in;
  if (need_h) {
    if (!xbounds || !xk || kx <= 0) throw std::runtime_error("pil_resample: horizontal pass needs its coefficient table");
    uint8_t* dst = need_v ? tmp : out;
    if (!dst) throw std::runtime_error("pil_resample: two passes need the [H, ow, C] scratch image");
    hipLaunchKernelGGL(resample_h_u8_kernel, dim3((unsigned)cdiv64((int64_t)H * ow * C, 256)), dim3(256), 0, s, src, H, W,
                       C, dst, ow, xbounds, xk, kx);
    src = dst;
  }
  if (need_v) {
    if (!ybounds || !yk || ky <= 0) throw std::runtime_error("pil_resample: vertical pass needs its coefficient table");
    hipLaunchKernelGGL(resample_v_u8_kernel, dim3((unsigned)cdiv64((int64_t)oh * ow * C, 256)), dim3(256), 0, s, src, H,
                       (int64_t)ow * C, out, oh, ybounds, yk, ky);
  } else if (!need_h) {
    HIP_TRY(hipMemcpyAsync(out, in, (size_t)H * W * C, hipMemcpyDeviceToDevice, s));
  }
}

// f-1: the rest of the CLIP input path (utils/refer_seg.py:578-587) on the resized uint8 HWC image: rescale by
// 1/255 in DOUBLE then to f32, (x - mean) / std in f32 -- the image processor's arithmetic, bit for bit -- and
// `F.interpolate(size=(S, S), mode="bilinear", align_corners=False)` (PyTorch upsample_bilinear2d, bil_idx above).
// Rows [y0, y0 + h) x columns [x0, x0 + w) of the [ih, iw, 3] image are the source (the centre crop when enabled).
__device__ inline float clip_px(const uint8_t* __restrict__ img, int iw, int y, int x, int c, float m, float sd) {
  const float v = (float)((double)img[((int64_t)y * iw + x) * 3 + c] * (1.0 / 255.0));
  return (v - m) / sd;
}
__global__ __launch_bounds__(256) void clip_finish_kernel(const uint8_t* __restrict__ img, int iw, int y0, int x0,
                                                          int h, int w, int S, float m0, float m1, float m2, float s0,
                                                          float s1, float s2, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)3 * S * S) return;
  const int c = (int)(i / ((int64_t)S * S)), Y = (int)((i / S) % S), X = (int)(i % S);
  const float m = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
  if (h == S && w == S) {  // interpolate to the same size is the identity in PyTorch as well
    out[i] = clip_px(img, iw, y0 + Y, x0 + X, c, m, sd);
    return;
  }
  int ya, yb, xa, xb;
  float ly, lx;
  bil_idx(Y, (float)h / (float)S, h, ya, yb, ly);
  bil_idx(X, (float)w / (float)S, w, xa, xb, lx);
  const float a = clip_px(img, iw, y0 + ya, x0 + xa, c, m, sd), b = clip_px(img, iw, y0 + ya, x0 + xb, c, m, sd);
  const float cc = clip_px(img, iw, y0 + yb, x0 + xa, c, m, sd), d = clip_px(img, iw, y0 + yb, x0 + xb, c, m, sd);
  out[i] = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * cc + lx * d);
}
void launch_clip_finish(const uint8_t* img, int ih, int iw, int y0, int x0, int h, int w, int S, const float* mean,
                        const float* std_, float* out, hipStream_t s) {
  if (h <= 0 || w <= 0 || y0 < 0 || x0 < 0 || y0 + h > ih || x0 + w > iw || S <= 0)
    throw std::runtime_error("clip_finish: source window outside the image");
  hipLaunchKernelGGL(clip_finish_kernel, dim3((unsigned)cdiv64((int64_t)3 * S * S, 256)), dim3(256), 0, s, img, iw, y0, x0,
                     h, w, S, mean[0], mean[1], mean[2], std_[0], std_[1], std_[2], out);
}

// f-3: reference-image tokens (anyref.py:335-338, :697-700): [n, L, H] CLIP features -> mean over groups of 16
// consecutive tokens -> [n, L/16, H] -> (if L/16 != n_out) mean over groups of n_out consecutive rows -> [n, n_out, H].
// Two rounded means in the reference's order (not one mean over 64 tokens).
__global__ __launch_bounds__(256) void pool_ref_tokens_kernel(const float* __restrict__ f, int L, int H, int n_out,
                                                              float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int n16 = L / 16;
  const bool second = n16 != n_out;
  if (i >= (int64_t)n_out * H) return;
  const int c = (int)(i % H), j = (int)(i / H);
  const float* base = f + (int64_t)blockIdx.y * L * H + c;
  const int per = second ? n_out : 1;                 // 16-token means per output row
  float acc2 = 0.f;
  for (int g = 0; g < per; ++g) {
    const float* p = base + (int64_t)(j * per + g) * 16 * H;
    float acc = 0.f;
    for (int t = 0; t < 16; ++t) acc += p[(int64_t)t * H];
    acc2 += acc / 16.f;
  }
  out[(int64_t)blockIdx.y * n_out * H + i] = second ? acc2 / (float)per : acc2;
}
void launch_pool_ref_tokens(const float* f, int n, int L, int H, int n_out, float* out, hipStream_t s) {
  if (n <= 0) return;
  if (L % 16 || (L / 16 != n_out && L / 16 != n_out * n_out))
    throw std::runtime_error("pool_ref_tokens: token count must be 16 * n_out or 16 * n_out^2 (anyref.py:335-338)");
  hipLaunchKernelGGL(pool_ref_tokens_kernel, dim3((unsigned)cdiv64((int64_t)n_out * H, 256), n), dim3(256), 0, s, f, L, H,
                     n_out, out);
}

// rows[i] of dst <- bias (SAM window layers: q/k/v of a zero-padded token is exactly the bias)
template <typename T>
__global__ __launch_bounds__(256) void fill_rows_bias_kernel(T* __restrict__ dst, int ld, const int* __restrict__ rows,
                                                             const float* __restrict__ bias, int N) {
  T* d = dst + (int64_t)rows[blockIdx.x] * ld;
  for (int n = threadIdx.x; n < N; n += 256) d[n] = from_f32<T>(bias ? bias[n] : 0.f);
}
template <typename T>
void launch_fill_rows_bias(void* dst, int ld, const int* rows, int nrows, const float* bias, int N, hipStream_t s) {
  if (nrows <= 0) return;
  hipLaunchKernelGGL((fill_rows_bias_kernel<T>), dim3(nrows), dim3(256), 0, s, reinterpret_cast<T*>(dst), ld, rows, bias, N);
}
template void launch_fill_rows_bias<float>(void*, int, const int*, int, const float*, int, hipStream_t);
template void launch_fill_rows_bias<bf16>(void*, int, const int*, int, const float*, int, hipStream_t);
template void launch_fill_rows_bias<f16>(void*, int, const int*, int, const float*, int, hipStream_t);

// ---------------------------------------------------------------------------------------------
// fp8 weight-only quantisation (BASELINE config 5: 13B LLM, fp8 weights).  One workgroup per row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(const float* __restrict__ src, int lds, int K,
                                                             uint8_t* __restrict__ q, int ldq,
                                                             float* __restrict__ scale, int sstride) {
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* row = src + (int64_t)n * lds;
  float amax = 0.f;
  for (int k = tid; k < K; k += 256) amax = fmaxf(amax, fabsf(row[k]));
  amax = wave_max(amax);
  __shared__ float red[4];
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc = amax > 0.f ? amax / FP8_E4M3_MAX : 1.f;
  if (tid == 0) scale[(int64_t)n * sstride] = sc;
  uint8_t* out = q + (int64_t)n * ldq;
  for (int k = tid * 2; k < K; k += 512) {  // two values per v_cvt_pk_fp8_f32 (RNE, saturating)
    const float a = row[k] / sc, b = k + 1 < K ? row[k + 1] / sc : 0.f;
    const int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    out[k] = (uint8_t)(w & 255);
    if (k + 1 < K) out[k + 1] = (uint8_t)((w >> 8) & 255);
  }
}
void launch_quant_fp8_rows(const float* src, int lds, int N, int K, uint8_t* q, int ldq, float* scale, hipStream_t s,
                           int scale_stride) {
  if (N <= 0) return;
  hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3(N), dim3(256), 0, s, src, lds, K, q, ldq, scale, scale_stride);
}

__global__ __launch_bounds__(256) void dequant_fp8_rows_kernel(const uint8_t* __restrict__ q, int ldq,
                                                               const float* __restrict__ scale, int N, int K,
                                                               bf16* __restrict__ out, int ldo) {
  // 16 values per thread step: one 16-byte load, two 16-byte stores
  const int64_t per_row = K / 16;
  const int64_t total = (int64_t)N * per_row;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / per_row), c = (int)(i % per_row);
    const uint4v v = *reinterpret_cast<const uint4v*>(q + (int64_t)n * ldq + c * 16);
    float f[16];
    unpack_fp8x16(v, f);
    const float sc = scale[n];
    uint32_t w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = (uint32_t)f2bf(f[2 * j] * sc).x | ((uint32_t)f2bf(f[2 * j + 1] * sc).x << 16);
    uint4v* o = reinterpret_cast<uint4v*>(out + (int64_t)n * ldo + c * 16);
    o[0] = uint4v{w[0], w[1], w[2], w[3]};
    o[1] = uint4v{w[4], w[5], w[6], w[7]};
  }
}
void launch_dequant_fp8_rows(const uint8_t* q, int ldq, const float* scale, int N, int K, void* out_bf16, int ldo,
                             hipStream_t s) {
  if (N <= 0) return;
  if (K % 16 || ldq % 16 || ldo % 8) throw std::runtime_error("dequant_fp8: K must be a multiple of 16");
  int64_t blocks = cdiv64((int64_t)N * (K / 16), 256);
  blocks = blocks > 4096 ? 4096 : blocks;
  hipLaunchKernelGGL(dequant_fp8_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, q, ldq, scale, N, K,
                     reinterpret_cast<bf16*>(out_bf16), ldo);
}

// ---------------------------------------------------------------------------------------------
// Audio front-end: Kaldi-compatible log-mel filterbank of one waveform clip, as
// torchaudio.compliance.kaldi.fbank computes it for the options of model/ImageBind/data.py:28-64 (see
// oracle/preprocess_oracle.py::kaldi_fbank for the restatement and what pins it), with the clip-mean removal in front
// and the pad / cut to `target_len` frames + Normalize(mean, std) behind it fused in.
// One workgroup per output frame column: 400 samples -> DC removal -> pre-emphasis -> Hann window -> zero pad to 512 ->
// DFT by direct summation in f64 against a 512-entry twiddle table (198 frames x 257 bins x 400 taps: microseconds;
// an FFT would buy nothing) -> power -> 128 triangular mel filters (f64 sums) -> log(max(., eps)) -> (x - mean) / std.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wave_sum_kernel(const float* __restrict__ w, int64_t n, double* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)w[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

template <int WIN, int PAD>
__global__ __launch_bounds__(256) void kaldi_fbank_kernel(const float* __restrict__ wave, int T, int64_t n_all,
                                                          const double* __restrict__ wsum, int shift, float preemph,
                                                          const float* __restrict__ banks, int n_mel,
                                                          const double* __restrict__ tw, int n_frames, int target_len,
                                                          float mean, float stdv, float* __restrict__ out) {
  constexpr int NB = PAD / 2 + 1;
  __shared__ double tc[PAD], ts[PAD];
  __shared__ double y[WIN];
  __shared__ float pw[NB];
  __shared__ double red[4];
  const int f = blockIdx.x, tid = threadIdx.x;
  if (f >= n_frames) {  // zero-padded frames: Normalize(0)
    for (int b = tid; b < n_mel; b += 256) out[(int64_t)b * target_len + f] = (0.f - mean) / stdv;
    return;
  }
  for (int i = tid; i < PAD; i += 256) {
    tc[i] = tw[i];
    ts[i] = tw[PAD + i];
  }
  const float gmean = (float)(wsum[0] / (double)n_all);  // waveform -= waveform.mean() (data.py:30), f32 like torch
  // frame, DC removal (f64 mean of the f32 samples)
  double part = 0.0;
  for (int i = tid; i < WIN; i += 256) {
    const double v = (double)(wave[(int64_t)f * shift + i] - gmean);
    y[i] = v;
    part += v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  const double fmean = (red[0] + red[1] + red[2] + red[3]) / (double)WIN;
  // pre-emphasis on the DC-free frame (replicate pad: y[-1] = y[0]) and the Hann window, into registers first
  double v0 = 0.0, v1 = 0.0;
  {
    const double PI2 = 6.283185307179586476925286766559;
    const int i0 = tid, i1 = tid + 256;
    if (i0 < WIN) {
      const double cur = y[i0] - fmean, prev = y[i0 > 0 ? i0 - 1 : 0] - fmean;
      v0 = (cur - (double)preemph * prev) * (0.5 - 0.5 * cos(PI2 * (double)i0 / (double)(WIN - 1)));
    }
    if (i1 < WIN) {
      const double cur = y[i1] - fmean, prev = y[i1 - 1] - fmean;
      v1 = (cur - (double)preemph * prev) * (0.5 - 0.5 * cos(PI2 * (double)i1 / (double)(WIN - 1)));
    }
  }
  __syncthreads();
  if (tid < WIN) y[tid] = v0;
  if (tid + 256 < WIN) y[tid + 256] = v1;
  __syncthreads();
  // power spectrum: bin k = tid (and bin 256 on thread 0's second pass)
  for (int k = tid; k < NB; k += 256) {
    double re = 0.0, im = 0.0;
    int idx = 0;  // (k * n) mod PAD
    for (int n = 0; n < WIN; ++n) {
      re += y[n] * tc[idx];
      im -= y[n] * ts[idx];
      idx = (idx + k) & (PAD - 1);
    }
    pw[k] = (float)(re * re + im * im);
  }
  __syncthreads();
  for (int b = tid; b < n_mel; b += 256) {
    const float* row = banks + (int64_t)b * NB;
    double e = 0.0;
    for (int k = 0; k < NB; ++k) e += (double)pw[k] * (double)row[k];
    const float le = logf(fmaxf((float)e, 1.1920928955078125e-07f));
    out[(int64_t)b * target_len + f] = (le - mean) / stdv;
  }
}

void launch_kaldi_fbank(const float* wave, int C, int T, int win, int shift, int padded, float preemph, const float* banks,
                        int n_mel, const double* tw, double* scratch, int target_len, float mean, float stdv, float* out,
                        hipStream_t s) {
  if (win != 400 || padded != 512) throw std::runtime_error("kaldi_fbank: built for 25 ms windows at 16 kHz (400 -> 512)");
  if (C < 1 || T < 0 || n_mel < 1 || target_len < 1 || shift < 1) throw std::runtime_error("kaldi_fbank: bad sizes");
  int n_frames = T < win ? 0 : 1 + (T - win) / shift;  // snip_edges
  if (n_frames > target_len) n_frames = target_len;     // data.py:57-58 cuts
  hipLaunchKernelGGL(wave_sum_kernel, dim3(1), dim3(256), 0, s, wave, (int64_t)C * T, scratch);
  hipLaunchKernelGGL((kaldi_fbank_kernel<400, 512>), dim3(target_len), dim3(256), 0, s, wave, T, (int64_t)C * T, scratch,
                     shift, preemph, banks, n_mel, tw, n_frames, target_len, mean, stdv, out);
}

}  // namespace anyref

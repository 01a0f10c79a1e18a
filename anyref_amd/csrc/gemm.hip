// Tiled MFMA GEMM for gfx950, f32 (parity) and bf16 (perf) instantiations + the split-K workspace pool.
// The templates live in gemm_impl.h; gemm_f16.hip instantiates the f16 flavour (SAM encoder) so that the two compile
// side by side.
#include "gemm_impl.h"

namespace anyref {

// per-stream slab workspace (grown on demand; streams never share one); defined by gemm.hip, shared by gemm_f16.hip
namespace {
struct SplitKWs { hipStream_t s; float* p; size_t cap; };
// process-wide, keyed by the stream handle (unique while the stream lives): a handle destroyed on another thread
// (Python GC) still finds and frees the workspaces of the streams it owns.  Entries are only touched under the lock.
std::mutex& splitk_pool_mu() {
  static std::mutex mu;
  return mu;
}
std::vector<SplitKWs>& splitk_pool() {
  static std::vector<SplitKWs> pool;
  return pool;
}
}  // namespace
float* splitk_workspace(hipStream_t s, size_t bytes) {
  using Ws = SplitKWs;
  std::lock_guard<std::mutex> lock(splitk_pool_mu());
  auto& pool = splitk_pool();
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

// make sure `s` owns a split-K workspace of at least `bytes` (call before capturing `s` into a graph:
// growing the workspace synchronises the stream)
void gemm_reserve_workspace(hipStream_t s, size_t bytes) { (void)splitk_workspace(s, bytes); }
void gemm_release_workspace(hipStream_t s) {
  std::lock_guard<std::mutex> lock(splitk_pool_mu());
  auto& pool = splitk_pool();
  for (size_t i = 0; i < pool.size(); ++i)
    if (pool[i].s == s) {
      (void)hipFree(pool[i].p);
      pool.erase(pool.begin() + i);
      return;
    }
}

template void launch_gemm<float>(const GemmArgs&, hipStream_t);
template void launch_gemm<bf16>(const GemmArgs&, hipStream_t);

}  // namespace anyref

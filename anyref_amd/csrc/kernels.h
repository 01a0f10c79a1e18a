// Launcher declarations for the gfx950 kernels.  Every launcher is templated on the
// activation/weight storage type T: `float` (parity mode, exact-f32 MFMA 16x16x4) or
// `bf16` (perf mode, MFMA 16x16x32 bf16).  Accumulation and the residual stream are fp32 in both.
#pragma once
#include <cstdlib>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "common.h"

namespace anyref {

// ---- optional per-kernel timing (bench.py's roofline leg) ---------------------------------------
// When a Profiler is installed for the calling thread, the GEMM / GEMV / attention launchers
// bracket every launch with a hipEvent pair on the launch stream and book the elapsed time, the
// algorithmic FLOPs and the algorithmic bytes under the kernel's tag.
struct ProfStat {
  double ms = 0;
  int64_t count = 0;
  double flops = 0, bytes = 0;
};
class Profiler {
 public:
  ~Profiler();
  // returns false when this launch is filtered out / not sampled (then end() must not be called)
  bool begin(const char* tag, double flops, double bytes, hipStream_t s);
  void end(hipStream_t s);
  void collect();  // after the stream has been synchronised
  void reset();
  // A bracket reads more than the kernel inside it (event packets are not free); that overhead is
  // measured here with a tiny kernel (ops.hip; synchronises the stream) and taken off every bracket
  // in collect(), so the booked time is the kernel's own duration.
  void calibrate(hipStream_t s);
  double null_ms = 0.0;
  bool on = false;
  std::string filter;    // non-empty: only tags starting with this are timed
  int sample_every = 1;  // time every n-th launch of a tag (keeps the event overhead out of the timed region)
  std::vector<std::pair<std::string, ProfStat>> stats() const;

 private:
  struct Rec {
    int tag;
    hipEvent_t a, b;
    double flops, bytes;
  };
  std::vector<Rec> recs_;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool_;
  size_t used_ = 0;
  std::vector<std::string> tags_;
  std::vector<ProfStat> stats_;
  std::vector<int64_t> seen_;
};
extern thread_local Profiler* g_prof;

// ---- kernel-side timestamps (bench.py's in-situ roofline) -----------------------------------------
// hipEvent brackets cannot be placed inside a replayed hipGraph and rocprofv3 serialises the two streams of a call,
// so neither sees a decode GEMV as it runs in production (graph replay, SAM encoder co-running).  With stamps on,
// a stamping kernel records the 100 MHz wall clock itself: every workgroup writes {earliest wave start, latest wave
// end} into its own 16-byte slot (plain stores, no global atomics); the host reduces a launch's slots to min / max.
// A launch inside a captured step cannot get a fresh slot per replay (its arguments are baked in), so slots are
// addressed as base + *epoch * stride: the captured step ends with a one-thread kernel that bumps *epoch.
struct StampArgs {
  unsigned long long* base = nullptr;  // null: the kernel does not stamp
  const int* epoch = nullptr;          // device int (a constant 0 for eager launches)
  unsigned stride = 0;                 // u64 elements per epoch
  int max_epoch = 0;                   // replays past this are not recorded
};
struct StampRow {
  std::string tag;
  double bytes = 0;
  double t0_us = 0, t1_us = 0;  // relative to the first stamp collected
  int epoch = -1;               // replay index of the captured step, -1: eager launch
  // spread over the launch's workgroups: last start - first start, last end - first end (the tail), and the median
  // workgroup's own span
  double start_spread_us = 0, end_spread_us = 0, wg_median_us = 0;
};
class Stamper {
 public:
  ~Stamper();
  void enable(bool on);  // allocates / frees the device buffers
  bool on = false;
  // next slot for a launch of `grid` workgroups (called by a stamping launcher)
  StampArgs slot(const char* tag, double bytes, int grid);
  void graph_begin(int key);          // the launches up to graph_end() are being captured under `key`
  void graph_end(hipStream_t cap);    // appends the epoch bump to the captured stream
  void graph_abort();                 // the capture under way failed: forget its launches, back to eager bookkeeping
  // launches that could not be stamped (record full) + graph replays past the epoch capacity since enable / the last collect
  int64_t dropped() const { return dropped_; }
  bool graph_known(int key) const { return graphs_.count(key) != 0; }
  void graph_replayed(int key) {
    if ((int)epoch_keys_.size() < max_epoch_) epoch_keys_.push_back(key);
    else ++dropped_;  // (the device side stops recording at max_epoch too)
  }
  std::vector<StampRow> collect();    // after the device has been synchronised; also resets the counters
 private:
  struct Rec { std::string tag; double bytes; int grid; size_t off; };
  unsigned long long* buf_ = nullptr;
  int* ctl_ = nullptr;  // [0] = epoch, [1] = constant 0
  size_t eager_cap_ = 0, eager_used_ = 0, graph_off_ = 0, graph_stride_ = 0;
  int max_epoch_ = 0;
  int capturing_ = -1;
  size_t cap_used_ = 0;
  int64_t dropped_ = 0;
  std::vector<Rec> eager_;
  std::map<int, std::vector<Rec>> graphs_;
  std::vector<int> epoch_keys_;
};
extern thread_local Stamper* g_stamp;
struct ProfScope {  // RAII bracket used inside the launchers
  hipStream_t s;
  bool active;
  ProfScope(const char* tag, double flops, double bytes, hipStream_t st) : s(st), active(g_prof && g_prof->on) {
    if (active) active = g_prof->begin(tag, flops, bytes, s);
  }
  ~ProfScope() {
    if (active) g_prof->end(s);
  }
};

// C[dst(m), n] = act(alpha * sum_k A[m,k] W[n,k] + bias[n]) (+ resid[dst(m), n])
struct GemmArgs {
  const void* A = nullptr;  // T [M,K], row stride lda
  const void* W = nullptr;  // T [N,K], row stride ldw  (nn.Linear layout)
  const float* bias = nullptr;
  void* C = nullptr;  // T or f32 [*, N], row stride ldc
  const float* resid = nullptr;  // f32, row stride ldr, indexed by destination row; may alias C
  const int* row_map = nullptr;  // destination row of source row m, or <0 to drop the row
  // source row of A for logical row m (16-bit LDS-DMA kernel only): the SAM window layers' proj reads the attention
  // output in window layout but multiplies the REAL tokens only, in token order (4096 of 4900 rows per image)
  const int* a_row_map = nullptr;
  int M = 0, N = 0, K = 0;
  int lda = 0, ldw = 0, ldc = 0, ldr = 0;
  int act = ACT_NONE;
  int c_f32 = 0;  // 1: C is f32, 0: C is T
  float alpha = 1.f;
  int order = 3;   // bit 0: XCD-chunked block remap, bit 1: M-fastest tile order
  // SwiGLU fused into the epilogue: W holds gate / up rows INTERLEAVED (row 2j = gate_j, 2j+1 = up_j), the
  // output is [M, N/2]: C[m, j] = silu(c[m, 2j]) * c[m, 2j+1]  (no activation / residual / row map with it)
  int swiglu_pairs = 0;
  // fp8 weight operand (bf16 LDS-DMA kernel only): W holds e4m3 bytes [N, K] (ldw, sW in bytes), the finished
  // column n is multiplied by col_scale[n]
  int w_fp8 = 0;
  const float* col_scale = nullptr;
  int group_m = 0;  // > 0 (set by the launcher): grouped tile order, this many tile rows per group
  // > 0: at most this many workgroups (bf16 LDS-DMA kernel, M >= 128-row tiles, batch 1); each walks several tiles
  int max_wg = 0;
  int vec_ok = 0;  // set by the launcher: N / strides / bases allow 4-wide vector epilogue accesses
  // optional: RMSNorm of the finished output rows fused into the split-K reduction (prefill o_proj /
  // down_proj -> the norm that follows).  Honoured only on the split-K path; *norm_done says whether it was.
  const float* norm_gain = nullptr;
  void* norm_out = nullptr;  // T [M, N], row stride norm_ld
  int norm_ld = 0;
  float norm_eps = 0.f;
  const float* norm_bias = nullptr;  // set: LayerNorm (mean removed, + bias) instead of RMSNorm
  // raw split-K: `slabs` K slices summed by the CONSUMER -- C is ignored, slice z goes to slabs_out + z * M * N (f32,
  // row stride N).  No bias / activation / residual.  (prefill qkv: the RoPE + cache kernel adds the two slices.)
  float* slabs_out = nullptr;
  int slabs = 0;
  bool* norm_done = nullptr;
  // optional batching over blockIdx.z (element strides)
  int batch = 1;
  int64_t sA = 0, sW = 0, sC = 0, sR = 0, sBias = 0;
};
template <typename T>
void launch_gemm(const GemmArgs& a, hipStream_t s);
void gemm_reserve_workspace(hipStream_t s, size_t bytes);
// free the split-K slab / attention partial workspaces pooled for a stream that is about to be destroyed
void gemm_release_workspace(hipStream_t s);
void attn_release_workspace(hipStream_t s);

// Small-M weight-streaming GEMV for the decode step (HBM-bound):
// y[b, n] = act(sum_k xn[b,k] W[n,k]) (+ resid) where xn = rmsnorm(x) * gain if gain != null.
struct GemvArgs {
  const float* x = nullptr;  // f32 [B,K] (residual stream / activations), row stride ldx
  const float* gain = nullptr;  // RMSNorm weight [K] or null (no norm)
  float eps = 1e-6f;
  const void* W = nullptr;   // T [N,K]
  const void* W2 = nullptr;  // T [N,K] or null; if set: y = silu(x.W) * (x.W2)  (SwiGLU)
  const float* bias = nullptr;
  float* y = nullptr;  // f32 [B,N], row stride ldy
  const float* resid = nullptr;  // f32 [B,N] row stride ldy, may alias y
  int B = 1, N = 0, K = 0, ldx = 0, ldy = 0;
  int act = ACT_NONE;
  int ldw = 0;  // row stride of W / W2 in elements (0: K); 2K for the interleaved gate/up matrix
  // fp8 weight-only mode (T = bf16 activations): W / W2 are e4m3 bytes [N,K], y = (sum_k x q) * wscale[n * ws_stride]
  int w_fp8 = 0;
  const float* wscale = nullptr;
  const float* wscale2 = nullptr;
  int ws_stride = 1;
  // optional f32 copy of the normalised input rows (needs `gain`): xn_out[(xn_row_map ? xn_row_map[b] : b) * xn_ld + k]
  float* xn_out = nullptr;
  const int* xn_row_map = nullptr;
  int xn_ld = 0;
  StampArgs stamp;  // filled by the launcher when kernel-side timestamps are on
  int grid = 0;     // > 0: this many workgroups instead of the launcher's rule (decode steps beside the co-running encoder)
};
template <typename T>
void launch_gemv(const GemvArgs& a, hipStream_t s);
// f32 weights, 1..8 activation rows, K <= 4096: the mask decoder's token GEMMs and the [SEG] hand-off MLP
void launch_gemv_skinny_f32(const GemvArgs& a, hipStream_t s);
// Row-wise fp8 (e4m3fn) weight quantisation: scale[n] = max|W[n,:]| / 448 (1 if the row is zero),
// q[n,k] = RNE_e4m3(W[n,k] / scale[n]); src f32 [N, K] (row stride lds), q [N, K] bytes (row stride ldq)
void launch_quant_fp8_rows(const float* src, int lds, int N, int K, uint8_t* q, int ldq, float* scale, hipStream_t s,
                           int scale_stride = 1);
// q * scale -> bf16 [N, K] (prefill GEMM operand)
void launch_dequant_fp8_rows(const uint8_t* q, int ldq, const float* scale, int N, int K, void* out_bf16, int ldo,
                             hipStream_t s);

// y[dst(m)] = LN(x[m]) * g + b   (rms: y = x * rsqrt(mean x^2 + eps) * g)
struct NormArgs {
  const float* x = nullptr;  // f32 [M,D], row stride ldx
  const float* gain = nullptr;
  const float* bias = nullptr;  // null for RMSNorm
  void* y = nullptr;            // T or f32, row stride ldy
  const int* row_map = nullptr;
  int M = 0, D = 0, ldx = 0, ldy = 0;
  float eps = 1e-5f;
  int rms = 0;
  int y_f32 = 0;
  int act = ACT_NONE;  // applied after the affine (LayerNorm2d -> GELU in the upscaler)
  // optional side job of the wide-row kernel (SAM window layers: the q/k/v rows of a window's zero-padded tokens are
  // exactly the qkv bias, image_encoder.py:175-179): fill_n extra workgroups write fill_dst[fill_rows[i], 0:fill_N) =
  // T(fill_bias) -- one launch less per layer (6.4 us each).  Honoured by norm_wide_kernel only (*fill_done says so).
  void* fill_dst = nullptr;
  int fill_ld = 0, fill_n = 0, fill_N = 0;
  const int* fill_rows = nullptr;
  const float* fill_bias = nullptr;
  bool* fill_done = nullptr;
};
template <typename T>
void launch_norm(const NormArgs& a, hipStream_t s);

// Fused multi-head attention, online softmax.  Element strides; Q/K/V/O are T.
// score(i,j) = scale * q_i.k_j + rel_h[i, j / kw] + rel_w[i, j % kw]; key j visible iff
// j < kv_len[b] and (!causal or j <= q_pos0[b] + i).
struct AttnArgs {
  const void *Q = nullptr, *K = nullptr, *V = nullptr;
  void* O = nullptr;
  int64_t q_bs = 0, q_rs = 0, q_hs = 0;  // batch / row / head strides
  int64_t k_bs = 0, k_rs = 0, k_hs = 0;
  int64_t v_bs = 0, v_rs = 0, v_hs = 0;
  int64_t o_bs = 0, o_rs = 0, o_hs = 0;
  int B = 0, H = 0, Sq = 0, Sk = 0, hd = 0;
  float scale = 1.f;
  int causal = 0;
  const int* q_pos0 = nullptr;  // [B] position of query row 0 (causal with a KV cache); null = 0
  const int* kv_len = nullptr;  // [B] or null (= Sk)
  const int* q_len = nullptr;   // [B] or null (= Sq): rows >= q_len are skipped
  const float* rel_h = nullptr;  // f32 [B,H,Sq,kh] or null
  const float* rel_w = nullptr;  // f32 [B,H,Sq,kw]
  int kh = 0, kw = 0;
  // alternative bias source (SAM encoder hot path): P[h][b*Sq+i][0:Np) = q_i . tab_h[j], [Np:2Np) = q_i . tab_w[j]
  // produced by one batched MFMA GEMM; the kernel applies the relative shift itself:
  // rel_h[i,ky] = P[..][y_i - ky + kh - 1], rel_w[i,kx] = P[..][Np + x_i - kx + kw - 1]   (get_rel_pos)
  const float* rel_p = nullptr;
  int64_t rel_hs = 0;  // head stride of P (elements)
  int rel_ld = 0;      // row stride of P = 2*Np
  // or (3), SAM windows in the resident-key form only: the tables themselves, T [2*kh-1][hd] / [2*kw-1][hd] at row
  // stride rel_tab_ld (2*k-1 <= 32); the kernel computes q . R^T with 12 MFMAs per wave from the query fragments it
  // already holds and scatters it to the shifted layout -- no P buffer, no GEMM launch
  const void* rel_tab_h = nullptr;
  const void* rel_tab_w = nullptr;
  int rel_tab_ld = 0;
  int o_f32 = 0;  // 1: O is f32 instead of T (decode step feeds the f32 GEMV)
  // 1 (T = float only): O is a split-pair matrix (common.h sp16: the A operand of the GEMM that follows in
  // ANYREF_MODE_PARITY16); o_bs / o_rs are whole matrix rows (multiples of 64 elements), o_hs % 4 == 0
  int o_split = 0;
  // 1 (T = float only, head dim 64 / 80 / 128): the f32 operands are multiplied as bf16 PAIRS on the 16-bit MFMA (three
  // passes per product, attention.hip attn_sp_body) instead of the f32 MFMA -- same f32-level result at 3/16 of the MFMA time
  int sp16 = 0;
  // split-pair kernel, set by its launcher when a capped launch is cut into head groups: this launch covers heads h_off ..
  // h_off + H - 1 of h_total (0: H) heads -- strides and the rel_h / rel_w arrays span all of them
  int h_off = 0, h_total = 0;
  int max_wg = 0;  // > 0: at most this many workgroups (bf16, head dim 80, the SAM forms); each walks several blocks
  // set by the launcher: keys split over kv_splits workgroups per query block, partials merged afterwards
  int kv_splits = 1;
  float* part_o = nullptr;   // [splits][B][H][Sq][hd] un-normalised O
  float* part_ml = nullptr;  // [splits][B][H][Sq][2] running max, running sum
};
template <typename T>
void launch_attention(const AttnArgs& a, hipStream_t s);
// whether launch_attention<T> runs this (non-causal, full-length) shape in the resident-key form, which takes the
// rel-pos TABLES (rel_tab_*) instead of a precomputed P
bool attention_takes_rel_tables(int elem_bytes, int hd, int Sq, int Sk, int kh, int kw, bool split = false);

// Decode-step attention for ONE new token per sequence, fused with RoPE and the KV-cache append:
// qkv f32 [B,3*H*hd] (this step's projections) -> rotates q,k at pos[b], appends k,v to the cache,
// attends over keys [0, pos[b]] and writes out f32 [B,H*hd].  Returns false if hd is unsupported
// (caller falls back to rope_cache + the generic attention kernel).
template <typename T>
bool launch_decode_attn(const float* qkv, int B, int H, int hd, const int* pos, const float* cs_tab, void* kc,
                        void* vc, int maxS, float scale, float* out, void* q_keep, hipStream_t s);

// Head-mean softmax row of ONE query per batch element over keys [0, kv_len):
// out[b, j] = mean_h softmax_j(scale * q[b,h].k[b,j,h])   (anyref.py:748-749)
template <typename T>
void launch_attn_row_mean(const void* q, int64_t q_bs, int64_t q_hs, const void* K, int64_t k_bs,
                          int64_t k_rs, int64_t k_hs, const int* kv_len, int B, int H, int hd,
                          float scale, float* out, int ld_out, hipStream_t s);

// ---- SAM decomposed rel-pos bias (image_encoder.py:354-392) -------------------------------
// rel_h[b,h,(y,x),ky] = sum_c q[b,(y,x),h,c] * tab_h[y - ky + size-1, c]; likewise rel_w with x.
template <typename T>
void launch_rel_pos(const void* q, int64_t q_bs, int64_t q_rs, int64_t q_hs, const float* tab_h,
                    const float* tab_w, int B, int H, int size, int hd, float* rel_h, float* rel_w,
                    hipStream_t s);

// ---- data movement / elementwise ------------------------------------------------------------
// non-overlapping patch im2col: img f32 [B,3,S,S] -> out T [B*g*g, Kp], k = c*p*p + ky*p + kx
template <typename T>
void launch_im2col_patch(const float* img, int B, int S, int p, void* out, int Kp, hipStream_t s);
// 3x3 pad-1 im2col on NHWC tokens: in T [B,g,g,C] -> out T [B*g*g, 9*C], k = (ky*3+kx)*C + c
template <typename T>
void launch_im2col_3x3(const void* in, int B, int g, int C, void* out, hipStream_t s);
// generic f32 -> T convert / copy with strides (rows x cols)
template <typename T>
void launch_convert(const float* in, int64_t ld_in, void* out, int64_t ld_out, int rows, int cols,
                    hipStream_t s);
// split-pair (sp16) rows -> f32: out[r, c] = hi + lo; ld_in in sp16 elements (a multiple of 64)
void launch_unsplit(const void* in, int64_t ld_in, float* out, int64_t ld_out, int rows, int cols, hipStream_t s);
// out[m, :] = a[m, :] + b[m % bmod, :]   (f32; position embeddings, keys+pe ...)
void launch_add_rows(const float* a, const float* b, int bmod, float* out, int M, int D, hipStream_t s);
template <typename T>
void launch_add_rows_to(const float* a, const float* b, int bmod, void* out, int M, int D, hipStream_t s);
// CLIP embeddings: x[b,0]=cls+pos[0]; x[b,1+i]=patch[b,i]+pos[1+i]   (f32)
void launch_clip_assemble(const float* patch, const float* cls, const float* pos, float* x, int B, int n,
                          int D, hipStream_t s, int rows = 0);  // rows > n + 1: zeroed spare rows per item
// ImageBind audio stem im2col (1 input channel, k x k kernel, stride st < k) and head tail (L2-normalise x scale)
template <typename T>
void launch_im2col_conv1(const float* img, int n, int Hh, int Ww, int k, int st, void* out, hipStream_t s);
void launch_l2norm_scale(const float* x, int rows, int D, float scale, float* y, hipStream_t s);
// token embedding gather + multimodal splice (LLaVA prepare_inputs; see oracle.splice_embeddings)
// ids i64 dev [B,Lmax], lens i32 dev [B]; image placeholder (-200) expands to n_img rows of img_feat[b].
// extra rows replace 1:1.  x f32 [B,Smax,D]; out_len i32 dev [B] (= len + n_img - 1 when an image is present).
void launch_embed_splice(const int64_t* ids, const int* lens, int B, int Lmax, const void* emb_table,
                         int emb_is_bf16, int vocab, const float* img_feat, int n_img, float* x, int Smax,
                         int D, int* out_len, hipStream_t s);
void launch_scatter_rows(const float* rows, const int* dst_b, const int* dst_pos, int n, float* x, int Smax,
                         int D, hipStream_t s);
// RoPE (rotate_half form) on q,k of a fused qkv buffer + append k,v to the cache.
// qkv T [B,S,3,H,hd]; pos0 i32 [B] dev; q_out T [B,S,H,hd]; kc/vc T [B,maxS,H,hd];
// cs_tab f32 [maxS][2][hd/2] = cos | sin of pos * inv_freq (built on the host like HF does)
// same from two f32 split-K slices of the projection (slab0 + slab1, rounded to bf16 first like the GEMM's own output);
// out_f32: q_out / kc / vc / q_keep are f32 and the sum is not rounded (ANYREF_MODE_PARITY16)
void launch_rope_cache_slabs(const float* slab0, const float* slab1, int B, int S, int H, int hd, const int* pos0,
                             const int* lens, const float* cs_tab, void* q_out, void* kc, void* vc, int maxS, void* q_keep,
                             hipStream_t s, bool out_f32 = false);
template <typename T>
void launch_rope_cache(const void* qkv, int B, int S, int H, int hd, const int* pos0, const int* lens,
                       const float* cs_tab, void* q_out, void* kc, void* vc, int maxS, void* q_keep,
                       hipStream_t s);
// same from an f32 qkv row buffer [B,3*H*hd] (decode step)
// q_keep (optional): rotated q also stored at [b, pos] of a [B,maxS,H,hd] buffer (rephrase branch)
template <typename T>
void launch_rope_cache_f32(const float* qkv, int B, int H, int hd, const int* pos, const float* cs_tab,
                           void* q_out, void* kc, void* vc, int maxS, void* q_keep, hipStream_t s);
// token embedding rows for the decode step: x[b,:] = table[ids[b],:]
void launch_embed_rows(const int64_t* ids, int B, const void* table, int is_bf16, int D, float* x, hipStream_t s);
// row_map[b] = b*maxS + pos[b]; kvlen[b] = pos[b] + 1
void launch_decode_index(const int* pos, int B, int maxS, int* row_map, int* kvlen, hipStream_t s);
// tokens[i, 0:n_out] = out_tokens; tokens[i, n_out] = pred[i]   (mask_decoder.py:127-141)
void launch_build_tokens(const float* out_tokens, int n_out, const float* pred, int n, int C, float* tokens,
                         hipStream_t s);
// act[m, j] = silu(gu[m, j]) * gu[m, F + j]
template <typename T>
void launch_swiglu(const void* gu, int M, int F, void* out, hipStream_t s);
// argmax over f32 rows (first index on ties) -> i64
void launch_argmax(const float* x, int M, int N, int ldx, int64_t* out, hipStream_t s, int* bump = nullptr);
// argmax, pos[b] += 1, and the next decode step's inputs in the same launch: x_next[b] = table[argmax], row_map[b] =
// b * maxS + pos[b], kvlen[b] = pos[b] + 1
void launch_argmax_next(const float* x, int M, int N, int ldx, int64_t* out, int* pos, const void* table, int is_bf16,
                        int D, int maxS, float* x_next, int* row_map, int* kvlen, hipStream_t s);
// ConvTranspose2d k2s2 output un-shuffle (+ LayerNorm2d + GELU): tmp f32 [n*g*g, 4*C] (col = (dy*2+dx)*C+c)
// -> out T [n*(2g)*(2g), C] NHWC
template <typename T>
void launch_upscale1(const float* tmp, int n, int g, int C, const float* ln_g, const float* ln_b, float eps,
                     void* out, hipStream_t s);
// second ConvT un-shuffle + GELU + hypernetwork product:
// masks[i,t,Y,X] = sum_c hyper[i,t,c] * gelu(tmp[i,(Y/2,X/2),(Y%2*2+X%2)*C + c])
void launch_upscale2_masks(const float* tmp, const float* hyper, int n, int ntok, int g2, int C, float* masks,
                           hipStream_t s);
// Sam.postprocess_masks (sam.py:137-172): two chained bilinear resizes (align_corners=False) + crop.
// low f32 [n, lh, lw] (lstride between masks) -> out [n, H, W]
void launch_postprocess(const float* low, int64_t lstride, int n, int lh, int lw, int S, int rh, int rw, int H,
                        int W, float* out, hipStream_t s);
// dense PE table (prompt_encoder.py:189-229): out f32 [g*g, 2*F], gauss f32 [2,F]
void launch_dense_pe(const float* gauss, int g, int F, float* out, hipStream_t s);
void launch_fill_i32(int* p, int v, int n, hipStream_t s);
void launch_add_i32(int* p, int v, int n, hipStream_t s);
template <typename T>
void launch_fill_rows_bias(void* dst, int ld, const int* rows, int nrows, const float* bias, int N, hipStream_t s);
// SURVEY.md §8 f-2 / f-1 (the steps right after / before the path)
void launch_iou_counts(const float* logits, const uint8_t* target, int n, int64_t hw, int64_t* counts, hipStream_t s);
void launch_avs_counts(const float* logits, const uint8_t* target, int n, int64_t hw, const float* cuts, int nth,
                       float cut_pred, int64_t* conf, int64_t* hist, hipStream_t s);
void launch_pool_ref_tokens(const float* f, int n, int L, int H, int n_out, float* out, hipStream_t s);
void launch_pil_resample_u8(const uint8_t* in, int H, int W, int C, uint8_t* tmp, uint8_t* out, int ow, int oh,
                            const int* xbounds, const int* xk, int kx, const int* ybounds, const int* yk, int ky,
                            hipStream_t s);
void launch_clip_finish(const uint8_t* img, int ih, int iw, int y0, int x0, int h, int w, int S, const float* mean,
                        const float* std_, float* out, hipStream_t s);
// Kaldi log-mel filterbank of one clip + pad / cut to target_len + Normalize (data.py:28-64,152-153): wave f32 [C, T] dev
// (channel 0 is analysed, the clip mean is over all channels), banks f32 [n_mel, padded / 2 + 1] dev, tw f64 [2 * padded] dev
// (cos then sin of 2 pi i / padded), scratch f64 [1] dev -> out f32 [n_mel, target_len] dev
void launch_kaldi_fbank(const float* wave, int C, int T, int win, int shift, int padded, float preemph, const float* banks,
                        int n_mel, const double* tw, double* scratch, int target_len, float mean, float stdv, float* out,
                        hipStream_t s);
void launch_sam_preprocess(const uint8_t* img, int h, int w, int S, const float* mean, const float* std_, float* out,
                           hipStream_t s);
// per-row broadcast add: out[m,:] = a[m,:] + v[:]  (f32)
void launch_add_vec(const float* a, const float* v, float* out, int M, int D, hipStream_t s);
// gather rows: out[i,:] = x[b[i], pos[i], :]
void launch_gather_rows(const float* x, int Smax, int D, const int* b, const int* pos, int n, float* out,
                        hipStream_t s);
// y += w * sum_j p[j] * X[j,:]  with p normalised to sum 1 over [s,e)   (rephrase, anyref.py:746-755)
void launch_rephrase(const float* hidden_b, int D, const float* attn_row, int s0, int e0, float weight,
                     float* y, hipStream_t s);
// convert generic dtype weight to f32 (dtype codes of anyref_hip.h)
void launch_to_f32(const void* in, int dtype, float* out, int64_t n, hipStream_t s);



}  // namespace anyref

// Stage orchestration of the AnyRef inference path on one MI355X.
//
//   encode_images : CLIP ViT tower (hidden_states[-2], CLS dropped) -> mm_projector      (anyref.py:334)
//   llm           : LLaVA splice -> LLaMA prefill -> greedy decode with a KV cache        (anyref.py:704-718)
//   handoff       : [SEG] hidden state (+ rephrase) -> text_hidden_fcs                    (anyref.py:723-770)
//   sam_encode    : ImageEncoderViT (windowed / global attention + rel-pos) + neck       (image_encoder.py)
//   mask_decode   : prompt encoder (text) + two-way transformer + upscaler + postprocess  (anyref.py:797-819)
//
// Template parameter T is the storage type of the big operands: float (parity mode) or bf16
// (perf mode).  The residual streams, every accumulation, the [SEG] hand-off and the whole mask
// decoder are fp32 in both modes (the decoder is 3.6 GFLOP: running it on the f32 MFMA costs
// microseconds and removes its bf16 error from the logits).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "model.h"

namespace anyref {

// =============================================================================================
// ModelBase
// =============================================================================================
ModelBase::ModelBase(const anyref_config& c, int device) : cfg(c), device_(device) { HIP_TRY(hipSetDevice(device)); }

ModelBase::~ModelBase() {
  (void)hipSetDevice(device_);
  (void)hipDeviceSynchronize();
  for (auto& kv : allocs_) (void)hipFree(kv.first);
}

void* ModelBase::dalloc(size_t bytes) {
  if (bytes == 0) bytes = 16;
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, bytes));
  allocs_[p] = bytes;
  bytes_ += (int64_t)bytes;
  return p;
}
void ModelBase::dfree(void* p) {
  auto it = allocs_.find(p);
  if (it == allocs_.end()) return;
  bytes_ -= (int64_t)it->second;
  (void)hipFree(p);
  allocs_.erase(it);
}

void ModelBase::set_weight(const char* name, const void* ptr, int is_device, int dtype, const int64_t* shape,
                           int ndim) {
  HIP_TRY(hipSetDevice(device_));
  if (finalized_) throw std::runtime_error("set_weight after finalize");
  RawTensor t;
  t.shape.assign(shape, shape + ndim);
  const int64_t n = t.numel();
  const size_t esz = dtype == ANYREF_F32 ? 4 : 2;
  if (dtype != ANYREF_F32 && dtype != ANYREF_BF16 && dtype != ANYREF_F16) throw std::runtime_error("bad dtype");
  auto it = raw_.find(name);
  if (it != raw_.end()) {
    dfree(it->second.p);
    raw_.erase(it);
  }
  t.p = (float*)dalloc((size_t)n * 4);
  const void* src = ptr;
  void* staged = nullptr;
  if (!is_device) {
    if (dtype == ANYREF_F32) {
      HIP_TRY(hipMemcpy(t.p, ptr, (size_t)n * 4, hipMemcpyHostToDevice));
      raw_[name] = t;
      return;
    }
    HIP_TRY(hipMalloc(&staged, (size_t)n * esz));
    HIP_TRY(hipMemcpy(staged, ptr, (size_t)n * esz, hipMemcpyHostToDevice));
    src = staged;
  }
  launch_to_f32(src, dtype, t.p, n, 0);
  HIP_TRY(hipStreamSynchronize(0));
  if (staged) (void)hipFree(staged);
  raw_[name] = t;
}

const RawTensor& ModelBase::raw(const std::string& name) const {
  auto it = raw_.find(name);
  if (it == raw_.end()) throw std::runtime_error("missing weight: " + name);
  return it->second;
}
void ModelBase::drop_raw() {
  for (auto& kv : raw_)
    if (kv.second.p) dfree(kv.second.p);
  raw_.clear();
}

// =============================================================================================
// Model<T>
// =============================================================================================
constexpr int kRowPadBytes = 128;  // see Lin::ld
// Storage roles of an arithmetic mode.  T names the GEMM A-operand (activation) type; W is what the weights are packed as,
// Q what the attention kernels read (q / k / v rows, the KV cache).  float / bf16 / f16: all three are T.  sp16
// (ANYREF_MODE_PARITY16): activations as split bf16 pairs, weights bf16 exactly as stored, attention operands f32.
template <typename T>
struct ModeTypes {
  using W = T;
  using Q = T;
};
template <>
struct ModeTypes<sp16> {
  using W = bf16;
  using Q = float;
};
template <typename T>
struct Lin {  // nn.Linear packed in T
  T* w = nullptr;
  float* b = nullptr;
  int n = 0, k = 0;  // k = padded row length
  // Row stride in elements (0: k).  The LLM linears keep 128 bytes between rows: decode GEMVs stream thousands
  // of rows at once, and with a stride that is a large power of two (or 512 * odd, K = 11008) the concurrent
  // rows alias onto few HBM channels -- measured 20.0 -> 18.0 us for down_proj, 18.1 -> 15.6 us at K = 8192.
  int ld = 0;
  int stride() const { return ld ? ld : k; }
  // fp8 weight-only mode (LLM linears): e4m3 bytes [n, k] + one f32 scale per output row; w stays null
  uint8_t* w8 = nullptr;
  float* ws = nullptr;
};
struct LinF {  // nn.Linear kept in f32
  float* w = nullptr;
  float* b = nullptr;
  int n = 0, k = 0;
};
struct Affine {
  float* g = nullptr;
  float* b = nullptr;
};

// T: storage type of the LLaMA / CLIP / audio operands (float: parity, bf16: perf).  TS: storage type of the SAM image
// encoder's operands -- f16 in the perf build: per-stage attribution on the parity workload (tests/test_gpu_c2_full.py,
// DESIGN.md §3) puts 3.8e-2 of the bf16 build's 3.7e-2 mask-logit error in this tower alone (CLIP 2.4e-3, LLaMA 8e-3);
// f16 has 3 more mantissa bits at the same MFMA rate and bytes, and is what the reference runs it in
// (eval_referseg.py:70-72).  ANYREF_SAM_BF16=1 builds the all-bf16 handle (A/B).
template <typename T, typename TS = T>
class Model : public ModelBase {
 public:
  using W = typename ModeTypes<T>::W;    // LLaMA / CLIP / audio weights
  using Q = typename ModeTypes<T>::Q;    // ... attention operands and the KV cache
  using WS = typename ModeTypes<TS>::W;  // SAM encoder weights
  using QS = typename ModeTypes<TS>::Q;  // ... attention operands
  static constexpr bool SPT = is_split<T>::value, SPS = is_split<TS>::value;
  static constexpr bool IS16 = sizeof(T) == 2 || SPT;    // the 16-bit MFMA paths (bf16 or split pairs)
  static constexpr bool IS16S = sizeof(TS) == 2 || SPS;
  // the q / k / v projections write the attention operand type: f32 in the f32 and the split-pair modes
  static constexpr bool QF32 = std::is_same<Q, float>::value, QF32S = std::is_same<QS, float>::value;
  // row stride of an A-operand matrix with `k` logical columns: split-pair rows are whole 64-column blocks
  template <typename E>
  static int apad(int k) {
    return is_split<E>::value ? round_up(k, 64) : k;
  }
  Model(const anyref_config& c, int device) : ModelBase(c, device) {
    fp8w_ = c.mode == ANYREF_MODE_PERF_FP8W;
    if (fp8w_ && sizeof(T) != 2) throw std::runtime_error("fp8 weights need the bf16 compute mode");
    // the split-pair encoder is twice as long as the 16-bit one: spread over 8 decode steps instead of 6
    // (scratch/side_share.py, parity16 at C2: 49.2 - 49.9 ms with 6, 48.3 - 48.6 with 8 - 10; caps other than 128 lose 1 - 10 ms)
    if (SPS && !getenv("ANYREF_SIDE_STEPS")) side_steps_ = 8;
  }
  ~Model() override {
    (void)hipSetDevice(device_);
    (void)hipDeviceSynchronize();
    for (auto& kv : decode_graphs_) (void)hipGraphExecDestroy(kv.second);
    // workspaces are pooled per stream handle: drop the ones of the streams this handle owns, or a later
    // stream that recycles the handle value would inherit them
    for (hipStream_t st : {cap_stream_, s2_})
      if (st) {
        gemm_release_workspace(st);
        attn_release_workspace(st);
        (void)hipStreamDestroy(st);
      }
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    if (ev_sam_) (void)hipEventDestroy(ev_sam_);
    if (next_host_) (void)hipHostFree(next_host_);
    if (stage_) (void)hipHostFree(stage_);
    for (auto e : ev_tok_)
      if (e) (void)hipEventDestroy(e);
    for (auto e : stage_ev_)
      if (e) (void)hipEventDestroy(e);
  }
  const char* mode_name() const override {
    if (SPT) return "f32 activations as bf16 pairs x bf16 weights";
    return sizeof(T) == 2 ? (fp8w_ ? (is_half16<TS>::value ? "bf16+fp8w, SAM f16" : "bf16+fp8w")
                                   : (is_half16<TS>::value ? "bf16, SAM f16" : "bf16"))
                          : "f32";
  }
  void finalize() override;
  void generate(hipStream_t s, const float* clip_images, const float* sam_images, const int64_t* input_ids,
                const int32_t* lens, int B, int Lmax, const float* extra_embeds, const int32_t* extra_slots,
                int n_extra, const int32_t* resized_hw, const int32_t* orig_hw, int max_new_tokens,
                int eos_token_id, int64_t* out_ids, int32_t* out_lens, int32_t* out_nseg, float* out_masks,
                int64_t out_masks_cap, int64_t* mask_offsets, float* out_low, float* out_hidden) override;
  void forward_teacher(hipStream_t s, const float* clip_images, const float* sam_images, const int64_t* input_ids,
                       const int32_t* lens, int B, int Lmax, const float* extra_embeds, const int32_t* extra_slots,
                       int n_extra, const int32_t* rephrase_start, const int32_t* resized_hw,
                       const int32_t* orig_hw, int32_t* out_nseg, float* out_masks, int64_t out_masks_cap,
                       int64_t* mask_offsets, float* out_low, float* out_hidden, float* out_logits) override;
  void encode_images(hipStream_t s, const float* clip_images, int B, float* out, float* clip_feat) override;
  void sam_encode(hipStream_t s, const float* sam_images, int B, float* out) override;
  void mask_decode(hipStream_t s, const float* image_emb, const float* pred_emb, int n, float* masks4, float* iou,
                   const int32_t* resized_hw, const int32_t* orig_hw, float* out_masks) override;
  void llm_forward(hipStream_t s, const float* embeds, const int32_t* lens, int B, int S, float* hidden,
                   float* logits, const int32_t* attn_q, float* attn_row) override;
  void project_audio(hipStream_t s, const float* audio_emb, int n, float* out) override;
  void audio_encode(hipStream_t s, const float* mel, int n, float* emb) override;
  void seg_tail(hipStream_t s, const float* sam_images, const int64_t* ids, const int32_t* ids_lens,
                const int32_t* ref_pos, int B, int Lmax, int teacher, const float* hidden, int hidden_rows,
                const float* attn_mean, const int32_t* resized_hw, const int32_t* orig_hw, int32_t* out_nseg,
                float* out_masks, int64_t out_masks_cap, int64_t* mask_offsets, float* out_low) override;
  void join_sam_public(hipStream_t s) { join_sam(s); }

 private:
  // ---- packing helpers ----
  float* own_f32(const std::string& name);  // take the raw f32 copy as is
  float* upload_f32(const std::vector<float>& v);
  std::vector<float> to_host(const std::string& name);
  template <typename E>
  E* pack_rows(E* dst, int dst_row0, const std::string& name, int rows, int cols, int kpad);
  template <typename E = W>
  Lin<E> pack_linear(const std::string& wname, const std::string& bname, int n, int k, int kalign = 8,
                     int rowpad = 0);
  LinF pack_linear_f32(const std::string& wname, const std::string& bname, int n, int k);
  Affine affine(const std::string& prefix, bool bias = true);
  void resample_rel_pos(const std::string& name, int rows, int hd);
  template <typename U>
  U* talloc(size_t n) {
    return reinterpret_cast<U*>(dalloc(n * sizeof(U)));
  }
  // an A-operand (activation) matrix [rows, width]; split-pair matrices are zeroed once: the columns that pad a row to
  // whole 64-blocks are never written and meet zero weight columns
  template <typename U>
  U* aalloc(size_t rows, int width) {
    const size_t n = rows * (size_t)apad<U>(width);
    U* p = talloc<U>(n);
    if (is_split<U>::value) HIP_TRY(hipMemset(p, 0, n * sizeof(U)));
    return p;
  }

  // ---- op helpers ----
  // nrm / nrm_out: RMSNorm (gain nrm->g, llm eps) of the output rows written to nrm_out as T when the GEMM
  // takes its split-K path; returns whether that happened (else the caller runs the norm itself)
  template <typename E>
  bool gemm(hipStream_t s, const E* A, int lda, const Lin<typename ModeTypes<E>::W>& l, void* C, int ldc, int M, int act, bool c_f32,
            const float* resid = nullptr, int ldr = 0, const int* row_map = nullptr, const Affine* nrm = nullptr,
            void* nrm_out = nullptr, bool swiglu = false, float ln_eps = -1.f, const int* a_row_map = nullptr) {  // nrm_out: E rows
    // nrm / nrm_out: the norm that follows (RMSNorm with the LLM's eps; LayerNorm with ln_eps when ln_eps >= 0) is
    // applied by the split-K reduction if the GEMM takes that path -- the return value says whether it did
    GemmArgs a;
    bool fused = false;
    a.swiglu_pairs = swiglu ? 1 : 0;
    if (nrm && nrm_out) {
      a.norm_gain = nrm->g; a.norm_out = nrm_out; a.norm_ld = l.n; a.norm_eps = cfg.llm_rms_eps; a.norm_done = &fused;
      if (ln_eps >= 0.f) {
        a.norm_bias = nrm->b;
        a.norm_eps = ln_eps;
      }
    }
    a.A = A; a.lda = apad<E>(lda); a.W = l.w; a.ldw = l.stride(); a.bias = l.b; a.C = C; a.ldc = c_f32 ? ldc : apad<E>(ldc);
    a.M = M; a.N = l.n;
    a.norm_ld = apad<E>(a.norm_ld);
    a.K = l.k; a.act = act; a.c_f32 = c_f32 ? 1 : 0; a.resid = resid; a.ldr = ldr; a.row_map = row_map;
    a.a_row_map = a_row_map;
    a.max_wg = cap_wg_;
    if constexpr (std::is_same<E, T>::value && !SPT) {
      if (l.w8) {
        if (l.k % 64 == 0) {  // fp8 bytes straight into the GEMM (widened to bf16 per fragment, scale in the epilogue)
          a.W = l.w8; a.w_fp8 = 1; a.col_scale = l.ws;
        } else {              // odd K: multiply a bf16 image of q * scale
          launch_dequant_fp8_rows(l.w8, l.stride(), l.ws, l.n, l.k, deq_buf_, l.k, s);
          a.W = deq_buf_; a.ldw = l.k;
        }
      }
    }
    launch_gemm<E>(a, s);
    return fused;
  }
  // nn.Linear (or a row range of a fused one) as the weight operand of a decode GEMV
  // Workgroups of the decode GEMVs in a step that runs beside the capped SAM encoder (0: the launcher's 512).  The encoder's
  // 256-row GEMM workgroups take every VGPR of their CU, so only the ~128 free CUs hold GEMV workgroups (two each): the
  // stamps show a 512-workgroup launch starting its second half 7 - 16 us late, one round behind the first.  With 256
  // workgroups (one round on the free CUs) a co-running step takes 3.97 -> 3.85 ms (image 39.8 - 40.3 -> 39.5 - 40.1 on one
  // box; 384 with the wave-pair kernels unbalanced: 4.1); alone 512 stays better (2.77 vs 2.89 ms).  Same sums either way.
  int gemv_grid_ = 0;
  void gemv_w(GemvArgs& g, const Lin<W>& l, int row0 = 0) const {
    g.ldw = l.stride();
    g.grid = gemv_grid_;
    if (l.w8) {
      g.W = l.w8 + (size_t)row0 * l.stride();
      g.wscale = l.ws + row0;
      g.w_fp8 = 1;
    } else {
      g.W = l.w + (size_t)row0 * l.stride();
    }
  }
  void gemv_w2(GemvArgs& g, const Lin<W>& l, int row0) const {
    if (l.w8) {
      g.W2 = l.w8 + (size_t)row0 * l.stride();
      g.wscale2 = l.ws + row0;
    } else {
      g.W2 = l.w + (size_t)row0 * l.stride();
    }
  }
  bool fp8w_ = false;
  W* deq_buf_ = nullptr;  // bf16 image of the largest fp8 weight (prefill operand)
  // pack rows of a raw f32 tensor as fp8 + scales into l (rows [row0, row0 + rows))
  // rstride 2: every second row of l (gate / up interleave), starting at row0
  void pack_rows_fp8(Lin<W>& l, int row0, const std::string& name, int rows, int cols, int rstride = 1) {
    const RawTensor& t = raw(name);
    if (t.numel() != (int64_t)rows * cols || cols != l.k)
      throw std::runtime_error("shape mismatch for " + name + " (fp8 pack)");
    launch_quant_fp8_rows(t.p, cols, rows, cols, l.w8 + (size_t)row0 * l.stride(), l.stride() * rstride, l.ws + row0, 0,
                          rstride);
  }
  Lin<W> alloc_fp8(int n, int k) {
    if (k % 16) throw std::runtime_error("fp8 weights need K % 16 == 0");
    Lin<W> l;
    l.n = n;
    l.k = k;
    l.ld = k + kRowPadBytes;
    l.w8 = reinterpret_cast<uint8_t*>(dalloc((size_t)n * l.ld));
    HIP_TRY(hipMemset(l.w8, 0, (size_t)n * l.ld));
    l.ws = talloc<float>(n);
    return l;
  }
  bool skinny_off_ = getenv("ANYREF_NO_SKINNY_GEMV") != nullptr;
  void gemmf(hipStream_t s, const float* A, int lda, const LinF& l, float* C, int ldc, int M, int act,
             const float* resid = nullptr, int ldr = 0) {
    if (M >= 1 && M <= 8 && l.k <= 4096 && l.k % 4 == 0 && lda % 4 == 0 && !((uintptr_t)A & 15) && (!resid || ldr == ldc) &&
        !skinny_off_) {
      GemvArgs g;  // a handful of token rows: weight streaming, not a 64 x 64 tile walk (gemm.hip)
      g.x = A; g.ldx = lda; g.W = l.w; g.bias = l.b; g.y = C; g.ldy = ldc; g.resid = resid; g.B = M; g.N = l.n;
      g.K = l.k; g.act = act;
      launch_gemv_skinny_f32(g, s);
      return;
    }
    GemmArgs a;
    a.A = A; a.lda = lda; a.W = l.w; a.ldw = l.k; a.bias = l.b; a.C = C; a.ldc = ldc; a.M = M; a.N = l.n;
    a.K = l.k; a.act = act; a.c_f32 = 1; a.resid = resid; a.ldr = ldr;
    launch_gemm<float>(a, s);
  }
  template <typename E = T>  // E: type of y when it is not f32
  void norm(hipStream_t s, const float* x, int ldx, const Affine& af, void* y, int ldy, int M, int D, float eps,
            bool y_f32, bool rms = false, const int* row_map = nullptr, int act = ACT_NONE) {
    NormArgs a;
    a.x = x; a.ldx = ldx; a.gain = af.g; a.bias = af.b; a.y = y; a.ldy = y_f32 ? ldy : apad<E>(ldy); a.M = M; a.D = D; a.eps = eps;
    a.rms = rms ? 1 : 0; a.y_f32 = y_f32 ? 1 : 0; a.row_map = row_map; a.act = act;
    launch_norm<E>(a, s);
  }

  // ---- stages ----
  void clip_tower(hipStream_t s, const float* images, int B);  // -> img_feat_ [B,n,H]
  void llm_prefill(hipStream_t s, int B, int Sp, const int* lens_dev, bool keep_q);
  void llm_decode_step(hipStream_t s, int B, bool keep_q);
  // blocks [blk0, blk1) of the encoder; blk0 == 0 also runs the patch embedding, blk1 < 0 (= to the end) the neck
  void sam_encoder(hipStream_t s, const float* images, int B, float* out, int blk0 = 0, int blk1 = -1);
  void mask_decoder(hipStream_t s, const float* image_emb, const float* pred_emb, int n, float* masks4,
                    float* iou);
  void run_tail(hipStream_t s, const float* sam_images, int B, const std::vector<int>& seg_b,
                const std::vector<int>& seg_pos, const std::vector<int>& reph_s, const int32_t* resized_hw,
                const int32_t* orig_hw, int32_t* out_nseg, float* out_masks, int64_t out_masks_cap,
                int64_t* mask_offsets, float* out_low, const float* attn_given = nullptr, int attn_n = 0);
  void join_sam(hipStream_t s);
  // mask of a [SEG] the greedy loop has just emitted, on the side stream, while the loop goes on (generate, batch 1)
  void early_seg(hipEvent_t hidden_ready, int hidden_row, const int32_t* resized_hw, const int32_t* orig_hw,
                 float* out_masks, int64_t out_masks_cap, float* out_low);
  int early_done_ = 0;       // [SEG]s of this call already decoded by early_seg (in order of appearance)
  bool early_stop_ = false;  // a limit was hit: the tail decodes (or refuses) the rest
  // SAM image encoder on a second stream: it depends on nothing but the image, is MFMA-bound, and
  // overlaps the HBM-bound LLM decode (fork at the start of a call, join before the mask decoder).
  // fed = true: only the fork point is set; the encoder's blocks are queued by sam_feed() as the decode loop advances
  void fork_sam(hipStream_t s, const float* sam_images, int B, bool fed = false);
  // CU share of the side stream (generate, batch 1).  A decode GEMV's throughput is proportional to the CUs it gets
  // (21-24 GB/s per CU, with 192 CUs as with 256), and an encoder GEMM / attention launch holds all 256 with
  // workgroups a GEMV workgroup cannot sit beside: uncapped, the two decode steps the encoder overlaps take 7.0 ms
  // instead of 2.9 (kernel-side stamps).  So the encoder's GEMM / attention launches are capped at side_wgs_
  // workgroups (ModelBase; 0 = uncapped), each walking several tiles, and the blocks are queued a few per decode
  // step (sam_feed) instead of all at the fork: whatever is not queued when the loop ends (a short answer) runs
  // uncapped, alone on the chip.  cap_wg_ is what the launch helpers read: non-zero only while sam_feed queues.
  void sam_feed(int upto, bool capped);
  int cap_wg_ = 0;
  int sam_next_blk_ = 0;            // encoder blocks [0, sam_next_blk_) are queued on s2_
  bool sam_enq_done_ = false;       // ... and so is the neck: ev_sam_ covers the whole encoder
  const float* sam_img_ = nullptr;  // the call's images / batch (valid while sam_forked_)
  int sam_B_ = 0;
  struct PendingEarly {             // an early [SEG] mask that waits for the encoder to be queued in full
    hipEvent_t ready;
    int hidden_row;
    const int32_t *resized_hw, *orig_hw;
    float* out_masks;
    int64_t cap;
    float* out_low;
  };
  std::vector<PendingEarly> early_pending_;
  void early_seg_run(const PendingEarly& e);
  hipStream_t s2_ = nullptr;
  hipEvent_t ev_fork_ = nullptr, ev_sam_ = nullptr;
  bool sam_forked_ = false;
  // One decode step is ~170 launches whose arguments never change (position and next token live on
  // the device), so it is captured once per (batch, keep_q) and replayed as a hipGraph.
  void decode_step_graph(hipStream_t s, int B, bool keep_q, bool corun = false);
  hipStream_t cap_stream_ = nullptr;
  std::map<int, hipGraphExec_t> decode_graphs_;
  int splice_inputs(hipStream_t s, const int64_t* input_ids, const int32_t* lens, int B, int Lmax,
                    const float* extra_embeds, const int32_t* extra_slots, int n_extra,
                    std::vector<int>& slen, std::vector<int>& img_pos);
  void ensure_q_last();

  // ---- CLIP ----
  struct ClipLayer {
    Affine ln1, ln2;
    Lin<W> qkv, out, fc1, fc2;
  };
  int clip_n_ = 0, clip_kp_ = 0;
  Lin<W> clip_patch_;
  float *clip_cls_ = nullptr, *clip_pos_ = nullptr;
  Affine clip_pre_;
  std::vector<ClipLayer> clip_layers_;
  Lin<W> mm_proj_;
  T *c_col_ = nullptr, *c_h_ = nullptr, *c_att_ = nullptr, *c_mlp_ = nullptr, *c_feat_ = nullptr;
  Q* c_qkv_ = nullptr;
  float *c_patch_ = nullptr, *c_x_ = nullptr, *img_feat_ = nullptr;

  // ---- LLM ----
  struct LlmLayer {
    Affine in_norm, post_norm;
    Lin<W> qkv, o, gu, down;
    W *gate_w = nullptr, *up_w = nullptr;  // views into gu.w
  };
  W* emb_table_ = nullptr;
  std::vector<LlmLayer> llm_layers_;
  Affine llm_norm_;
  Lin<W> lm_head_;
  float* rope_tab_ = nullptr;
  Q *kcache_ = nullptr, *vcache_ = nullptr, *q_last_ = nullptr;
  size_t cache_layer_stride_ = 0;
  float *l_x_ = nullptr, *hidden_all_ = nullptr, *l_logits_ = nullptr, *l_xlast_ = nullptr;
  T *l_h_ = nullptr, *l_att_ = nullptr, *l_act_ = nullptr;
  Q *l_qkv_ = nullptr, *l_q_ = nullptr;
  float *d_x_ = nullptr, *d_qkv_ = nullptr, *d_att_ = nullptr, *d_act_ = nullptr;
  Q* d_q_ = nullptr;
  int64_t *ids_dev_ = nullptr, *next_dev_ = nullptr;
  float* qkv_slabs_ = nullptr;  // [2][<= 320][3H] f32: the two K slices of the prefill qkv projection
  bool qkv_slabs_off_ = getenv("ANYREF_NO_QKV_SLABS") != nullptr;
  int *lens_dev_ = nullptr, *slen_dev_ = nullptr, *pos_dev_ = nullptr, *kvlen_dev_ = nullptr, *rowmap_dev_ = nullptr,
      *idx_a_ = nullptr, *idx_b_ = nullptr;
  int64_t* next_host_ = nullptr;  // pinned, two slots of max_batch tokens (the greedy loop runs one step ahead)
  hipEvent_t ev_tok_[2] = {nullptr, nullptr};
  // pinned int staging for the small index vectors a call uploads (no host wait for a pageable copy to drain).
  // One region per purpose, each with an event recorded behind its last queued copy: the host waits on that event
  // before it rewrites the region (stage_begin), so a call that returns with copies still queued -- seg_tail and
  // forward_teacher never sync, and run_tail's copies sit in front of a 10+ ms encoder -- cannot have them read the
  // NEXT call's indices.  The wait is free whenever the copy has already run (always, in generate's loop).
  int* stage_ = nullptr;
  int *stage_first_ = nullptr, *stage_extra_ = nullptr, *stage_seg_ = nullptr, *stage_kl_ = nullptr;
  enum { ST_FIRST = 0, ST_EXTRA, ST_SEG, ST_KL, ST_N };
  hipEvent_t stage_ev_[ST_N] = {nullptr, nullptr, nullptr, nullptr};
  bool stage_busy_[ST_N] = {false, false, false, false};
  void stage_begin(int r) {
    if (stage_busy_[r]) HIP_TRY(hipEventSynchronize(stage_ev_[r]));
    stage_busy_[r] = false;
  }
  void stage_end(int r, hipStream_t s) {
    HIP_TRY(hipEventRecord(stage_ev_[r], s));
    stage_busy_[r] = true;
  }
  // stage_begin ... copies ... stage_end, with the event recorded on EVERY exit: a throw between the first queued copy and
  // stage_end must not leave the region marked free while a copy still reads it
  struct StageScope {
    Model* m;
    int r;
    hipStream_t s;
    StageScope(Model* m_, int r_, hipStream_t s_) : m(m_), r(r_), s(s_) { m->stage_begin(r); }
    ~StageScope() {
      try {
        m->stage_end(r, s);
      } catch (...) {
      }
    }
  };

  // ---- ImageBind audio trunk (f-4; present iff cfg.aud_blocks > 0) ----
  struct AudBlock {
    Affine ln1, ln2;
    Lin<W> qkv, out, fc1, fc2;
    float *bias_k = nullptr, *bias_v = nullptr;
  };
  std::vector<AudBlock> aud_blocks_;
  Lin<W> aud_stem_, aud_head_;
  Affine aud_stem_ln_, aud_head_ln_;
  float *aud_cls_ = nullptr, *aud_pos_ = nullptr, aud_scale_ = 20.f;
  int aud_np_ = 0, aud_rows_ = 0;   // patches per clip; rows per clip in the work buffers (tokens + 1 spare)
  int* aud_kvrow_ = nullptr;        // the spare row of every clip (holds bias_k / bias_v as the extra key / value)
  T *a_col_ = nullptr, *a_h_ = nullptr, *a_att_ = nullptr, *a_mlp_ = nullptr, *a_clsrow_ = nullptr;
  Q* a_qkv_ = nullptr;
  float *a_patch_ = nullptr, *a_x_ = nullptr, *a_emb_ = nullptr;

  // ---- glue ----
  LinF fc1_, fc2_;
  Lin<W> audio_proj_;
  bool has_audio_ = false;
  float *seg_h_ = nullptr, *seg_t_ = nullptr, *pred_emb_ = nullptr, *attn_row_ = nullptr;

  // ---- SAM encoder ----
  struct SamBlock {
    Affine ln1, ln2;
    Lin<WS> qkv, proj, lin1, lin2;
    Lin<QS> rel;  // (multiplied with q rows: the attention operand type) [2*Np, hd]: rows [0,2sz-1) = rel_pos_h, rows [Np, Np+2sz-1) = rel_pos_w, Np = 2*sz
    bool global = false;
  };
  Lin<WS> sam_patch_;
  float* sam_pos_ = nullptr;
  std::vector<SamBlock> sam_blocks_;
  Lin<WS> neck0_, neck2_;
  Affine neck1_, neck3_;
  int sam_g_ = 0, sam_nw_ = 0, sam_wrows_ = 0;  // grid, windows per side, window-layout rows per image
  int *win2tok_ = nullptr, *tok2win_ = nullptr, *pad_rows_ = nullptr;  // pad_rows_: window-layout rows with no token
  int n_pad_rows_ = 0;                                                  // per image
  TS *s_col_ = nullptr, *s_hglob_ = nullptr, *s_att_ = nullptr, *s_mlp_ = nullptr, *s_n1_ = nullptr, *s_col3_ = nullptr;
  QS* s_qkv_ = nullptr;
  float *s_x_ = nullptr, *s_relh_ = nullptr, *s_relw_ = nullptr, *s_n0_ = nullptr, *s_n2_ = nullptr,
        *sam_emb_ = nullptr;

  // ---- prompt encoder + mask decoder (f32) ----
  struct DecAttn {
    LinF q, k, v, o;
  };
  struct DecLayer {
    DecAttn self, t2i, i2t;
    Affine n1, n2, n3, n4;
    LinF lin1, lin2;
  };
  float *dense_pe_ = nullptr, *no_mask_ = nullptr, *out_tokens_ = nullptr;
  std::vector<DecLayer> dec_layers_;
  DecAttn dec_final_;
  Affine dec_norm_final_, up_ln_;
  LinF up1_, up2_;
  LinF hyper_[3], iou_head_[3];  // hyper: stacked over the mask tokens
  float *m_tokens_ = nullptr, *m_q_ = nullptr, *m_qp_ = nullptr, *m_keys_ = nullptr, *m_kp_ = nullptr,
        *m_qh_ = nullptr, *m_kh_ = nullptr, *m_vh_ = nullptr, *m_att_ = nullptr, *m_tmp_ = nullptr, *m_mlp_ = nullptr,
        *m_bigq_ = nullptr, *m_bigk_ = nullptr, *m_bigv_ = nullptr, *m_bigatt_ = nullptr, *m_up0_ = nullptr,
        *m_up1_ = nullptr, *m_up2_ = nullptr, *m_hy0_ = nullptr, *m_hy1_ = nullptr, *m_hyper_ = nullptr,
        *m_masks_ = nullptr, *m_iou_ = nullptr, *m_src_ = nullptr;
};

// ---------------------------------------------------------------------------------------------
// packing
// ---------------------------------------------------------------------------------------------
template <typename T, typename TS>
float* Model<T, TS>::own_f32(const std::string& name) {
  auto it = raw_.find(name);
  if (it == raw_.end()) throw std::runtime_error("missing weight: " + name);
  float* p = it->second.p;
  it->second.p = nullptr;  // ownership moves to the packed model (still tracked in allocs_)
  return p;
}
template <typename T, typename TS>
std::vector<float> Model<T, TS>::to_host(const std::string& name) {
  const RawTensor& t = raw(name);
  std::vector<float> v((size_t)t.numel());
  HIP_TRY(hipMemcpy(v.data(), t.p, v.size() * 4, hipMemcpyDeviceToHost));
  return v;
}
template <typename T, typename TS>
float* Model<T, TS>::upload_f32(const std::vector<float>& v) {
  float* p = talloc<float>(v.size());
  HIP_TRY(hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  return p;
}
template <typename T, typename TS>
template <typename E>
E* Model<T, TS>::pack_rows(E* dst, int dst_row0, const std::string& name, int rows, int cols, int kpad) {
  const RawTensor& t = raw(name);
  if (t.numel() != (int64_t)rows * cols)
    throw std::runtime_error("shape mismatch for " + name + ": expected " + std::to_string(rows) + "x" +
                             std::to_string(cols) + ", got " + std::to_string(t.numel()) + " elements");
  launch_convert<E>(t.p, cols, dst + (int64_t)dst_row0 * kpad, kpad, rows, cols, 0);
  return dst;
}
template <typename T, typename TS>
template <typename E>
Lin<E> Model<T, TS>::pack_linear(const std::string& wname, const std::string& bname, int n, int k, int kalign,
                                 int rowpad) {
  Lin<E> l;
  l.n = n;
  if (SPT || SPS) kalign = std::max(kalign, 64);  // split-pair A rows come in whole 64-column blocks (zero weight columns)
  l.k = round_up(k, kalign);
  l.ld = l.k + rowpad;
  l.w = talloc<E>((size_t)n * l.ld);
  if (l.ld != k) HIP_TRY(hipMemset(l.w, 0, (size_t)n * l.ld * sizeof(E)));
  pack_rows(l.w, 0, wname, n, k, l.ld);
  if (!bname.empty()) {
    if (raw(bname).numel() != n) throw std::runtime_error("bias shape mismatch for " + bname);
    l.b = own_f32(bname);
  }
  return l;
}
template <typename T, typename TS>
LinF Model<T, TS>::pack_linear_f32(const std::string& wname, const std::string& bname, int n, int k) {
  LinF l;
  l.n = n;
  l.k = k;
  if (raw(wname).numel() != (int64_t)n * k) throw std::runtime_error("shape mismatch for " + wname);
  if (k % 4) throw std::runtime_error("f32 linear needs K % 4 == 0: " + wname);
  l.w = own_f32(wname);
  if (!bname.empty()) l.b = own_f32(bname);
  return l;
}
template <typename T, typename TS>
Affine Model<T, TS>::affine(const std::string& prefix, bool bias) {
  Affine a;
  a.g = own_f32(prefix + ".weight");
  if (bias) a.b = own_f32(prefix + ".bias");
  return a;
}

// get_rel_pos's table fix-up (image_encoder.py:333-345): a rel_pos table [L, hd] with L != 2*size-1 is resampled along
// L with F.interpolate(mode="linear", align_corners=False) -- output row i at src = max((i + .5) * L / rows - .5, 0),
// blending rows floor(src) and floor(src) + 1 (clamped) with (1 - frac, frac), in fp32 like ATen.  Input-independent,
// so it is done on the host at finalize and the kernels only ever see (2*size-1)-row tables.
template <typename T, typename TS>
void Model<T, TS>::resample_rel_pos(const std::string& name, int rows, int hd) {
  const RawTensor& t = raw(name);
  if (t.shape.size() != 2 || t.shape[1] != hd || t.shape[0] < 1)
    throw std::runtime_error("rel_pos table " + name + " is not [L, head_dim]");
  const int L = (int)t.shape[0];
  if (L == rows) return;
  const std::vector<float> src = to_host(name);
  std::vector<float> dst((size_t)rows * hd);
  const float scale = (float)L / (float)rows;
  for (int i = 0; i < rows; ++i) {
    const float x = std::max(scale * ((float)i + 0.5f) - 0.5f, 0.f);
    const int i0 = std::min((int)floorf(x), L - 1), i1 = std::min(i0 + 1, L - 1);
    const float l1 = std::min(std::max(x - (float)i0, 0.f), 1.f), l0 = 1.f - l1;
    for (int d = 0; d < hd; ++d) dst[(size_t)i * hd + d] = l0 * src[(size_t)i0 * hd + d] + l1 * src[(size_t)i1 * hd + d];
  }
  RawTensor r;
  r.shape = {rows, hd};
  r.p = upload_f32(dst);
  dfree(raw_[name].p);
  raw_[name] = r;
}

static const char* CLIP_P = "model.vision_tower.vision_tower.vision_model.";
static const char* SAM_P = "model.visual_model.";

template <typename T, typename TS>
void Model<T, TS>::finalize() {
  HIP_TRY(hipSetDevice(device_));
  if (finalized_) return;
  const anyref_config& c = cfg;
  const int MB = c.max_batch;
  // ================= CLIP =================
  {
    const std::string p = CLIP_P;
    const int Dc = c.clip_dim, g = c.clip_image / c.clip_patch;
    clip_n_ = g * g;
    const int K = 3 * c.clip_patch * c.clip_patch;
    clip_patch_ = pack_linear(p + "embeddings.patch_embedding.weight", "", Dc, K, 64);  // 588 -> 640 (zero weights)
    clip_kp_ = clip_patch_.k;
    clip_cls_ = own_f32(p + "embeddings.class_embedding");
    clip_pos_ = own_f32(p + "embeddings.position_embedding.weight");
    clip_pre_ = affine(p + "pre_layrnorm");
    clip_layers_.resize(c.clip_layers_run);
    for (int i = 0; i < c.clip_layers_run; ++i) {
      const std::string lp = p + "encoder.layers." + std::to_string(i) + ".";
      ClipLayer& L = clip_layers_[i];
      L.ln1 = affine(lp + "layer_norm1");
      L.ln2 = affine(lp + "layer_norm2");
      L.qkv.n = 3 * Dc;
      L.qkv.k = Dc;
      if (SPT && Dc % 64) throw std::runtime_error("parity16: clip_dim must be a multiple of 64");
      L.qkv.w = talloc<W>((size_t)3 * Dc * Dc);
      L.qkv.b = talloc<float>(3 * Dc);
      const char* names[3] = {"q_proj", "k_proj", "v_proj"};
      for (int j = 0; j < 3; ++j) {
        pack_rows(L.qkv.w, j * Dc, lp + "self_attn." + names[j] + ".weight", Dc, Dc, Dc);
        HIP_TRY(hipMemcpy(L.qkv.b + j * Dc, raw(lp + "self_attn." + names[j] + ".bias").p, Dc * 4,
                          hipMemcpyDeviceToDevice));
      }
      L.out = pack_linear(lp + "self_attn.out_proj.weight", lp + "self_attn.out_proj.bias", Dc, Dc);
      L.fc1 = pack_linear(lp + "mlp.fc1.weight", lp + "mlp.fc1.bias", c.clip_mlp, Dc);
      L.fc2 = pack_linear(lp + "mlp.fc2.weight", lp + "mlp.fc2.bias", Dc, c.clip_mlp);
    }
    mm_proj_ = pack_linear("model.mm_projector.weight", "model.mm_projector.bias", c.llm_dim, Dc);
    const size_t R = (size_t)MB * (clip_n_ + 1);
    c_col_ = aalloc<T>((size_t)MB * clip_n_, clip_kp_);
    c_patch_ = talloc<float>((size_t)MB * clip_n_ * Dc);
    c_x_ = talloc<float>(R * Dc);
    c_h_ = aalloc<T>(R, Dc);
    c_qkv_ = talloc<Q>(R * 3 * Dc);
    c_att_ = aalloc<T>(R, Dc);
    c_mlp_ = aalloc<T>(R, c.clip_mlp);
    c_feat_ = aalloc<T>((size_t)MB * clip_n_, Dc);
    img_feat_ = talloc<float>((size_t)MB * clip_n_ * c.llm_dim);
  }
  // ================= LLaMA =================
  {
    const int H = c.llm_dim, F = c.llm_mlp, V = c.llm_vocab, S = c.llm_max_seq, nh = c.llm_heads, hd = H / nh;
    if (raw("model.embed_tokens.weight").numel() != (int64_t)V * H)
      throw std::runtime_error("embed_tokens shape mismatch (vocab/dim)");
    if (SPT && H % 64) throw std::runtime_error("parity16: llm_dim must be a multiple of 64");
    emb_table_ = talloc<W>((size_t)V * H);
    pack_rows(emb_table_, 0, "model.embed_tokens.weight", V, H, H);
    llm_layers_.resize(c.llm_layers);
    for (int i = 0; i < c.llm_layers; ++i) {
      const std::string lp = "model.layers." + std::to_string(i) + ".";
      LlmLayer& L = llm_layers_[i];
      L.in_norm = affine(lp + "input_layernorm", false);
      L.post_norm = affine(lp + "post_attention_layernorm", false);
      const char* names[3] = {"q_proj", "k_proj", "v_proj"};
      if (fp8w_) {
        L.qkv = alloc_fp8(3 * H, H);
        for (int j = 0; j < 3; ++j) pack_rows_fp8(L.qkv, j * H, lp + "self_attn." + names[j] + ".weight", H, H);
        L.o = alloc_fp8(H, H);
        pack_rows_fp8(L.o, 0, lp + "self_attn.o_proj.weight", H, H);
        L.gu = alloc_fp8(2 * F, H);  // rows interleaved: 2j = gate_j, 2j + 1 = up_j (SwiGLU in the GEMM epilogue)
        pack_rows_fp8(L.gu, 0, lp + "mlp.gate_proj.weight", F, H, 2);
        pack_rows_fp8(L.gu, 1, lp + "mlp.up_proj.weight", F, H, 2);
        L.down = alloc_fp8(H, F);
        pack_rows_fp8(L.down, 0, lp + "mlp.down_proj.weight", H, F);
      } else {
        const int rp = kRowPadBytes / (int)sizeof(W), ldh = H + rp;
        L.qkv.n = 3 * H;
        L.qkv.k = H;
        L.qkv.ld = ldh;
        L.qkv.w = talloc<W>((size_t)3 * H * ldh);
        HIP_TRY(hipMemset(L.qkv.w, 0, (size_t)3 * H * ldh * sizeof(W)));
        for (int j = 0; j < 3; ++j) pack_rows(L.qkv.w, j * H, lp + "self_attn." + names[j] + ".weight", H, H, ldh);
        L.o = pack_linear(lp + "self_attn.o_proj.weight", "", H, H, 8, rp);
        L.gu.n = 2 * F;
        L.gu.k = H;
        L.gu.ld = ldh;
        L.gu.w = talloc<W>((size_t)2 * F * ldh);  // rows interleaved: 2j = gate_j, 2j + 1 = up_j
        HIP_TRY(hipMemset(L.gu.w, 0, (size_t)2 * F * ldh * sizeof(W)));
        pack_rows(L.gu.w, 0, lp + "mlp.gate_proj.weight", F, H, 2 * ldh);
        pack_rows(L.gu.w + ldh, 0, lp + "mlp.up_proj.weight", F, H, 2 * ldh);
        L.gate_w = L.gu.w;
        L.up_w = L.gu.w + ldh;
        L.down = pack_linear(lp + "mlp.down_proj.weight", "", H, F, 8, rp);
        if (F % 8) throw std::runtime_error("llm_mlp must be a multiple of 8");
      }
      // free the raw copies of this layer early (7B in f32 is 27 GB)
      HIP_TRY(hipStreamSynchronize(0));
      for (const char* nm : {"self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                             "self_attn.o_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight",
                             "mlp.down_proj.weight"}) {
        auto it = raw_.find(lp + nm);
        if (it != raw_.end()) {
          dfree(it->second.p);
          raw_.erase(it);
        }
      }
    }
    llm_norm_ = affine("model.norm", false);
    if (fp8w_) {
      lm_head_ = alloc_fp8(V, H);
      pack_rows_fp8(lm_head_, 0, "lm_head.weight", V, H);
      const size_t big = std::max((size_t)2 * F * H, std::max((size_t)3 * H * H, (size_t)V * H));
      deq_buf_ = talloc<W>(big);
    } else {
      lm_head_ = pack_linear("lm_head.weight", "", V, H, 8, kRowPadBytes / (int)sizeof(W));
    }
    // rotary table, same fp32 op order as HF LlamaRotaryEmbedding
    std::vector<float> tab((size_t)S * hd);
    for (int pos = 0; pos < S; ++pos)
      for (int d = 0; d < hd / 2; ++d) {
        const float inv = 1.0f / powf(c.llm_rope_theta, (float)(2 * d) / (float)hd);
        const float fr = (float)pos * inv;
        tab[((size_t)pos * 2) * (hd / 2) + d] = cosf(fr);
        tab[((size_t)pos * 2 + 1) * (hd / 2) + d] = sinf(fr);
      }
    rope_tab_ = upload_f32(tab);
    cache_layer_stride_ = (size_t)MB * S * H;
    kcache_ = talloc<Q>(cache_layer_stride_ * c.llm_layers);
    vcache_ = talloc<Q>(cache_layer_stride_ * c.llm_layers);
    const size_t R = (size_t)MB * S;
    l_x_ = talloc<float>(R * H);
    hidden_all_ = talloc<float>(R * H);
    l_h_ = aalloc<T>(R, std::max(H, c.audio_dim));  // (project_audio stages its input rows here)
    l_qkv_ = talloc<Q>(R * 3 * H);
    if ((sizeof(T) == 2 || SPT) && (size_t)3 * H / 96 * 2 <= 256) qkv_slabs_ = talloc<float>((size_t)2 * 320 * 3 * H);  // prefill qkv K slices (R <= 320)
    l_q_ = talloc<Q>(R * H);
    l_att_ = aalloc<T>(R, H);
    l_act_ = aalloc<T>(R, F);
    l_logits_ = talloc<float>((size_t)MB * V);
    l_xlast_ = talloc<float>((size_t)MB * H);
    d_x_ = talloc<float>((size_t)MB * H);
    d_qkv_ = talloc<float>((size_t)MB * 3 * H);
    d_att_ = talloc<float>((size_t)MB * H);
    d_act_ = talloc<float>((size_t)MB * F);
    d_q_ = talloc<Q>((size_t)MB * H);
    ids_dev_ = talloc<int64_t>((size_t)MB * S);
    next_dev_ = talloc<int64_t>(MB);
    lens_dev_ = talloc<int>(MB);
    slen_dev_ = talloc<int>(MB);
    pos_dev_ = talloc<int>(MB);
    kvlen_dev_ = talloc<int>(MB);
    rowmap_dev_ = talloc<int>(MB);
    idx_a_ = talloc<int>((size_t)MB * std::max(c.max_seg, 64));
    idx_b_ = talloc<int>((size_t)MB * std::max(c.max_seg, 64));
    HIP_TRY(hipHostMalloc((void**)&next_host_, sizeof(int64_t) * MB * 2));
    for (auto& e : ev_tok_) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    {
      const int n_first = 3 * MB, n_extra_max = 2 * MB * std::max(c.max_seg, 64), n_seg = 2 * MB * c.max_seg;
      HIP_TRY(hipHostMalloc((void**)&stage_, sizeof(int) * (size_t)(n_first + n_extra_max + n_seg + MB)));
      stage_first_ = stage_;
      stage_extra_ = stage_first_ + n_first;
      stage_seg_ = stage_extra_ + n_extra_max;
      stage_kl_ = stage_seg_ + n_seg;
      for (auto& e : stage_ev_) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (c.rephrase_weight > 0.f) ensure_q_last();
  }
  // ================= glue =================
  {
    const int H = c.llm_dim;
    fc1_ = pack_linear_f32("model.text_hidden_fcs.0.0.weight", "model.text_hidden_fcs.0.0.bias", H, H);
    fc2_ = pack_linear_f32("model.text_hidden_fcs.0.2.weight", "model.text_hidden_fcs.0.2.bias", c.out_dim, H);
    has_audio_ = has_raw("model.audio_projector.weight");
    if (has_audio_)
      audio_proj_ = pack_linear("model.audio_projector.weight", "model.audio_projector.bias", H, c.audio_dim);
    const size_t ns = (size_t)MB * c.max_seg;
    seg_h_ = talloc<float>(ns * H);
    seg_t_ = talloc<float>(ns * H);
    pred_emb_ = talloc<float>(ns * c.out_dim);
    attn_row_ = talloc<float>((size_t)MB * c.llm_max_seq);
  }
  // ================= SAM image encoder =================
  {
    const std::string p = std::string(SAM_P) + "image_encoder.";
    const int D = c.sam_dim, g = c.sam_img / c.sam_patch, ws = c.sam_window, hd = D / c.sam_heads, C = c.sam_out_chans;
    const int nh_sam = c.sam_heads;
    sam_g_ = g;
    sam_nw_ = cdiv(g, ws);
    sam_wrows_ = sam_nw_ * sam_nw_ * ws * ws;
    if (SPS && (D % 64 || C % 64)) throw std::runtime_error("parity16: sam_dim / sam_out_chans must be multiples of 64");
    sam_patch_ = pack_linear<WS>(p + "patch_embed.proj.weight", p + "patch_embed.proj.bias", D, 3 * c.sam_patch * c.sam_patch);
    if (raw(p + "pos_embed").numel() != (int64_t)g * g * D) throw std::runtime_error("pos_embed shape mismatch");
    sam_pos_ = own_f32(p + "pos_embed");
    sam_blocks_.resize(c.sam_depth);
    for (int i = 0; i < c.sam_depth; ++i) {
      const std::string bp = p + "blocks." + std::to_string(i) + ".";
      SamBlock& L = sam_blocks_[i];
      for (int j = 0; j < c.sam_n_global; ++j)
        if (c.sam_global_idx[j] == i) L.global = true;
      L.ln1 = affine(bp + "norm1");
      L.ln2 = affine(bp + "norm2");
      L.qkv = pack_linear<WS>(bp + "attn.qkv.weight", bp + "attn.qkv.bias", 3 * D, D);
      L.proj = pack_linear<WS>(bp + "attn.proj.weight", bp + "attn.proj.bias", D, D);
      L.lin1 = pack_linear<WS>(bp + "mlp.lin1.weight", bp + "mlp.lin1.bias", c.sam_mlp_ratio * D, D);
      L.lin2 = pack_linear<WS>(bp + "mlp.lin2.weight", bp + "mlp.lin2.bias", D, c.sam_mlp_ratio * D);
      const int sz = L.global ? g : ws;
      // a table of another length (checkpoint trained at another window / image size) is resampled once, here
      resample_rel_pos(bp + "attn.rel_pos_h", 2 * sz - 1, hd);
      resample_rel_pos(bp + "attn.rel_pos_w", 2 * sz - 1, hd);
      const int Np = 2 * sz;
      // K padded with ZERO weights to a multiple of 64 (the fast GEMM's K tile): the A operand
      // then reads a few finite q/k values past this head's 80 columns, multiplied by zero.
      const int kp = round_up(hd, 64) <= 3 * D - (nh_sam - 1) * hd ? round_up(hd, 64) : hd;
      L.rel.n = 2 * Np;
      L.rel.k = kp;
      L.rel.w = talloc<QS>((size_t)2 * Np * kp);
      HIP_TRY(hipMemset(L.rel.w, 0, (size_t)2 * Np * kp * sizeof(QS)));
      pack_rows(L.rel.w, 0, bp + "attn.rel_pos_h", 2 * sz - 1, hd, kp);
      pack_rows(L.rel.w, Np, bp + "attn.rel_pos_w", 2 * sz - 1, hd, kp);
    }
    neck0_ = pack_linear<WS>(p + "neck.0.weight", "", C, D);
    neck1_ = affine(p + "neck.1");
    neck3_ = affine(p + "neck.3");
    {  // 3x3 conv weight [O][C][3][3] -> [O][(ky*3+kx)*C + c]
      std::vector<float> w = to_host(p + "neck.2.weight"), r((size_t)C * 9 * C);
      if (w.size() != r.size()) throw std::runtime_error("neck.2 shape mismatch");
      for (int o = 0; o < C; ++o)
        for (int ci = 0; ci < C; ++ci)
          for (int t = 0; t < 9; ++t) r[((size_t)o * 9 + t) * C + ci] = w[((size_t)o * C + ci) * 9 + t];
      float* rf = upload_f32(r);
      neck2_.n = C;
      neck2_.k = 9 * C;
      neck2_.w = talloc<WS>(r.size());
      launch_convert<WS>(rf, 9 * C, neck2_.w, 9 * C, C, 9 * C, 0);
      HIP_TRY(hipStreamSynchronize(0));
      dfree(rf);
    }
    // window <-> token row maps for up to MB images
    std::vector<int> w2t((size_t)MB * sam_wrows_), t2w((size_t)MB * g * g);
    for (int b = 0; b < MB; ++b)
      for (int wy = 0; wy < sam_nw_; ++wy)
        for (int wx = 0; wx < sam_nw_; ++wx)
          for (int ty = 0; ty < ws; ++ty)
            for (int tx = 0; tx < ws; ++tx) {
              const int y = wy * ws + ty, x = wx * ws + tx;
              const int wr = b * sam_wrows_ + ((wy * sam_nw_ + wx) * ws + ty) * ws + tx;
              if (y < g && x < g) {
                w2t[wr] = b * g * g + y * g + x;
                t2w[b * g * g + y * g + x] = wr;
              } else {
                w2t[wr] = -1;
              }
            }
    {
      std::vector<int> pads;
      for (size_t i = 0; i < w2t.size(); ++i)
        if (w2t[i] < 0) pads.push_back((int)i);
      n_pad_rows_ = (int)(pads.size() / MB);
      if (!pads.empty()) {
        pad_rows_ = talloc<int>(pads.size());
        HIP_TRY(hipMemcpy(pad_rows_, pads.data(), pads.size() * 4, hipMemcpyHostToDevice));
      }
    }
    win2tok_ = talloc<int>(w2t.size());
    tok2win_ = talloc<int>(t2w.size());
    HIP_TRY(hipMemcpy(win2tok_, w2t.data(), w2t.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(tok2win_, t2w.data(), t2w.size() * 4, hipMemcpyHostToDevice));
    const size_t RT = (size_t)MB * g * g, RW = std::max((size_t)MB * sam_wrows_, RT);
    s_col_ = aalloc<TS>(RT, sam_patch_.k);
    s_x_ = talloc<float>(RT * D);
    s_hglob_ = aalloc<TS>(RT, D);
    s_qkv_ = talloc<QS>(RW * 3 * D);
    s_att_ = aalloc<TS>(RW, D);
    s_mlp_ = aalloc<TS>(RT, c.sam_mlp_ratio * D);
    const size_t rel_g = (size_t)MB * c.sam_heads * g * g * 4 * g;          // [H][B*g*g][2*Np], Np = 2g
    const size_t rel_w = (size_t)MB * sam_wrows_ * c.sam_heads * 4 * ws;     // [H][B*wrows][2*Np], Np = 2ws
    s_relh_ = talloc<float>(std::max(rel_g, rel_w));
    s_n0_ = talloc<float>(RT * C);
    s_n1_ = aalloc<TS>(RT, C);
    s_col3_ = aalloc<TS>(RT, 9 * C);
    s_n2_ = talloc<float>(RT * C);
    sam_emb_ = talloc<float>(RT * C);
  }
  // ================= prompt encoder + mask decoder (f32) =================
  {
    const std::string pp = std::string(SAM_P) + "prompt_encoder.";
    const std::string p = std::string(SAM_P) + "mask_decoder.";
    const int C = c.sam_out_chans, g = sam_g_, NK = g * g, nt = c.num_mask_tokens;
    if (raw(pp + "pe_layer.positional_encoding_gaussian_matrix").numel() != C) throw std::runtime_error("gaussian matrix shape");
    dense_pe_ = talloc<float>((size_t)NK * C);
    launch_dense_pe(raw(pp + "pe_layer.positional_encoding_gaussian_matrix").p, g, C / 2, dense_pe_, 0);
    no_mask_ = own_f32(pp + "no_mask_embed.weight");
    out_tokens_ = talloc<float>((size_t)(nt + 1) * C);
    HIP_TRY(hipMemcpy(out_tokens_, raw(p + "iou_token.weight").p, C * 4, hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemcpy(out_tokens_ + C, raw(p + "mask_tokens.weight").p, (size_t)nt * C * 4, hipMemcpyDeviceToDevice));
    auto attn = [&](const std::string& ap, int internal) {
      DecAttn a;
      a.q = pack_linear_f32(ap + "q_proj.weight", ap + "q_proj.bias", internal, C);
      a.k = pack_linear_f32(ap + "k_proj.weight", ap + "k_proj.bias", internal, C);
      a.v = pack_linear_f32(ap + "v_proj.weight", ap + "v_proj.bias", internal, C);
      a.o = pack_linear_f32(ap + "out_proj.weight", ap + "out_proj.bias", C, internal);
      return a;
    };
    dec_layers_.resize(c.dec_depth);
    for (int i = 0; i < c.dec_depth; ++i) {
      const std::string lp = p + "transformer.layers." + std::to_string(i) + ".";
      DecLayer& L = dec_layers_[i];
      L.self = attn(lp + "self_attn.", C);
      L.t2i = attn(lp + "cross_attn_token_to_image.", C / 2);
      L.i2t = attn(lp + "cross_attn_image_to_token.", C / 2);
      L.n1 = affine(lp + "norm1");
      L.n2 = affine(lp + "norm2");
      L.n3 = affine(lp + "norm3");
      L.n4 = affine(lp + "norm4");
      L.lin1 = pack_linear_f32(lp + "mlp.lin1.weight", lp + "mlp.lin1.bias", c.dec_mlp, C);
      L.lin2 = pack_linear_f32(lp + "mlp.lin2.weight", lp + "mlp.lin2.bias", C, c.dec_mlp);
    }
    dec_final_ = attn(p + "transformer.final_attn_token_to_image.", C / 2);
    dec_norm_final_ = affine(p + "transformer.norm_final_attn");
    up_ln_ = affine(p + "output_upscaling.1");
    auto convT = [&](const std::string& name, int cin, int cout) {  // [cin][cout][2][2] -> [(dy*2+dx)*cout+co][ci]
      std::vector<float> w = to_host(name + ".weight"), bsrc = to_host(name + ".bias");
      if ((int)w.size() != cin * cout * 4 || (int)bsrc.size() != cout) throw std::runtime_error("convT shape " + name);
      std::vector<float> r((size_t)4 * cout * cin), rb((size_t)4 * cout);
      for (int ci = 0; ci < cin; ++ci)
        for (int co = 0; co < cout; ++co)
          for (int t = 0; t < 4; ++t) r[((size_t)t * cout + co) * cin + ci] = w[((size_t)ci * cout + co) * 4 + t];
      for (int t = 0; t < 4; ++t)
        for (int co = 0; co < cout; ++co) rb[(size_t)t * cout + co] = bsrc[co];
      LinF l;
      l.n = 4 * cout;
      l.k = cin;
      l.w = upload_f32(r);
      l.b = upload_f32(rb);
      return l;
    };
    up1_ = convT(p + "output_upscaling.0", C, C / 4);
    up2_ = convT(p + "output_upscaling.3", C / 4, C / 8);
    // hypernetwork MLPs stacked over the mask tokens: layer j weights [nt][out][in]
    const int dims[4] = {C, C, C, C / 8};
    for (int j = 0; j < 3; ++j) {
      const int in = dims[j], out = dims[j + 1];
      hyper_[j].n = out;
      hyper_[j].k = in;
      hyper_[j].w = talloc<float>((size_t)nt * out * in);
      hyper_[j].b = talloc<float>((size_t)nt * out);
      for (int t = 0; t < nt; ++t) {
        const std::string hp = p + "output_hypernetworks_mlps." + std::to_string(t) + ".layers." + std::to_string(j);
        if (raw(hp + ".weight").numel() != (int64_t)out * in) throw std::runtime_error("hyper mlp shape " + hp);
        HIP_TRY(hipMemcpy(hyper_[j].w + (size_t)t * out * in, raw(hp + ".weight").p, (size_t)out * in * 4,
                          hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemcpy(hyper_[j].b + (size_t)t * out, raw(hp + ".bias").p, (size_t)out * 4, hipMemcpyDeviceToDevice));
      }
    }
    const int idims[4] = {C, C, C, nt};
    for (int j = 0; j < 3; ++j) {
      const std::string ip = p + "iou_prediction_head.layers." + std::to_string(j);
      if (idims[j + 1] % 1) {}
      iou_head_[j] = pack_linear_f32(ip + ".weight", ip + ".bias", idims[j + 1], idims[j]);
    }
    const int n = c.max_seg, NQ = nt + 2;
    m_tokens_ = talloc<float>((size_t)n * NQ * C);
    m_q_ = talloc<float>((size_t)n * NQ * C);
    m_qp_ = talloc<float>((size_t)n * NQ * C);
    m_qh_ = talloc<float>((size_t)n * NQ * C);
    m_kh_ = talloc<float>((size_t)n * NQ * C);
    m_vh_ = talloc<float>((size_t)n * NQ * C);
    m_att_ = talloc<float>((size_t)n * NQ * C);
    m_tmp_ = talloc<float>((size_t)n * NQ * C);
    m_mlp_ = talloc<float>((size_t)n * NQ * c.dec_mlp);
    m_src_ = talloc<float>((size_t)NK * C);
    m_keys_ = talloc<float>((size_t)n * NK * C);
    m_kp_ = talloc<float>((size_t)n * NK * C);
    m_bigq_ = talloc<float>((size_t)n * NK * C / 2);
    m_bigk_ = talloc<float>((size_t)n * NK * C / 2);
    m_bigv_ = talloc<float>((size_t)n * NK * C / 2);
    m_bigatt_ = talloc<float>((size_t)n * NK * C / 2);
    m_up0_ = talloc<float>((size_t)n * NK * C);              // ConvT1 GEMM out [n*NK, 4*C/4]
    m_up1_ = talloc<float>((size_t)n * 4 * NK * (C / 4));    // [n*(2g)^2, C/4]
    m_up2_ = talloc<float>((size_t)n * 4 * NK * (C / 2));    // ConvT2 GEMM out [n*4NK, 4*C/8]
    m_hy0_ = talloc<float>((size_t)nt * n * C);
    m_hy1_ = talloc<float>((size_t)nt * n * C);
    m_hyper_ = talloc<float>((size_t)n * nt * (C / 8));
    m_masks_ = talloc<float>((size_t)n * nt * 16 * NK);
    m_iou_ = talloc<float>((size_t)n * nt);
  }
  // ================= ImageBind audio trunk =================
  if (c.aud_blocks > 0) {
    const std::string ap = "model.audio_encoder.";
    const int D = c.aud_dim, k = c.aud_kernel, st = c.aud_stride;
    if (D % c.aud_heads || k * k % 64 || c.aud_clips < 1) throw std::runtime_error("audio trunk: unsupported shape");
    const int gh = (c.aud_mel - k) / st + 1, gw = (c.aud_len - k) / st + 1;
    aud_np_ = gh * gw;
    aud_rows_ = aud_np_ + 2;  // [CLS] + patches + the add_bias_kv row
    const std::string pp = ap + "modality_preprocessors.audio.";
    aud_stem_ = pack_linear(pp + "rgbt_stem.proj.weight", "", D, k * k, 64);
    aud_stem_ln_ = affine(pp + "rgbt_stem.norm_layer");
    aud_cls_ = own_f32(pp + "cls_token");
    aud_pos_ = own_f32(pp + "pos_embedding_helper.pos_embed");
    if (raw(pp + "pos_embedding_helper.pos_embed").numel() != (int64_t)(aud_np_ + 1) * D)
      throw std::runtime_error("audio pos_embed does not match the mel / kernel / stride geometry");
    aud_blocks_.resize(c.aud_blocks);
    for (int i = 0; i < c.aud_blocks; ++i) {
      const std::string bp = ap + "modality_trunks.audio.blocks." + std::to_string(i) + ".";
      AudBlock& Bk = aud_blocks_[i];
      Bk.ln1 = affine(bp + "norm_1");
      Bk.ln2 = affine(bp + "norm_2");
      Bk.qkv = pack_linear(bp + "attn.in_proj_weight", bp + "attn.in_proj_bias", 3 * D, D);
      Bk.out = pack_linear(bp + "attn.out_proj.weight", bp + "attn.out_proj.bias", D, D);
      Bk.bias_k = own_f32(bp + "attn.bias_k");
      Bk.bias_v = own_f32(bp + "attn.bias_v");
      Bk.fc1 = pack_linear(bp + "mlp.fc1.weight", bp + "mlp.fc1.bias", 4 * D, D);
      Bk.fc2 = pack_linear(bp + "mlp.fc2.weight", bp + "mlp.fc2.bias", D, 4 * D);
    }
    aud_head_ln_ = affine(ap + "modality_heads.audio.0");
    aud_head_ = pack_linear(ap + "modality_heads.audio.2.weight", "", c.audio_dim, D);
    const std::vector<float> ls = to_host(ap + "modality_postprocessors.audio.1.log_logit_scale");
    aud_scale_ = std::min(expf(ls.at(0)), 100.f);  // LearnableLogitScaling: clip(exp(log_scale), max = 100)
    const int NC = c.aud_clips * MB;
    const size_t Rr = (size_t)NC * aud_rows_;
    if (SPT && D % 64) throw std::runtime_error("parity16: aud_dim must be a multiple of 64");
    a_col_ = aalloc<T>((size_t)NC * aud_np_, k * k);
    a_patch_ = talloc<float>((size_t)NC * aud_np_ * D);
    a_x_ = talloc<float>(Rr * D);
    a_h_ = aalloc<T>(Rr, D);
    a_qkv_ = talloc<Q>(Rr * 3 * D);
    a_att_ = aalloc<T>(Rr, D);
    HIP_TRY(hipMemset(a_att_, 0, Rr * D * sizeof(T)));  // the spare rows are never written by the attention
    a_mlp_ = aalloc<T>(Rr, 4 * D);
    a_clsrow_ = aalloc<T>((size_t)NC, D);
    a_emb_ = talloc<float>((size_t)NC * c.audio_dim);
    std::vector<int> rows(NC);
    for (int i = 0; i < NC; ++i) rows[i] = i * aud_rows_ + aud_np_ + 1;
    aud_kvrow_ = talloc<int>(NC);
    HIP_TRY(hipMemcpy(aud_kvrow_, rows.data(), NC * sizeof(int), hipMemcpyHostToDevice));
  }
  {
    int least = 0, greatest = 0;  // the side stream yields to the caller's stream
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(hipStreamCreateWithPriority(&s2_, hipStreamNonBlocking, least));
  }
  const unsigned ev_flags = getenv("ANYREF_JOIN_TIMING") ? hipEventDefault : hipEventDisableTiming;
  HIP_TRY(hipEventCreateWithFlags(&ev_fork_, ev_flags));
  HIP_TRY(hipEventCreateWithFlags(&ev_sam_, ev_flags));
  HIP_TRY(hipDeviceSynchronize());
  drop_raw();
  finalized_ = true;
}

template <typename T, typename TS>
void Model<T, TS>::fork_sam(hipStream_t s, const float* sam_images, int B, bool fed) {
  if (!overlap_) return;  // encoder then runs on `s` inside run_tail
  HIP_TRY(hipEventRecord(ev_fork_, s));
  HIP_TRY(hipStreamWaitEvent(s2_, ev_fork_, 0));
  sam_forked_ = true;
  sam_enq_done_ = false;
  sam_next_blk_ = 0;
  sam_img_ = sam_images;
  sam_B_ = B;
  early_pending_.clear();
  if (!fed) sam_feed((int)sam_blocks_.size(), false);
}
// queue encoder blocks [sam_next_blk_, upto) on the side stream (block 0 brings the patch embedding, the last one the neck)
template <typename T, typename TS>
void Model<T, TS>::sam_feed(int upto, bool capped) {
  const int nblk = (int)sam_blocks_.size();
  if (!sam_forked_ || sam_enq_done_) return;
  upto = std::min(upto, nblk);
  if (upto <= sam_next_blk_) return;
  cap_wg_ = capped ? side_cap_now_ : 0;
  try {
    sam_encoder(s2_, sam_img_, sam_B_, sam_emb_, sam_next_blk_, upto);
  } catch (...) {
    cap_wg_ = 0;
    throw;
  }
  cap_wg_ = 0;
  sam_next_blk_ = upto;
  if (upto == nblk) {
    HIP_TRY(hipEventRecord(ev_sam_, s2_));
    sam_enq_done_ = true;
    for (const PendingEarly& e : early_pending_) early_seg_run(e);  // the masks that waited for the last block
    early_pending_.clear();
  }
}

template <typename T, typename TS>
void Model<T, TS>::decode_step_graph(hipStream_t s, int B, bool keep_q, bool corun) {
  static const int corun_grid = getenv("ANYREF_GEMV_CORUN_GRID") ? atoi(getenv("ANYREF_GEMV_CORUN_GRID")) : 256;  // lab knob
  struct GridScope {  // the step's GEMV launches (eager or being captured) read gemv_grid_
    int& g;
    ~GridScope() { g = 0; }
  } grid_scope{gemv_grid_};
  gemv_grid_ = corun ? corun_grid : 0;
  if (!use_graphs_ || (g_prof && g_prof->on)) {  // the sampled profiler brackets kernels with events: eager
    llm_decode_step(s, B, keep_q);
    return;
  }
  // with kernel-side timestamps on, the step is a graph of its own: its GEMVs carry their stamp slots
  const bool stamped = stamp.on;
  const int key = B * 8 + (corun && corun_grid > 0 ? 4 : 0) + (keep_q ? 2 : 0) + (stamped ? 1 : 0);
  auto it = decode_graphs_.find(key);
  if (it == decode_graphs_.end()) {
    if (keep_q) ensure_q_last();
    if (!cap_stream_) HIP_TRY(hipStreamCreateWithFlags(&cap_stream_, hipStreamNonBlocking));
    // The MFMA decode path (B > 4) uses split-K and a captured graph bakes the slab pointer in: the workspace of
    // the capture stream is reserved ONCE, for max_batch, before the first capture.  A later capture of any
    // B <= max_batch then never regrows (= frees) a buffer an earlier graph still points at.
    gemm_reserve_workspace(cap_stream_, (size_t)8 * cfg.max_batch *
                                            std::max(std::max(3 * cfg.llm_dim, 2 * cfg.llm_mlp), cfg.llm_vocab) * 4);
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    HIP_TRY(hipStreamBeginCapture(cap_stream_, hipStreamCaptureModeRelaxed));
    try {
      if (stamped) stamp.graph_begin(key);
      llm_decode_step(cap_stream_, B, keep_q);
      if (stamped) stamp.graph_end(cap_stream_);
    } catch (...) {
      if (stamped) stamp.graph_abort();  // or every later eager GEMV would be booked as a slot of this dead graph
      hipStreamEndCapture(cap_stream_, &g);
      if (g) hipGraphDestroy(g);
      throw;
    }
    HIP_TRY(hipStreamEndCapture(cap_stream_, &g));
    HIP_TRY(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    HIP_TRY(hipGraphDestroy(g));
    it = decode_graphs_.emplace(key, ge).first;
  }
  HIP_TRY(hipGraphLaunch(it->second, s));
  if (stamped) stamp.graph_replayed(key);
}

template <typename T, typename TS>
void Model<T, TS>::ensure_q_last() {
  if (!q_last_) q_last_ = talloc<Q>((size_t)cfg.max_batch * cfg.llm_max_seq * cfg.llm_dim);
}

// ---------------------------------------------------------------------------------------------
// CLIP tower + projector
// ---------------------------------------------------------------------------------------------
template <typename T, typename TS>
void Model<T, TS>::clip_tower(hipStream_t s, const float* images, int B) {
  const anyref_config& c = cfg;
  const int Dc = c.clip_dim, n = clip_n_, S = n + 1, R = B * S, hd = Dc / c.clip_heads;
  launch_im2col_patch<T>(images, B, c.clip_image, c.clip_patch, c_col_, clip_kp_, s);
  gemm(s, c_col_, clip_kp_, clip_patch_, c_patch_, Dc, B * n, ACT_NONE, true);
  launch_clip_assemble(c_patch_, clip_cls_, clip_pos_, c_x_, B, n, Dc, s);
  norm(s, c_x_, Dc, clip_pre_, c_x_, Dc, R, Dc, c.clip_eps, true);
  bool h_ready = false;  // ln1 already applied by the previous layer's fc2 reduction
  for (size_t li = 0; li < clip_layers_.size(); ++li) {
    auto& L = clip_layers_[li];
    if (!h_ready) norm(s, c_x_, Dc, L.ln1, c_h_, Dc, R, Dc, c.clip_eps, false);
    gemm(s, c_h_, Dc, L.qkv, c_qkv_, 3 * Dc, R, ACT_NONE, QF32);
    AttnArgs a;
    a.Q = c_qkv_; a.K = c_qkv_ + Dc; a.V = c_qkv_ + 2 * Dc; a.O = c_att_;
    a.q_bs = a.k_bs = a.v_bs = (int64_t)S * 3 * Dc;
    a.q_rs = a.k_rs = a.v_rs = 3 * Dc;
    a.q_hs = a.k_hs = a.v_hs = hd;
    a.o_bs = (int64_t)S * Dc; a.o_rs = Dc; a.o_hs = hd;
    a.B = B; a.H = c.clip_heads; a.Sq = S; a.Sk = S; a.hd = hd;
    a.scale = 1.f / sqrtf((float)hd);
    a.o_split = SPT; a.sp16 = SPT;
    launch_attention<Q>(a, s);
    // the two LayerNorms of a block ride on the split-K reductions of the GEMMs in front of them (perf mode)
    if (!gemm(s, c_att_, Dc, L.out, c_x_, Dc, R, ACT_NONE, true, c_x_, Dc, nullptr, &L.ln2, c_h_, false, c.clip_eps))
      norm(s, c_x_, Dc, L.ln2, c_h_, Dc, R, Dc, c.clip_eps, false);
    gemm(s, c_h_, Dc, L.fc1, c_mlp_, c.clip_mlp, R, ACT_QUICK_GELU, false);
    const Affine* next = li + 1 < clip_layers_.size() ? &clip_layers_[li + 1].ln1 : nullptr;
    h_ready = gemm(s, c_mlp_, c.clip_mlp, L.fc2, c_x_, Dc, R, ACT_NONE, true, c_x_, Dc, nullptr, next, c_h_, false,
                   c.clip_eps);
  }
  for (int b = 0; b < B; ++b)  // drop CLS ("patch" feature select)
    launch_convert<T>(c_x_ + ((size_t)b * S + 1) * Dc, Dc, c_feat_ + (size_t)b * n * apad<T>(Dc), apad<T>(Dc), n, Dc, s);
  gemm(s, c_feat_, Dc, mm_proj_, img_feat_, c.llm_dim, B * n, ACT_NONE, true);
}

template <typename T, typename TS>
void Model<T, TS>::encode_images(hipStream_t s, const float* clip_images, int B, float* out, float* clip_feat) {
  HIP_TRY(hipSetDevice(device_));
  if (B > cfg.max_batch) throw std::runtime_error("batch exceeds max_batch");
  clip_tower(s, clip_images, B);
  HIP_TRY(hipMemcpyAsync(out, img_feat_, (size_t)B * clip_n_ * cfg.llm_dim * 4, hipMemcpyDeviceToDevice, s));
  if (clip_feat)
    for (int b = 0; b < B; ++b)
      HIP_TRY(hipMemcpyAsync(clip_feat + (size_t)b * clip_n_ * cfg.clip_dim,
                             c_x_ + ((size_t)b * (clip_n_ + 1) + 1) * cfg.clip_dim,
                             (size_t)clip_n_ * cfg.clip_dim * 4, hipMemcpyDeviceToDevice, s));
}

// ImageBindModel.get_audio_feature (imagebind_model.py:477-511), the embedding half: see anyref_audio_encode
template <typename T, typename TS>
void Model<T, TS>::audio_encode(hipStream_t s, const float* mel, int n, float* emb) {
  HIP_TRY(hipSetDevice(device_));
  const anyref_config& c = cfg;
  if (aud_blocks_.empty()) throw std::runtime_error("this handle has no ImageBind audio trunk (aud_blocks = 0)");
  if (n < 1 || n > c.aud_clips * c.max_batch) throw std::runtime_error("audio_encode: too many clips for max_batch");
  const int D = c.aud_dim, nh = c.aud_heads, hd = D / nh, np = aud_np_, RS = aud_rows_, R = n * RS, St = np + 1;
  // stem: conv (no bias) as im2col + GEMM, LayerNorm, [CLS] + positions (multimodal_preprocessors.py:121-157,255-271)
  launch_im2col_conv1<T>(mel, n, c.aud_mel, c.aud_len, c.aud_kernel, c.aud_stride, a_col_, s);
  gemm(s, a_col_, aud_stem_.k, aud_stem_, a_patch_, D, n * np, ACT_NONE, true);
  norm(s, a_patch_, D, aud_stem_ln_, a_patch_, D, n * np, D, 1e-5f, true);
  launch_clip_assemble(a_patch_, aud_cls_, aud_pos_, a_x_, n, np, D, s, RS);
  for (auto& Bk : aud_blocks_) {  // pre-LN blocks (transformer.py:94-170), nn.MultiheadAttention(add_bias_kv=True)
    norm(s, a_x_, D, Bk.ln1, a_h_, D, R, D, 1e-6f, false);
    gemm(s, a_h_, D, Bk.qkv, a_qkv_, 3 * D, R, ACT_NONE, QF32);
    // the appended key / value of every clip is the learned bias_k / bias_v row (not projected)
    launch_fill_rows_bias<Q>(a_qkv_ + D, 3 * D, aud_kvrow_, n, Bk.bias_k, D, s);
    launch_fill_rows_bias<Q>(a_qkv_ + 2 * D, 3 * D, aud_kvrow_, n, Bk.bias_v, D, s);
    AttnArgs a;
    a.Q = a_qkv_; a.K = a_qkv_ + D; a.V = a_qkv_ + 2 * D; a.O = a_att_;
    a.q_bs = a.k_bs = a.v_bs = (int64_t)RS * 3 * D;
    a.q_rs = a.k_rs = a.v_rs = 3 * D;
    a.q_hs = a.k_hs = a.v_hs = hd;
    a.o_bs = (int64_t)RS * D; a.o_rs = D; a.o_hs = hd;
    a.B = n; a.H = nh; a.Sq = St; a.Sk = St + 1; a.hd = hd;
    a.scale = 1.f / sqrtf((float)hd);
    a.o_split = SPT; a.sp16 = SPT;
    launch_attention<Q>(a, s);
    gemm(s, a_att_, D, Bk.out, a_x_, D, R, ACT_NONE, true, a_x_, D);
    norm(s, a_x_, D, Bk.ln2, a_h_, D, R, D, 1e-6f, false);
    gemm(s, a_h_, D, Bk.fc1, a_mlp_, 4 * D, R, ACT_GELU, false);
    gemm(s, a_mlp_, 4 * D, Bk.fc2, a_x_, D, R, ACT_NONE, true, a_x_, D);
  }
  // head: LayerNorm -> [CLS] -> Linear (no bias) -> L2-normalise x logit scale (imagebind_model.py:391-395,425-428)
  norm(s, a_x_, RS * D, aud_head_ln_, a_clsrow_, D, n, D, 1e-6f, false);
  gemm(s, a_clsrow_, D, aud_head_, a_emb_, c.audio_dim, n, ACT_NONE, true);
  launch_l2norm_scale(a_emb_, n, c.audio_dim, aud_scale_, emb, s);
}

template <typename T, typename TS>
void Model<T, TS>::project_audio(hipStream_t s, const float* audio_emb, int n, float* out) {
  HIP_TRY(hipSetDevice(device_));
  if (!has_audio_) throw std::runtime_error("model.audio_projector.* was not provided");
  if (n > cfg.max_batch * 64) throw std::runtime_error("too many audio rows");
  // stage through the LLM scratch (idle at this point)
  launch_convert<T>(audio_emb, cfg.audio_dim, l_h_, apad<T>(cfg.audio_dim), n, cfg.audio_dim, s);
  gemm(s, l_h_, cfg.audio_dim, audio_proj_, out, cfg.llm_dim, n, ACT_NONE, true);
}

// ---------------------------------------------------------------------------------------------
// LLaMA
// ---------------------------------------------------------------------------------------------
template <typename T, typename TS>
void Model<T, TS>::llm_prefill(hipStream_t s, int B, int Sp, const int* lens_dev, bool keep_q) {
  // l_x_ holds the spliced embeddings [B,Sp,H] (compact).  Fills the KV cache, hidden_all_[b, 0:Sp]
  // (post final norm) and next_dev_ (greedy token after each prompt).
  const anyref_config& c = cfg;
  const int H = c.llm_dim, F = c.llm_mlp, nh = c.llm_heads, hd = H / nh, S = c.llm_max_seq, R = B * Sp;
  const int nl = c.llm_layers;
  bool h_ready = false;  // l_h_ already holds in_norm(x) (fused into the previous layer's down_proj reduction)
  for (int i = 0; i < nl; ++i) {
    LlmLayer& L = llm_layers_[i];
    Q* kc = kcache_ + cache_layer_stride_ * i;
    Q* vc = vcache_ + cache_layer_stride_ * i;
    if (!h_ready) norm(s, l_x_, H, L.in_norm, l_h_, H, R, H, c.llm_rms_eps, false, true);
    Q* qkeep = (keep_q && i == nl - 1) ? q_last_ : nullptr;
    if (qkv_slabs_ && !L.qkv.w8 && R > 192 && R <= 320 && H % 128 == 0 && hd % 16 == 0 && !qkv_slabs_off_) {
      // one image's prompt: the projection as two K slices on whole-M tiles (256 workgroups), summed by the RoPE kernel
      GemmArgs a;
      a.A = l_h_; a.lda = H; a.W = L.qkv.w; a.ldw = L.qkv.stride(); a.M = R; a.N = 3 * H; a.K = H;
      a.slabs_out = qkv_slabs_; a.slabs = 2;
      launch_gemm<T>(a, s);
      launch_rope_cache_slabs(qkv_slabs_, qkv_slabs_ + (size_t)R * 3 * H, B, Sp, nh, hd, nullptr, lens_dev, rope_tab_, l_q_, kc,
                              vc, S, qkeep, s, QF32);
    } else {
      gemm(s, l_h_, H, L.qkv, l_qkv_, 3 * H, R, ACT_NONE, QF32);
      launch_rope_cache<Q>(l_qkv_, B, Sp, nh, hd, nullptr, lens_dev, rope_tab_, l_q_, kc, vc, S, qkeep, s);
    }
    AttnArgs a;
    a.Q = l_q_; a.K = kc; a.V = vc; a.O = l_att_;
    a.q_bs = (int64_t)Sp * H; a.q_rs = H; a.q_hs = hd;
    a.k_bs = a.v_bs = (int64_t)S * H; a.k_rs = a.v_rs = H; a.k_hs = a.v_hs = hd;
    a.o_bs = (int64_t)Sp * H; a.o_rs = H; a.o_hs = hd;
    a.B = B; a.H = nh; a.Sq = Sp; a.Sk = Sp; a.hd = hd;
    a.scale = 1.f / sqrtf((float)hd);
    a.causal = 1; a.kv_len = lens_dev; a.q_len = lens_dev;
    a.o_split = SPT; a.sp16 = SPT;
    launch_attention<Q>(a, s);
    if (!gemm(s, l_att_, H, L.o, l_x_, H, R, ACT_NONE, true, l_x_, H, nullptr, &L.post_norm, l_h_))
      norm(s, l_x_, H, L.post_norm, l_h_, H, R, H, c.llm_rms_eps, false, true);
    gemm(s, l_h_, H, L.gu, l_act_, F, R, ACT_NONE, false, nullptr, 0, nullptr, nullptr, nullptr, true);  // silu(g) * u
    h_ready = gemm(s, l_act_, F, L.down, l_x_, H, R, ACT_NONE, true, l_x_, H, nullptr,
                   i + 1 < nl ? &llm_layers_[i + 1].in_norm : nullptr, l_h_);
  }
  for (int b = 0; b < B; ++b)
    norm(s, l_x_ + (size_t)b * Sp * H, H, llm_norm_, hidden_all_ + (size_t)b * S * H, H, Sp, H, c.llm_rms_eps, true,
         true);
}

template <typename T, typename TS>
void Model<T, TS>::llm_decode_step(hipStream_t s, int B, bool keep_q) {
  // next_dev_ -> embed -> all layers at pos_dev_ -> hidden_all_[b,pos] -> logits -> next_dev_; pos += 1
  const anyref_config& c = cfg;
  const int H = c.llm_dim, F = c.llm_mlp, nh = c.llm_heads, hd = H / nh, S = c.llm_max_seq, nl = c.llm_layers;
  // d_x_ (embedding of next_dev_), rowmap_dev_ and kvlen_dev_ were written by the launch that chose next_dev_
  // (launch_argmax_next: the first-token argmax in generate(), then the tail of every step)
  // More than 4 sequences per call: the FMA GEMV would stream every weight twice (4 batch rows per pass) and
  // its VALU work grows with B, so the linears go through the MFMA GEMM (M = B rows of a 64/128-row tile,
  // split-K to fill the chip: weights are read once) with the norms / SwiGLU as in prefill.
  // (bf16 activations: up to 8 sequences stay on the GEMV -- 8 rows per pass over the q / k / v, o and gate / up weights, two passes
  //  of four over down_proj; the tiled GEMM streamed the decode rows' weights at 2 TB/s.  ANYREF_GEMV_ROWS8=0: the round-4 rule)
  static const bool rows8 = !(getenv("ANYREF_GEMV_ROWS8") && atoi(getenv("ANYREF_GEMV_ROWS8")) == 0);
  const bool mfma_decode = IS16 && B > ((std::is_same<T, bf16>::value && rows8) ? 8 : 4);
  bool h_ready = false;
  for (int i = 0; i < nl; ++i) {
    LlmLayer& L = llm_layers_[i];
    Q* kc = kcache_ + cache_layer_stride_ * i;
    Q* vc = vcache_ + cache_layer_stride_ * i;
    if (mfma_decode) {
      if (!h_ready) norm(s, d_x_, H, L.in_norm, l_h_, H, B, H, c.llm_rms_eps, false, true);
      gemm(s, l_h_, H, L.qkv, d_qkv_, 3 * H, B, ACT_NONE, true);
    } else {
      GemvArgs g;
      g.x = d_x_; g.ldx = H; g.gain = L.in_norm.g; g.eps = c.llm_rms_eps; gemv_w(g, L.qkv); g.y = d_qkv_;
      g.ldy = 3 * H; g.B = B; g.N = 3 * H; g.K = H;
      launch_gemv<T>(g, s);
    }
    Q* qk = (keep_q && i == nl - 1) ? q_last_ : nullptr;
    if (!launch_decode_attn<Q>(d_qkv_, B, nh, hd, pos_dev_, rope_tab_, kc, vc, S, 1.f / sqrtf((float)hd), d_att_, qk,
                               s)) {
      launch_rope_cache_f32<Q>(d_qkv_, B, nh, hd, pos_dev_, rope_tab_, d_q_, kc, vc, S, qk, s);
      AttnArgs a;
      a.Q = d_q_; a.K = kc; a.V = vc; a.O = d_att_; a.o_f32 = 1;
      a.q_bs = H; a.q_rs = H; a.q_hs = hd;
      a.k_bs = a.v_bs = (int64_t)S * H; a.k_rs = a.v_rs = H; a.k_hs = a.v_hs = hd;
      a.o_bs = H; a.o_rs = H; a.o_hs = hd;
      a.B = B; a.H = nh; a.Sq = 1; a.Sk = S; a.hd = hd;
      a.scale = 1.f / sqrtf((float)hd);
      a.kv_len = kvlen_dev_;
      launch_attention<Q>(a, s);
    }
    if (mfma_decode) {
      launch_convert<T>(d_att_, H, l_att_, apad<T>(H), B, H, s);
      if (!gemm(s, l_att_, H, L.o, d_x_, H, B, ACT_NONE, true, d_x_, H, nullptr, &L.post_norm, l_h_))
        norm(s, d_x_, H, L.post_norm, l_h_, H, B, H, c.llm_rms_eps, false, true);
      gemm(s, l_h_, H, L.gu, l_act_, F, B, ACT_NONE, false, nullptr, 0, nullptr, nullptr, nullptr, true);
      h_ready = gemm(s, l_act_, F, L.down, d_x_, H, B, ACT_NONE, true, d_x_, H, nullptr,
                     i + 1 < nl ? &llm_layers_[i + 1].in_norm : nullptr, l_h_);
      continue;
    }
    GemvArgs o;
    o.x = d_att_; o.ldx = H; gemv_w(o, L.o); o.y = d_x_; o.resid = d_x_; o.ldy = H; o.B = B; o.N = H; o.K = H;
    launch_gemv<T>(o, s);
    GemvArgs m;
    m.x = d_x_; m.ldx = H; m.gain = L.post_norm.g; m.eps = c.llm_rms_eps; gemv_w(m, L.gu, 0); gemv_w2(m, L.gu, 1); m.ldw = 2 * L.gu.stride(); m.ws_stride = 2;
    m.y = d_act_; m.ldy = F; m.B = B; m.N = F; m.K = H;
    launch_gemv<T>(m, s);
    GemvArgs d;
    d.x = d_act_; d.ldx = F; gemv_w(d, L.down); d.y = d_x_; d.resid = d_x_; d.ldy = H; d.B = B; d.N = H; d.K = F;
    launch_gemv<T>(d, s);
  }
  // final RMSNorm inside the lm_head GEMV's input stage; its f32 rows are hidden_states[-1] of this position
  GemvArgs h;
  h.x = d_x_; h.ldx = H; h.gain = llm_norm_.g; h.eps = c.llm_rms_eps; gemv_w(h, lm_head_); h.y = l_logits_;
  h.ldy = c.llm_vocab; h.B = B; h.N = c.llm_vocab; h.K = H;
  h.xn_out = hidden_all_; h.xn_row_map = rowmap_dev_; h.xn_ld = H;
  launch_gemv<T>(h, s);
  // argmax, pos += 1, and the next step's embedding row / cache row index / key count
  launch_argmax_next(l_logits_, B, c.llm_vocab, c.llm_vocab, next_dev_, pos_dev_, emb_table_, sizeof(W) == 2, H, S, d_x_,
                     rowmap_dev_, kvlen_dev_, s);
}

template <typename T, typename TS>
void Model<T, TS>::llm_forward(hipStream_t s, const float* embeds, const int32_t* lens, int B, int Sn, float* hidden,
                           float* logits, const int32_t* attn_q, float* attn_row) {
  HIP_TRY(hipSetDevice(device_));
  const anyref_config& c = cfg;
  if (B > c.max_batch || Sn > c.llm_max_seq) throw std::runtime_error("llm_forward: batch/seq exceed the model's limits");
  const int H = c.llm_dim;
  HIP_TRY(hipMemcpyAsync(l_x_, embeds, (size_t)B * Sn * H * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(lens_dev_, lens, B * 4, hipMemcpyHostToDevice, s));
  const bool keep_q = attn_q != nullptr;
  if (keep_q) ensure_q_last();
  llm_prefill(s, B, Sn, lens_dev_, keep_q);
  for (int b = 0; b < B; ++b)
    HIP_TRY(hipMemcpyAsync(hidden + (size_t)b * Sn * H, hidden_all_ + (size_t)b * c.llm_max_seq * H,
                           (size_t)Sn * H * 4, hipMemcpyDeviceToDevice, s));
  if (logits) {
    launch_convert<T>(hidden, H, l_h_, apad<T>(H), B * Sn, H, s);
    gemm(s, l_h_, H, lm_head_, logits, c.llm_vocab, B * Sn, ACT_NONE, true);
  }
  if (attn_q) {
    const int nh = c.llm_heads, hd = H / nh, S = c.llm_max_seq;
    std::vector<int> kl(B);
    for (int b = 0; b < B; ++b) kl[b] = attn_q[b] + 1;
    HIP_TRY(hipMemcpyAsync(kvlen_dev_, kl.data(), B * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    Q* kc = kcache_ + cache_layer_stride_ * (c.llm_layers - 1);
    for (int b = 0; b < B; ++b)
      launch_attn_row_mean<Q>(q_last_ + ((size_t)b * S + attn_q[b]) * H, 0, hd, kc + (size_t)b * S * H, 0, H, hd,
                              kvlen_dev_ + b, 1, nh, hd, 1.f / sqrtf((float)hd), attn_row + (size_t)b * Sn, Sn, s);
  }
}

// ---------------------------------------------------------------------------------------------
// SAM image encoder
// ---------------------------------------------------------------------------------------------
template <typename T, typename TS>
void Model<T, TS>::sam_encoder(hipStream_t s, const float* images, int B, float* out, int blk0, int blk1) {
  const anyref_config& c = cfg;
  const int D = c.sam_dim, g = sam_g_, NT = g * g, RT = B * NT, ws = c.sam_window, nh = c.sam_heads, hd = D / nh;
  const int C = c.sam_out_chans, WR = sam_wrows_, nW = sam_nw_ * sam_nw_;
  const int nblk = (int)sam_blocks_.size();
  const bool to_end = blk1 < 0 || blk1 >= nblk;
  if (to_end) blk1 = nblk;
  // Lab knob (ANYREF_SAM_PER_IMAGE=1, default off): several images as one image at a time through all blocks.  The premise -- a
  // batch's activations (4 images: 84 MB residual + 126 MB q / k / v + 168 MB MLP per block) fall out of the caches -- does not
  // hold: scratch/sam_batch_split.py, SAM-H alone, B = 4: 31.2 ms in one call (7.8 per image) vs 34.3 ms as four (8.6); only
  // B = 2 loses in one call (18.3 vs 17.2 ms: tile rounds at M = 8192); parity16 59.7 vs 65.5 ms.
  static const bool per_image = getenv("ANYREF_SAM_PER_IMAGE") && atoi(getenv("ANYREF_SAM_PER_IMAGE")) != 0;
  if (B > 1 && per_image && blk0 == 0 && to_end) {
    for (int b = 0; b < B; ++b)
      sam_encoder(s, images + (size_t)b * 3 * c.sam_img * c.sam_img, 1, out + (size_t)b * NT * C, 0, -1);
    return;
  }
  if (blk0 == 0) {
    launch_im2col_patch<TS>(images, B, c.sam_img, c.sam_patch, s_col_, sam_patch_.k, s);
    GemmArgs a;
    a.A = s_col_; a.lda = sam_patch_.k; a.W = sam_patch_.w; a.ldw = sam_patch_.k; a.bias = sam_patch_.b;
    a.C = s_x_; a.ldc = D; a.M = NT; a.N = D; a.K = sam_patch_.k; a.c_f32 = 1;
    a.resid = sam_pos_; a.ldr = D;  // + absolute position embedding, shared by every image
    a.batch = B; a.sA = (int64_t)NT * sam_patch_.k; a.sC = (int64_t)NT * D; a.sR = 0;
    a.max_wg = cap_wg_;  // (honoured at batch 1, where the CU share is in use)
    launch_gemm<TS>(a, s);
  }
  for (int bi = blk0; bi < blk1; ++bi) {
    auto& L = sam_blocks_[bi];
    AttnArgs a;
    a.max_wg = cap_wg_;
    a.q_hs = a.k_hs = a.v_hs = hd; a.o_hs = hd;
    a.q_rs = a.k_rs = a.v_rs = 3 * D; a.o_rs = D;
    a.H = nh; a.hd = hd; a.scale = 1.f / sqrtf((float)hd);
    a.Q = s_qkv_; a.K = s_qkv_ + D; a.V = s_qkv_ + 2 * D; a.O = s_att_;
    a.o_split = SPS; a.sp16 = SPS;
    // decomposed rel-pos bias: P = q . [rel_pos_h | rel_pos_w]^T for every head in ONE batched MFMA GEMM
    // (batch = heads, A = the q columns of the fused qkv buffer); the attention kernel applies the shift.
    auto rel_gemm = [&](int rows) {
      GemmArgs r;
      r.A = s_qkv_; r.lda = 3 * D; r.sA = hd; r.W = L.rel.w; r.ldw = L.rel.k; r.sW = 0;
      r.C = s_relh_; r.ldc = L.rel.n; r.sC = (int64_t)rows * L.rel.n; r.M = rows; r.N = L.rel.n; r.K = L.rel.k;
      r.c_f32 = 1; r.batch = nh;
      // (exempt from the side-stream CU share: a batched launch has no capped form; ~1024 short workgroups, 4 launches per image)
      launch_gemm<QS>(r, s);
      a.rel_p = s_relh_; a.rel_ld = L.rel.n; a.rel_hs = (int64_t)rows * L.rel.n;
    };
    if (L.global) {
      norm<TS>(s, s_x_, D, L.ln1, s_hglob_, D, RT, D, 1e-6f, false);
      gemm(s, s_hglob_, D, L.qkv, s_qkv_, 3 * D, RT, ACT_NONE, QF32S);
      rel_gemm(RT);
      a.q_bs = a.k_bs = a.v_bs = (int64_t)NT * 3 * D; a.o_bs = (int64_t)NT * D;
      a.B = B; a.Sq = NT; a.Sk = NT; a.kh = g; a.kw = g;
      launch_attention<QS>(a, s);
      gemm(s, s_att_, D, L.proj, s_x_, D, RT, ACT_NONE, true, s_x_, D);
    } else {
      const int RW = B * WR, S2 = ws * ws;
      // qkv over the REAL tokens only, scattered into the window layout by the GEMM epilogue; the pad rows of
      // a window (zero input after norm1, image_encoder.py:175-179) get exactly the bias.  16 % fewer GEMM
      // rows at SAM-H (4900 -> 4096 per image), which also makes the 256^2 tile fit (240 tiles).
      bool filled = false;
      {  // norm1, with the pad rows of the window layout (q/k/v = the bias) written by extra workgroups of the same launch
        NormArgs na;
        na.x = s_x_; na.ldx = D; na.gain = L.ln1.g; na.bias = L.ln1.b; na.y = s_hglob_; na.ldy = D; na.M = RT; na.D = D;
        na.eps = 1e-6f;
        na.fill_dst = s_qkv_; na.fill_ld = 3 * D; na.fill_rows = pad_rows_; na.fill_n = n_pad_rows_ * B; na.fill_bias = L.qkv.b;
        na.fill_N = 3 * D; na.fill_done = &filled;
        launch_norm<TS>(na, s);
      }
      gemm(s, s_hglob_, D, L.qkv, s_qkv_, 3 * D, RT, ACT_NONE, QF32S, nullptr, 0, tok2win_);
      if (!filled) launch_fill_rows_bias<QS>(s_qkv_, 3 * D, pad_rows_, n_pad_rows_ * B, L.qkv.b, 3 * D, s);
      if (attention_takes_rel_tables((int)sizeof(QS), hd, S2, S2, ws, ws, SPS)) {  // (16-bit towers: sizeof(QS) == sizeof(TS))
        // window bias straight from the tables inside the attention kernel (rows 0.. = rel_pos_h, Np.. = rel_pos_w)
        a.rel_tab_h = L.rel.w; a.rel_tab_w = L.rel.w + (size_t)(L.rel.n / 2) * L.rel.k; a.rel_tab_ld = L.rel.k;
      } else {
        rel_gemm(RW);
      }
      a.q_bs = a.k_bs = a.v_bs = (int64_t)S2 * 3 * D; a.o_bs = (int64_t)S2 * D;
      a.B = B * nW; a.Sq = S2; a.Sk = S2; a.kh = ws; a.kw = ws;
      launch_attention<QS>(a, s);
      // proj over the REAL tokens only, gathered from the window layout by the GEMM's A-row map (the pad rows' outputs
      // were dropped by the epilogue before: 4900 -> 4096 rows per image, as for qkv)
      if (IS16S && D % 64 == 0)
        gemm(s, s_att_, D, L.proj, s_x_, D, RT, ACT_NONE, true, s_x_, D, nullptr, nullptr, nullptr, false, -1.f, tok2win_);
      else
        gemm(s, s_att_, D, L.proj, s_x_, D, RW, ACT_NONE, true, s_x_, D, win2tok_);
    }
    norm<TS>(s, s_x_, D, L.ln2, s_hglob_, D, RT, D, 1e-6f, false);
    gemm(s, s_hglob_, D, L.lin1, s_mlp_, c.sam_mlp_ratio * D, RT, ACT_GELU, false);
    gemm(s, s_mlp_, c.sam_mlp_ratio * D, L.lin2, s_x_, D, RT, ACT_NONE, true, s_x_, D);
  }
  if (!to_end) return;
  // neck: 1x1 conv -> LN2d -> 3x3 conv -> LN2d (channels-last tokens; fp32 LayerNorm as the
  // reference forces under fp16, image_encoder.py:119-122)
  launch_convert<TS>(s_x_, D, s_hglob_, apad<TS>(D), RT, D, s);
  gemm(s, s_hglob_, D, neck0_, s_n0_, C, RT, ACT_NONE, true);
  norm<TS>(s, s_n0_, C, neck1_, s_n1_, C, RT, C, 1e-6f, false);
  launch_im2col_3x3<TS>(s_n1_, B, g, C, s_col3_, s);
  gemm(s, s_col3_, 9 * C, neck2_, s_n2_, C, RT, ACT_NONE, true);
  norm<TS>(s, s_n2_, C, neck3_, out, C, RT, C, 1e-6f, true);
}

template <typename T, typename TS>
void Model<T, TS>::sam_encode(hipStream_t s, const float* sam_images, int B, float* out) {
  HIP_TRY(hipSetDevice(device_));
  if (B > cfg.max_batch) throw std::runtime_error("batch exceeds max_batch");
  sam_encoder(s, sam_images, B, out);
}

// ---------------------------------------------------------------------------------------------
// prompt encoder (text) + mask decoder, all f32
// ---------------------------------------------------------------------------------------------
template <typename T, typename TS>
void Model<T, TS>::mask_decoder(hipStream_t s, const float* image_emb, const float* pred_emb, int n, float* masks4,
                            float* iou) {
  const anyref_config& c = cfg;
  const int C = c.sam_out_chans, g = sam_g_, NK = g * g, nt = c.num_mask_tokens, NQ = nt + 2, Ci = C / 2;
  const int nh = c.dec_heads;
  // tokens = [iou, mask x nt, text prompt]; src = image embedding + no_mask dense embedding
  launch_build_tokens(out_tokens_, nt + 1, pred_emb, n, C, m_tokens_, s);
  launch_add_vec(image_emb, no_mask_, m_src_, NK, C, s);
  for (int i = 0; i < n; ++i)
    HIP_TRY(hipMemcpyAsync(m_keys_ + (size_t)i * NK * C, m_src_, (size_t)NK * C * 4, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(m_q_, m_tokens_, (size_t)n * NQ * C * 4, hipMemcpyDeviceToDevice, s));

  auto attn = [&](const float* Q, const float* K, const float* V, float* O, int Sq, int Sk, int internal) {
    AttnArgs a;
    const int hd = internal / nh;
    a.Q = Q; a.K = K; a.V = V; a.O = O;
    a.q_bs = (int64_t)Sq * internal; a.q_rs = internal; a.q_hs = hd;
    a.k_bs = a.v_bs = (int64_t)Sk * internal; a.k_rs = a.v_rs = internal; a.k_hs = a.v_hs = hd;
    a.o_bs = (int64_t)Sq * internal; a.o_rs = internal; a.o_hs = hd;
    a.B = n; a.H = nh; a.Sq = Sq; a.Sk = Sk; a.hd = hd; a.scale = 1.f / sqrtf((float)hd);
    launch_attention<float>(a, s);
  };
  auto lnf = [&](float* x, const Affine& af, int M) {
    NormArgs a;
    a.x = x; a.ldx = C; a.gain = af.g; a.bias = af.b; a.y = x; a.ldy = C; a.M = M; a.D = C; a.eps = 1e-5f; a.y_f32 = 1;
    launch_norm<float>(a, s);
  };
  // token -> image attention with weights `A`; queries += out
  auto token_to_image = [&](const DecAttn& A) {
    launch_add_rows(m_q_, m_tokens_, n * NQ, m_qp_, n * NQ, C, s);   // q = queries + query_pe
    launch_add_rows(m_keys_, dense_pe_, NK, m_kp_, n * NK, C, s);    // k = keys + key_pe
    gemmf(s, m_qp_, C, A.q, m_qh_, Ci, n * NQ, ACT_NONE);
    gemmf(s, m_kp_, C, A.k, m_bigk_, Ci, n * NK, ACT_NONE);
    gemmf(s, m_keys_, C, A.v, m_bigv_, Ci, n * NK, ACT_NONE);
    attn(m_qh_, m_bigk_, m_bigv_, m_att_, NQ, NK, Ci);
    gemmf(s, m_att_, Ci, A.o, m_q_, C, n * NQ, ACT_NONE, m_q_, C);
  };

  for (int i = 0; i < c.dec_depth; ++i) {
    DecLayer& L = dec_layers_[i];
    // (1) self attention of the tokens (layer 0: no PE, output replaces the queries; transformer.py:153-160)
    const float* qin = m_q_;
    if (i > 0) {
      launch_add_rows(m_q_, m_tokens_, n * NQ, m_qp_, n * NQ, C, s);
      qin = m_qp_;
    }
    gemmf(s, qin, C, L.self.q, m_qh_, C, n * NQ, ACT_NONE);
    gemmf(s, qin, C, L.self.k, m_kh_, C, n * NQ, ACT_NONE);
    gemmf(s, m_q_, C, L.self.v, m_vh_, C, n * NQ, ACT_NONE);
    attn(m_qh_, m_kh_, m_vh_, m_att_, NQ, NQ, C);
    if (i == 0)
      gemmf(s, m_att_, C, L.self.o, m_q_, C, n * NQ, ACT_NONE);
    else
      gemmf(s, m_att_, C, L.self.o, m_q_, C, n * NQ, ACT_NONE, m_q_, C);
    lnf(m_q_, L.n1, n * NQ);
    // (2) tokens attend to the image
    token_to_image(L.t2i);
    lnf(m_q_, L.n2, n * NQ);
    // (3) MLP on the tokens
    gemmf(s, m_q_, C, L.lin1, m_mlp_, c.dec_mlp, n * NQ, ACT_RELU);
    gemmf(s, m_mlp_, c.dec_mlp, L.lin2, m_q_, C, n * NQ, ACT_NONE, m_q_, C);
    lnf(m_q_, L.n3, n * NQ);
    // (4) image attends to the tokens: q = keys + key_pe, k = queries + query_pe, v = queries
    launch_add_rows(m_q_, m_tokens_, n * NQ, m_qp_, n * NQ, C, s);
    launch_add_rows(m_keys_, dense_pe_, NK, m_kp_, n * NK, C, s);
    gemmf(s, m_kp_, C, L.i2t.q, m_bigq_, Ci, n * NK, ACT_NONE);
    gemmf(s, m_qp_, C, L.i2t.k, m_kh_, Ci, n * NQ, ACT_NONE);
    gemmf(s, m_q_, C, L.i2t.v, m_vh_, Ci, n * NQ, ACT_NONE);
    attn(m_bigq_, m_kh_, m_vh_, m_bigatt_, NK, NQ, Ci);
    gemmf(s, m_bigatt_, Ci, L.i2t.o, m_keys_, C, n * NK, ACT_NONE, m_keys_, C);
    lnf(m_keys_, L.n4, n * NK);
  }
  token_to_image(dec_final_);
  lnf(m_q_, dec_norm_final_, n * NQ);

  // ---- upscaler: ConvT(k2s2) = GEMM + un-shuffle, LayerNorm2d + GELU, ConvT, GELU, hyper product ----
  gemmf(s, m_keys_, C, up1_, m_up0_, C, n * NK, ACT_NONE);
  launch_upscale1<float>(m_up0_, n, g, C / 4, up_ln_.g, up_ln_.b, 1e-6f, m_up1_, s);
  gemmf(s, m_up1_, C / 4, up2_, m_up2_, C / 2, n * 4 * NK, ACT_NONE);
  // hypernetwork MLPs, batched over the mask tokens (A = hs[:, 1+t, :])
  for (int j = 0; j < 3; ++j) {
    GemmArgs a;
    const int in = hyper_[j].k, outn = hyper_[j].n;
    if (j == 0) {
      a.A = m_q_ + C; a.lda = NQ * C; a.sA = C;
    } else {
      a.A = j == 1 ? m_hy0_ : m_hy1_; a.lda = in; a.sA = (int64_t)n * in;
    }
    a.W = hyper_[j].w; a.ldw = in; a.sW = (int64_t)outn * in;
    a.bias = hyper_[j].b; a.sBias = outn;
    if (j < 2) {
      a.C = j == 0 ? m_hy0_ : m_hy1_; a.ldc = outn; a.sC = (int64_t)n * outn; a.act = ACT_RELU;
    } else {
      a.C = m_hyper_; a.ldc = nt * outn; a.sC = outn;  // -> [n][nt][C/8]
    }
    a.M = n; a.N = outn; a.K = in; a.c_f32 = 1; a.batch = nt;
    launch_gemm<float>(a, s);
  }
  launch_upscale2_masks(m_up2_, m_hyper_, n, nt, 2 * g, C / 8, masks4 ? masks4 : m_masks_, s);
  if (iou) {
    gemmf(s, m_q_, NQ * C, iou_head_[0], m_hy0_, C, n, ACT_RELU);
    gemmf(s, m_hy0_, C, iou_head_[1], m_hy1_, C, n, ACT_RELU);
    // last layer has N = nt (not a multiple of 4 in K? K = C, fine)
    gemmf(s, m_hy1_, C, iou_head_[2], iou, nt, n, ACT_NONE);
  }
}

template <typename T, typename TS>
void Model<T, TS>::mask_decode(hipStream_t s, const float* image_emb, const float* pred_emb, int n, float* masks4,
                           float* iou, const int32_t* resized_hw, const int32_t* orig_hw, float* out_masks) {
  HIP_TRY(hipSetDevice(device_));
  if (n > cfg.max_seg) throw std::runtime_error("more prompts than max_seg");
  if (n <= 0) return;
  mask_decoder(s, image_emb, pred_emb, n, masks4, iou);
  if (out_masks) {
    const int L = 4 * sam_g_;
    const float* m = masks4 ? masks4 : m_masks_;
    launch_postprocess(m, (int64_t)cfg.num_mask_tokens * L * L, n, L, L, cfg.sam_img, resized_hw[0], resized_hw[1],
                       orig_hw[0], orig_hw[1], out_masks, s);
  }
}

// ---------------------------------------------------------------------------------------------
// generate / forward
// ---------------------------------------------------------------------------------------------
template <typename T, typename TS>
int Model<T, TS>::splice_inputs(hipStream_t s, const int64_t* input_ids, const int32_t* lens, int B, int Lmax,
                            const float* extra_embeds, const int32_t* extra_slots, int n_extra,
                            std::vector<int>& slen, std::vector<int>& img_pos) {
  const anyref_config& c = cfg;
  const int n_img = clip_n_, H = c.llm_dim;
  slen.assign(B, 0);
  img_pos.assign(B, -1);
  int Sp = 0;
  for (int b = 0; b < B; ++b) {
    if (lens[b] <= 0 || lens[b] > Lmax) throw std::runtime_error("bad prompt length");
    for (int i = 0; i < lens[b]; ++i)
      if (input_ids[(size_t)b * Lmax + i] == -200) {
        if (img_pos[b] >= 0) throw std::runtime_error("more than one image placeholder in a prompt");
        img_pos[b] = i;
      }
    slen[b] = lens[b] + (img_pos[b] >= 0 ? n_img - 1 : 0);
    Sp = std::max(Sp, slen[b]);
  }
  if (Sp > c.llm_max_seq) throw std::runtime_error("prompt longer than llm_max_seq");
  HIP_TRY(hipMemcpyAsync(ids_dev_, input_ids, (size_t)B * Lmax * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(lens_dev_, lens, B * 4, hipMemcpyHostToDevice, s));
  launch_embed_splice(ids_dev_, lens_dev_, B, Lmax, emb_table_, sizeof(W) == 2, c.llm_vocab, img_feat_, n_img, l_x_, Sp,
                      H, slen_dev_, s);
  if (n_extra > 0) {
    if (n_extra > (int)cfg.max_batch * std::max(cfg.max_seg, 64)) throw std::runtime_error("too many extra slots");
    {
    StageScope stage_scope(this, ST_EXTRA, s);
    int *eb = stage_extra_, *ep = stage_extra_ + n_extra;  // pinned: the copies below need no host wait
    for (int i = 0; i < n_extra; ++i) {
      const int b = extra_slots[2 * i], p = extra_slots[2 * i + 1];
      if (b < 0 || b >= B || p < 0 || p >= lens[b]) throw std::runtime_error("extra slot out of range");
      eb[i] = b;
      ep[i] = (img_pos[b] >= 0 && p > img_pos[b]) ? p + n_img - 1 : p;
    }
    HIP_TRY(hipMemcpyAsync(idx_a_, eb, n_extra * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(idx_b_, ep, n_extra * 4, hipMemcpyHostToDevice, s));
    }
    launch_scatter_rows(extra_embeds, idx_a_, idx_b_, n_extra, l_x_, Sp, H, s);
  }
  return Sp;
}

template <typename T, typename TS>
void Model<T, TS>::join_sam(hipStream_t s) {
  static const bool timing = getenv("ANYREF_JOIN_TIMING") != nullptr;  // diagnostic: which stream the join waits for
  if (sam_forked_ && timing) {
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, s));
    HIP_TRY(hipStreamWaitEvent(s, ev_sam_, 0));
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    float wait_ms = 0.f, sam_ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&wait_ms, e0, e1));
    HIP_TRY(hipEventElapsedTime(&sam_ms, ev_fork_, ev_sam_));
    fprintf(stderr, "[anyref] join: main stream waited %.3f ms for the SAM encoder (fork -> done %.3f ms)\n", wait_ms, sam_ms);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  if (sam_forked_ && !sam_enq_done_) HIP_TRY(hipEventRecord(ev_sam_, s2_));  // a call left half-way: wait for what was queued
  if (sam_forked_) HIP_TRY(hipStreamWaitEvent(s, ev_sam_, 0));  // everything after this on `s` sees the image embeddings
  sam_forked_ = false;
  early_pending_.clear();
}
namespace {
// generate / forward fork the encoder early; whatever path leaves them (a throw included) joins it
template <typename M>
struct SamJoinGuard {
  M* m;
  hipStream_t s;
  ~SamJoinGuard() {
    try {
      m->join_sam_public(s);
    } catch (...) {
    }
  }
};
}  // namespace

// A generated [SEG] needs nothing that comes after it: its hidden row is final when the token is read, and the image
// embedding is on the side stream anyway.  So its hand-off MLP, mask decoder and postprocess are queued on that stream
// (behind the encoder) the moment the host sees the token, and run under the remaining decode steps instead of after
// the loop (the reference always emits at least the EOS after a [SEG]; ~1 ms per image at batch 1).
template <typename T, typename TS>
void Model<T, TS>::early_seg(hipEvent_t hidden_ready, int hidden_row, const int32_t* resized_hw, const int32_t* orig_hw,
                         float* out_masks, int64_t out_masks_cap, float* out_low) {
  const PendingEarly e{hidden_ready, hidden_row, resized_hw, orig_hw, out_masks, out_masks_cap, out_low};
  if (!sam_enq_done_) {  // blocks still to be fed: the mask is queued behind the last of them (sam_feed)
    if ((int)early_pending_.size() + early_done_ >= cfg.max_seg) early_stop_ = true;
    else early_pending_.push_back(e);
    return;
  }
  early_seg_run(e);
}
template <typename T, typename TS>
void Model<T, TS>::early_seg_run(const PendingEarly& e) {
  const anyref_config& c = cfg;
  const int H = c.llm_dim, slot = early_done_, L = 4 * sam_g_;
  const int64_t hw = (int64_t)e.orig_hw[0] * e.orig_hw[1];
  if (early_stop_ || slot >= c.max_seg || (slot + 1) * hw > e.cap) {  // run_tail refuses the call with the proper message
    early_stop_ = true;
    return;
  }
  HIP_TRY(hipStreamWaitEvent(s2_, e.ready, 0));
  HIP_TRY(hipMemcpyAsync(seg_h_ + (size_t)slot * H, hidden_all_ + (size_t)e.hidden_row * H, (size_t)H * 4,
                         hipMemcpyDeviceToDevice, s2_));
  gemmf(s2_, seg_h_ + (size_t)slot * H, H, fc1_, seg_t_ + (size_t)slot * H, H, 1, ACT_RELU);
  gemmf(s2_, seg_t_ + (size_t)slot * H, H, fc2_, pred_emb_ + (size_t)slot * c.out_dim, c.out_dim, 1, ACT_NONE);
  mask_decoder(s2_, sam_emb_, pred_emb_ + (size_t)slot * c.out_dim, 1, nullptr, nullptr);
  launch_postprocess(m_masks_, (int64_t)c.num_mask_tokens * L * L, 1, L, L, c.sam_img, e.resized_hw[0], e.resized_hw[1],
                     e.orig_hw[0], e.orig_hw[1], e.out_masks + slot * hw, s2_);
  if (e.out_low)
    HIP_TRY(hipMemcpyAsync(e.out_low + (size_t)slot * L * L, m_masks_, (size_t)L * L * 4, hipMemcpyDeviceToDevice, s2_));
  HIP_TRY(hipEventRecord(ev_sam_, s2_));  // the join now waits for this mask too
  ++early_done_;
}

template <typename T, typename TS>
void Model<T, TS>::run_tail(hipStream_t s, const float* sam_images, int B, const std::vector<int>& seg_b,
                        const std::vector<int>& seg_pos, const std::vector<int>& reph_s, const int32_t* resized_hw,
                        const int32_t* orig_hw, int32_t* out_nseg, float* out_masks, int64_t out_masks_cap,
                        int64_t* mask_offsets, float* out_low, const float* attn_given, int attn_n) {
  // seg_b/seg_pos: (image, row of hidden_all_) of every [SEG]; reph_s: rephrase start row per image
  // attn_given (seg_tail only): caller's head-mean last-layer attention [B, attn_n, attn_n] instead of the K cache
  const anyref_config& c = cfg;
  const int H = c.llm_dim, S = c.llm_max_seq, nseg = (int)seg_b.size();
  // join BEFORE anything below can throw: a refused call must not leave the encoder running on the side stream
  // over the caller's images, nor a stale "forked" flag for the next call
  if (sam_forked_ && !sam_enq_done_) sam_feed((int)sam_blocks_.size(), false);  // (also flushes the pending early masks)
  const bool sam_ready = sam_forked_;
  const int done = early_done_;  // generate, batch 1: rows [0, done) were decoded by early_seg, masks and all
  early_done_ = 0;
  join_sam(s);
  for (int b = 0; b < B; ++b) out_nseg[b] = 0;
  for (int i = 0; i < nseg; ++i) out_nseg[seg_b[i]]++;
  for (int b = 0; b < B; ++b)
    if (out_nseg[b] > c.max_seg) throw std::runtime_error("more [SEG] tokens in one image than max_seg");
  int64_t off = 0;
  for (int b = 0; b < B; ++b) {
    mask_offsets[b] = off;
    off += (int64_t)out_nseg[b] * orig_hw[2 * b] * orig_hw[2 * b + 1];
  }
  if (off > out_masks_cap) throw std::runtime_error("out_masks capacity too small");
  if (done > nseg) throw std::runtime_error("internal: more early [SEG] masks than [SEG] tokens");
  if (nseg == done) return;
  {
    StageScope stage_scope(this, ST_SEG, s);
    for (int i = 0; i < nseg; ++i) {
      stage_seg_[i] = seg_b[i];
      stage_seg_[nseg + i] = seg_pos[i];
    }
    HIP_TRY(hipMemcpyAsync(idx_a_, stage_seg_, nseg * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(idx_b_, stage_seg_ + nseg, nseg * 4, hipMemcpyHostToDevice, s));
  }
  launch_gather_rows(hidden_all_, S, H, idx_a_, idx_b_, nseg, seg_h_, s);
  if (c.rephrase_weight > 0.f) {
    StageScope stage_scope(this, ST_KL, s);  // (the event is recorded at the end of this block whether or not a copy was queued)
    // anyref.py:735-755,767-769: the first [SEG] of image i gets + w * sum_j attn_j * hidden_j
    const int nh = c.llm_heads, hd = H / nh;
    Q* kc = kcache_ + cache_layer_stride_ * (c.llm_layers - 1);
    std::vector<char> done(B, 0);
    for (int i = 0; i < nseg; ++i) {
      const int b = seg_b[i];
      if (done[b]) continue;
      done[b] = 1;
      const int e0 = seg_pos[i], s0 = reph_s[b];
      if (e0 <= s0) continue;
      if (attn_given) {
        HIP_TRY(hipMemcpyAsync(attn_row_ + (size_t)b * S, attn_given + ((size_t)b * attn_n + e0) * attn_n,
                               (size_t)attn_n * 4, hipMemcpyDeviceToDevice, s));
      } else {
        stage_kl_[b] = e0 + 1;
        HIP_TRY(hipMemcpyAsync(kvlen_dev_ + b, stage_kl_ + b, 4, hipMemcpyHostToDevice, s));
        launch_attn_row_mean<Q>(q_last_ + ((size_t)b * S + e0) * H, 0, hd, kc + (size_t)b * S * H, 0, H, hd,
                                kvlen_dev_ + b, 1, nh, hd, 1.f / sqrtf((float)hd), attn_row_ + (size_t)b * S, S, s);
      }
      launch_rephrase(hidden_all_ + (size_t)b * S * H, H, attn_row_ + (size_t)b * S, s0, e0, c.rephrase_weight,
                      seg_h_ + (size_t)i * H, s);
    }
  }
  gemmf(s, seg_h_, H, fc1_, seg_t_, H, nseg, ACT_RELU);
  gemmf(s, seg_t_, H, fc2_, pred_emb_, c.out_dim, nseg, ACT_NONE);
  if (!sam_ready) sam_encoder(s, sam_images, B, sam_emb_);
  const int NK = sam_g_ * sam_g_, C = c.sam_out_chans, L = 4 * sam_g_;
  int row = 0;
  for (int b = 0; b < B; ++b) {
    const int n = out_nseg[b];
    if (n == 0) continue;
    const int j0 = b == 0 ? done : 0;  // early masks (batch 1) are already in place
    const int64_t hw = (int64_t)orig_hw[2 * b] * orig_hw[2 * b + 1];
    // seg rows of image b are contiguous because the host emits them image by image
    if (n > j0) {
      mask_decoder(s, sam_emb_ + (size_t)b * NK * C, pred_emb_ + (size_t)(row + j0) * c.out_dim, n - j0, nullptr, nullptr);
      launch_postprocess(m_masks_, (int64_t)c.num_mask_tokens * L * L, n - j0, L, L, c.sam_img, resized_hw[2 * b],
                         resized_hw[2 * b + 1], orig_hw[2 * b], orig_hw[2 * b + 1], out_masks + mask_offsets[b] + j0 * hw,
                         s);
      if (out_low)
        for (int j = j0; j < n; ++j)
          HIP_TRY(hipMemcpyAsync(out_low + ((size_t)b * c.max_seg + j) * L * L,
                                 m_masks_ + (size_t)(j - j0) * c.num_mask_tokens * L * L, (size_t)L * L * 4,
                                 hipMemcpyDeviceToDevice, s));
    }
    row += n;
  }
}

template <typename T, typename TS>
void Model<T, TS>::generate(hipStream_t s, const float* clip_images, const float* sam_images, const int64_t* input_ids,
                        const int32_t* lens, int B, int Lmax, const float* extra_embeds, const int32_t* extra_slots,
                        int n_extra, const int32_t* resized_hw, const int32_t* orig_hw, int max_new_tokens,
                        int eos_token_id, int64_t* out_ids, int32_t* out_lens, int32_t* out_nseg, float* out_masks,
                        int64_t out_masks_cap, int64_t* mask_offsets, float* out_low, float* out_hidden) {
  HIP_TRY(hipSetDevice(device_));
  const anyref_config& c = cfg;
  if (!finalized_) throw std::runtime_error("generate before finalize");
  if (B <= 0 || B > c.max_batch) throw std::runtime_error("batch exceeds max_batch");
  if (max_new_tokens < 1) throw std::runtime_error("max_new_tokens must be >= 1");
  const int H = c.llm_dim, S = c.llm_max_seq, n_img = clip_n_;
  const bool keep_q = c.rephrase_weight > 0.f;

  SamJoinGuard<Model<T, TS>> join_guard{this, s};
  // Batch 1 with a CU share set: the encoder is fed to the side stream a few capped blocks at a time (fork_sam's
  // note) -- the first ones beside the CLIP tower (257 tokens: ~170 launches of <= 96 workgroups that leave most CUs
  // idle), none during prefill (MFMA-bound itself), the rest per decode step; larger batches bring more encoder work
  // than the loop can hide at a reduced share: queued whole, uncapped, after prefill.
  const int nblk = (int)sam_blocks_.size();
  static const bool any_b = getenv("ANYREF_SIDE_ANYB") != nullptr;  // lab: the CU share for batches > 1 too
  // (the split-pair mode too: its GEMMs have the capped forms; its attention launches stay uncapped.  Same share / step
  //  count as perf, alternated on one box: 57.1 - 57.3 ms uncapped-after-prefill -> 52.7 - 53.3 ms fed; 160 / 192 / 96 workgroups
  //  or 9 steps: 53.2 - 55.8)
  // Round 4: 2 - 4 images per call are fed too, at 160 workgroups over 3 steps (bench.py --config c3, one box, ms per 4-image
  // call: whole / uncapped after prefill 81.5; fed at 128 x 9 steps 83.6, 128 x 3 85.1, 144 x 4 83.4, **160 x 3 77.7**, 160 x 6 78.1,
  // 176 x 6 78.4, 192 x 2 - 9 78.3 - 78.9, 200 - 224: 80.6 - 80.9; without the head blocks beside CLIP 81.2; two images 56.3 -> 54.8).
  // Eight images lose (128.7 -> 135.6 - 164.6 ms): whole, as before.
  const int side_cap = B == 1 ? side_wgs_ : side_wgs_b_;
  const bool fed = (B <= 4 || any_b) && side_cap > 0 && overlap_ && IS16;
  const int steps_now = B == 1 ? side_steps_ : side_steps_b_;
  side_cap_now_ = side_cap;
  const int per_step = std::max(1, (nblk + steps_now - 1) / std::max(1, steps_now));
  if (fed) {
    fork_sam(s, sam_images, B, true);
    {
      // (lab knob: a share of its own for the blocks beside the CLIP tower, whose launches leave most CUs idle)
      static const int head_wgs = getenv("ANYREF_SIDE_HEAD_WGS") ? atoi(getenv("ANYREF_SIDE_HEAD_WGS")) : 0;
      if (head_wgs > 0) side_cap_now_ = head_wgs;
      sam_feed(std::min(side_head_, nblk - 1), true);
      side_cap_now_ = side_cap;
    }
  }
  clip_tower(s, clip_images, B);
  std::vector<int> slen, img_pos;
  if (extra_ev_) {  // extra_embeds were queued on another stream (anyref_set_extra_event): first use is the splice
    hipEvent_t ev = reinterpret_cast<hipEvent_t>(extra_ev_);
    extra_ev_ = nullptr;
    HIP_TRY(hipStreamWaitEvent(s, ev, 0));
  }
  const int Sp = splice_inputs(s, input_ids, lens, B, Lmax, extra_embeds, extra_slots, n_extra, slen, img_pos);
  for (int b = 0; b < B; ++b)
    if (slen[b] + max_new_tokens > S) throw std::runtime_error("prompt + max_new_tokens exceeds llm_max_seq");
  {
    // (lab knob: encoder blocks queued beside prefill too; default none -- both are MFMA-bound)
    static const int pf_blocks = getenv("ANYREF_SIDE_PREFILL_BLOCKS") ? atoi(getenv("ANYREF_SIDE_PREFILL_BLOCKS")) : 0;
    if (fed && pf_blocks > 0) sam_feed(std::min(sam_next_blk_ + pf_blocks, nblk - 1), true);
  }
  llm_prefill(s, B, Sp, slen_dev_, keep_q);
  // Fork the SAM encoder only now: CLIP + prefill are MFMA-bound themselves, the decode loop that
  // follows is HBM-bound and leaves the matrix cores to the encoder on the second stream.
  if (fed) {  // the blocks behind the head ones wait for the end of prefill
    HIP_TRY(hipEventRecord(ev_fork_, s));
    HIP_TRY(hipStreamWaitEvent(s2_, ev_fork_, 0));
    sam_feed(sam_next_blk_ + per_step, true);
  } else {
    fork_sam(s, sam_images, B);
  }
  // first token: logits of the last prompt row of every sequence
  {
    {
    StageScope stage_scope(this, ST_FIRST, s);
    int *bb = stage_first_, *pp = stage_first_ + B, *sl = stage_first_ + 2 * B;
    for (int b = 0; b < B; ++b) {
      bb[b] = b;
      pp[b] = slen[b] - 1;
      sl[b] = slen[b] - 1;  // launch_argmax_next bumps it to the first generated position
    }
    HIP_TRY(hipMemcpyAsync(idx_a_, bb, B * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(idx_b_, pp, B * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(pos_dev_, sl, B * 4, hipMemcpyHostToDevice, s));
    }
    launch_gather_rows(hidden_all_, S, H, idx_a_, idx_b_, B, l_xlast_, s);
    GemvArgs h;
    h.x = l_xlast_; h.ldx = H; gemv_w(h, lm_head_); h.y = l_logits_; h.ldy = c.llm_vocab; h.B = B; h.N = c.llm_vocab;
    h.K = H;
    launch_gemv<T>(h, s);
    launch_argmax_next(l_logits_, B, c.llm_vocab, c.llm_vocab, next_dev_, pos_dev_, emb_table_, sizeof(W) == 2, H, S, d_x_,
                       rowmap_dev_, kvlen_dev_, s);
  }
  // greedy loop (HF greedy search: stop a row at EOS, pad finished rows; anyref.py:704-716)
  std::vector<std::vector<int64_t>> gen(B);
  std::vector<char> fin(B, 0);
  // With an EOS id the host must see token k before it may queue step k + 1 (a step queued past the EOS would be a
  // whole wasted pass over the weights).  Without one (fixed-length generation) nothing depends on the token's
  // value: its copy and an event are queued, THEN the next step, and the host waits for the event while the device
  // already runs that step -- no host round trip between steps.
  const bool ahead = eos_token_id < 0;
  // early [SEG] masks: batch 1, encoder on the side stream, no rephrasing (that reads attention over the whole answer),
  // no [SEG] inside the prompt (the tail's order is by position)
  early_done_ = 0;
  early_stop_ = false;
  bool early = !early_off_ && sam_forked_ && B == 1 && c.rephrase_weight <= 0.f;
  for (int i = 1; early && i < lens[0]; ++i)
    if (input_ids[i] >= c.seg_lo && input_ids[i] <= c.seg_hi) early = false;
  // a step runs beside the capped encoder while blocks are still being fed, and for one more step after the last feed
  // (the blocks queued during a step run at about half speed: through the step that follows)
  int corun_tail = 1;
  auto corun_step = [&]() {
    if (!fed) return false;
    if (!sam_enq_done_) return true;
    return corun_tail-- > 0;
  };
  for (int step = 0; step < max_new_tokens; ++step) {
    int64_t* tok = next_host_ + (size_t)(step & 1) * c.max_batch;
    const bool last = step == max_new_tokens - 1;
    HIP_TRY(hipMemcpyAsync(tok, next_dev_, B * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(ev_tok_[step & 1], s));
    if (ahead && !last) {
      decode_step_graph(s, B, keep_q, corun_step());
      if (fed) sam_feed(sam_next_blk_ + per_step, true);  // the encoder's share of this step (host is idle until the token)
    }
    HIP_TRY(hipEventSynchronize(ev_tok_[step & 1]));
    bool all = true;
    for (int b = 0; b < B; ++b) {
      if (!fin[b]) {
        gen[b].push_back(tok[b]);
        if (eos_token_id >= 0 && tok[b] == eos_token_id) fin[b] = 1;
        // token `step` was predicted by hidden row slen + step - 1, final since ev_tok_ of this step
        if (early && !early_stop_ && tok[b] >= c.seg_lo && tok[b] <= c.seg_hi)
          early_seg(ev_tok_[step & 1], slen[0] + step - 1, resized_hw, orig_hw, out_masks, out_masks_cap, out_low);
      }
      all = all && fin[b];
    }
    if (all || last) break;
    if (!ahead) {
      decode_step_graph(s, B, keep_q, corun_step());
      if (fed) sam_feed(sam_next_blk_ + per_step, true);
    }
  }
  if (fed) sam_feed(nblk, false);  // the loop is over: what is left of the encoder has the chip to itself
  const int Lout = Lmax + max_new_tokens;
  std::vector<int> seg_b, seg_pos, reph(B, 0);
  for (int b = 0; b < B; ++b) {
    int64_t* row = out_ids + (size_t)b * Lout;
    for (int i = 0; i < Lout; ++i) row[i] = 0;
    for (int i = 0; i < lens[b]; ++i) row[i] = input_ids[(size_t)b * Lmax + i];
    for (size_t i = 0; i < gen[b].size(); ++i) row[lens[b] + i] = gen[b][i];
    out_lens[b] = lens[b] + (int)gen[b].size();
    // [SEG] search in output_ids[:,1:] (anyref.py:723-726); hidden row = p + n_img - 1 (the "+255")
    const int shift = img_pos[b] >= 0 ? n_img - 1 : 0;
    for (int p = 0; p + 1 < out_lens[b]; ++p) {
      const int64_t id = row[p + 1];
      if (id >= c.seg_lo && id <= c.seg_hi) {
        const int hp = p + shift;
        if (hp >= slen[b] + (int)gen[b].size() - 1) continue;  // no hidden state past the last forward
        seg_b.push_back(b);
        seg_pos.push_back(hp);
      }
    }
    reph[b] = lens[b] - 1 + shift;  // input_ids[i,1:].shape[0] + 255 (anyref.py:745)
  }
  run_tail(s, sam_images, B, seg_b, seg_pos, reph, resized_hw, orig_hw, out_nseg, out_masks, out_masks_cap,
           mask_offsets, out_low);
  if (out_hidden)
    HIP_TRY(hipMemcpyAsync(out_hidden, hidden_all_, (size_t)B * S * H * 4, hipMemcpyDeviceToDevice, s));
}

template <typename T, typename TS>
void Model<T, TS>::forward_teacher(hipStream_t s, const float* clip_images, const float* sam_images,
                               const int64_t* input_ids, const int32_t* lens, int B, int Lmax,
                               const float* extra_embeds, const int32_t* extra_slots, int n_extra,
                               const int32_t* rephrase_start, const int32_t* resized_hw, const int32_t* orig_hw,
                               int32_t* out_nseg, float* out_masks, int64_t out_masks_cap, int64_t* mask_offsets,
                               float* out_low, float* out_hidden, float* out_logits) {
  HIP_TRY(hipSetDevice(device_));
  const anyref_config& c = cfg;
  if (!finalized_) throw std::runtime_error("forward before finalize");
  if (B <= 0 || B > c.max_batch) throw std::runtime_error("batch exceeds max_batch");
  const int H = c.llm_dim, S = c.llm_max_seq, n_img = clip_n_;
  const bool keep_q = c.rephrase_weight > 0.f;
  early_done_ = 0;  // a generate() that threw between early_seg and run_tail must not leak its count into this call
  early_stop_ = false;
  SamJoinGuard<Model<T, TS>> join_guard{this, s};
  fork_sam(s, sam_images, B);
  clip_tower(s, clip_images, B);
  std::vector<int> slen, img_pos;
  if (extra_ev_) {  // extra_embeds were queued on another stream (anyref_set_extra_event): first use is the splice
    hipEvent_t ev = reinterpret_cast<hipEvent_t>(extra_ev_);
    extra_ev_ = nullptr;
    HIP_TRY(hipStreamWaitEvent(s, ev, 0));
  }
  const int Sp = splice_inputs(s, input_ids, lens, B, Lmax, extra_embeds, extra_slots, n_extra, slen, img_pos);
  llm_prefill(s, B, Sp, slen_dev_, keep_q);
  if (out_logits) {
    // [B, Sp, vocab] for the caller's LM loss
    for (int b = 0; b < B; ++b) {
      launch_convert<T>(hidden_all_ + (size_t)b * S * H, H, l_h_, apad<T>(H), slen[b], H, s);
      gemm(s, l_h_, H, lm_head_, out_logits + (size_t)b * Sp * c.llm_vocab, c.llm_vocab, slen[b], ACT_NONE, true);
    }
  }
  std::vector<int> seg_b, seg_pos, reph(B, 0);
  for (int b = 0; b < B; ++b) {
    const int shift = img_pos[b] >= 0 ? n_img - 1 : 0;
    for (int p = 1; p < lens[b]; ++p) {  // hidden that predicted the [SEG]: pos - 1 + 255 (anyref.py:282)
      const int64_t id = input_ids[(size_t)b * Lmax + p];
      if (id >= c.seg_lo && id <= c.seg_hi) {
        seg_b.push_back(b);
        seg_pos.push_back(p - 1 + shift);
      }
    }
    reph[b] = rephrase_start ? rephrase_start[b] - 1 + shift : 0;  // where(labels>0)[0][0] - 1 + 255 (anyref.py:378)
  }
  run_tail(s, sam_images, B, seg_b, seg_pos, reph, resized_hw, orig_hw, out_nseg, out_masks, out_masks_cap,
           mask_offsets, out_low);
  if (out_hidden)
    HIP_TRY(hipMemcpyAsync(out_hidden, hidden_all_, (size_t)B * S * H * 4, hipMemcpyDeviceToDevice, s));
}

template <typename T, typename TS>
void Model<T, TS>::seg_tail(hipStream_t s, const float* sam_images, const int64_t* ids, const int32_t* ids_lens,
                        const int32_t* ref_pos, int B, int Lmax, int teacher, const float* hidden, int hidden_rows,
                        const float* attn_mean, const int32_t* resized_hw, const int32_t* orig_hw, int32_t* out_nseg,
                        float* out_masks, int64_t out_masks_cap, int64_t* mask_offsets, float* out_low) {
  HIP_TRY(hipSetDevice(device_));
  const anyref_config& c = cfg;
  if (!finalized_) throw std::runtime_error("seg_tail before finalize");
  if (B <= 0 || B > c.max_batch) throw std::runtime_error("batch exceeds max_batch");
  const int H = c.llm_dim, S = c.llm_max_seq, shift = clip_n_ - 1;  // the reference's hard-coded "+255"
  if (hidden_rows <= 0 || hidden_rows > S) throw std::runtime_error("hidden_rows exceeds llm_max_seq");
  if (c.rephrase_weight > 0.f && !attn_mean) throw std::runtime_error("rephrase_weight > 0 needs attn_mean");
  early_done_ = 0;  // see forward_teacher
  early_stop_ = false;
  for (int b = 0; b < B; ++b)
    HIP_TRY(hipMemcpyAsync(hidden_all_ + (size_t)b * S * H, hidden + (size_t)b * hidden_rows * H,
                           (size_t)hidden_rows * H * 4, hipMemcpyDeviceToDevice, s));
  std::vector<int> seg_b, seg_pos, reph(B, 0);
  for (int b = 0; b < B; ++b) {
    if (ids_lens[b] <= 0 || ids_lens[b] > Lmax) throw std::runtime_error("bad sequence length");
    const int64_t* row = ids + (size_t)b * Lmax;
    // generate: [SEG] in output_ids[:,1:] at p -> hidden row p + 255 (anyref.py:723-726,:758);
    // forward:  [SEG] in input_ids at p       -> hidden row p - 1 + 255 (anyref.py:273-282)
    for (int p = 1; p < ids_lens[b]; ++p)
      if (row[p] >= c.seg_lo && row[p] <= c.seg_hi) {
        const int hp = p - 1 + shift;
        if (hp >= hidden_rows) continue;
        seg_b.push_back(b);
        seg_pos.push_back(hp);
      }
    // generate: prompt length L -> L - 1 + 255 (:745); forward: where(labels > 0)[0][0] - 1 + 255 (:378)
    reph[b] = ref_pos ? ref_pos[b] - 1 + shift : 0;
  }
  (void)teacher;  // both searches land on the same row arithmetic; kept in the ABI to name the caller's convention
  run_tail(s, sam_images, B, seg_b, seg_pos, reph, resized_hw, orig_hw, out_nseg, out_masks, out_masks_cap,
           mask_offsets, out_low, attn_mean, hidden_rows);
}

std::unique_ptr<ModelBase> make_model(const anyref_config& cfg, int device) {
  if (cfg.mode == ANYREF_MODE_PARITY) return std::unique_ptr<ModelBase>(new Model<float>(cfg, device));
  if (cfg.mode == ANYREF_MODE_PARITY16) return std::unique_ptr<ModelBase>(new Model<sp16, sp16>(cfg, device));
  if (cfg.mode == ANYREF_MODE_PERF || cfg.mode == ANYREF_MODE_PERF_FP8W) {
    static const bool sam_bf16 = getenv("ANYREF_SAM_BF16") != nullptr;  // A/B: the all-bf16 handle of rounds 1-2
    if (sam_bf16) return std::unique_ptr<ModelBase>(new Model<bf16, bf16>(cfg, device));
    return std::unique_ptr<ModelBase>(new Model<bf16, f16>(cfg, device));
  }
  throw std::runtime_error("unknown mode");
}

}  // namespace anyref

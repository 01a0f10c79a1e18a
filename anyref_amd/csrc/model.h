// Host-side model: weight store, packed weights, workspaces, stage orchestration.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/anyref_hip.h"
#include "kernels.h"

namespace anyref {

struct RawTensor {
  float* p = nullptr;  // f32 device copy
  std::vector<int64_t> shape;
  int64_t numel() const {
    int64_t n = 1;
    for (auto s : shape) n *= s;
    return n;
  }
};

// Mode-independent interface behind the C-ABI.
class ModelBase {
 public:
  explicit ModelBase(const anyref_config& c, int device);
  virtual ~ModelBase();

  void set_weight(const char* name, const void* ptr, int is_device, int dtype, const int64_t* shape, int ndim);
  virtual void finalize() = 0;
  virtual const char* mode_name() const = 0;

  virtual void generate(hipStream_t s, const float* clip_images, const float* sam_images, const int64_t* input_ids,
                        const int32_t* lens, int B, int Lmax, const float* extra_embeds,
                        const int32_t* extra_slots, int n_extra, const int32_t* resized_hw, const int32_t* orig_hw,
                        int max_new_tokens, int eos_token_id, int64_t* out_ids, int32_t* out_lens,
                        int32_t* out_nseg, float* out_masks, int64_t out_masks_cap, int64_t* mask_offsets,
                        float* out_low, float* out_hidden) = 0;
  virtual void forward_teacher(hipStream_t s, const float* clip_images, const float* sam_images,
                               const int64_t* input_ids, const int32_t* lens, int B, int Lmax,
                               const float* extra_embeds, const int32_t* extra_slots, int n_extra,
                               const int32_t* rephrase_start, const int32_t* resized_hw, const int32_t* orig_hw,
                               int32_t* out_nseg, float* out_masks, int64_t out_masks_cap, int64_t* mask_offsets,
                               float* out_low, float* out_hidden, float* out_logits) = 0;
  virtual void encode_images(hipStream_t s, const float* clip_images, int B, float* out, float* clip_feat) = 0;
  virtual void sam_encode(hipStream_t s, const float* sam_images, int B, float* out) = 0;
  virtual void mask_decode(hipStream_t s, const float* image_emb, const float* pred_emb, int n, float* masks4,
                           float* iou, const int32_t* resized_hw, const int32_t* orig_hw, float* out_masks) = 0;
  virtual void llm_forward(hipStream_t s, const float* embeds, const int32_t* lens, int B, int S, float* hidden,
                           float* logits, const int32_t* attn_q, float* attn_row) = 0;
  virtual void project_audio(hipStream_t s, const float* audio_emb, int n, float* out) = 0;
  virtual void audio_encode(hipStream_t s, const float* mel, int n, float* emb) = 0;
  virtual void seg_tail(hipStream_t s, const float* sam_images, const int64_t* ids, const int32_t* ids_lens,
                        const int32_t* ref_pos, int B, int Lmax, int teacher, const float* hidden, int hidden_rows,
                        const float* attn_mean, const int32_t* resized_hw, const int32_t* orig_hw,
                        int32_t* out_nseg, float* out_masks, int64_t out_masks_cap, int64_t* mask_offsets,
                        float* out_low) = 0;

  int64_t device_bytes() const { return bytes_; }
  void set_seg_range(int lo, int hi) {
    cfg.seg_lo = lo;
    cfg.seg_hi = hi;
  }
  // two-stream overlap of the SAM encoder with the LLM decode (default on; measurement harness can
  // switch it off to time kernels without a co-running stream)
  void set_overlap(bool on) { overlap_ = on; }
  // masks of generated [SEG]s decoded on the side stream while the greedy loop goes on (default on; batch 1)
  void set_early_tail(bool on) { early_off_ = !on; }
  // the NEXT generate / forward_teacher waits for this event on its stream just before it reads extra_embeds (the splice):
  // the caller may produce them (ImageBind audio trunk + projector) on a stream of its own, beside the CLIP tower
  void set_extra_event(void* ev) { extra_ev_ = ev; }
  void* extra_ev_ = nullptr;
  // hipGraph replay of the greedy decode step (default on)
  void set_graphs(bool on) { use_graphs_ = on; }
  // workgroup cap of the encoder's GEMM / attention launches while it co-runs with the decode loop (0 = uncapped)
  // `steps`: the decode steps the capped encoder is spread over (its blocks are queued ceil(depth / steps) per step)
  void set_side_share(int wgs, int steps) {
    side_wgs_ = wgs < 0 ? 0 : wgs;
    if (steps > 0) side_steps_ = steps;
  }
  bool early_off_ = getenv("ANYREF_NO_EARLY_TAIL") != nullptr;
  Profiler prof;
  Stamper stamp;  // kernel-side timestamps of the decode GEMVs (bench.py's in-situ roofline)
  std::string err;
  int n_unknown = 0;

 protected:
  void* dalloc(size_t bytes);  // tracked hipMalloc (freed in the destructor)
  void dfree(void* p);
  const RawTensor& raw(const std::string& name) const;
  bool has_raw(const std::string& name) const { return raw_.count(name) != 0; }
  void drop_raw();

  anyref_config cfg;
  int device_;
  int64_t bytes_ = 0;
  std::unordered_map<void*, size_t> allocs_;
  std::map<std::string, RawTensor> raw_;
  bool finalized_ = false;
  bool overlap_ = true;
  bool use_graphs_ = true;
  int side_wgs_ = getenv("ANYREF_SIDE_WGS") ? atoi(getenv("ANYREF_SIDE_WGS")) : 128;
  int side_steps_ = getenv("ANYREF_SIDE_STEPS") ? atoi(getenv("ANYREF_SIDE_STEPS")) : 6;
  int side_head_ = getenv("ANYREF_SIDE_HEAD") ? atoi(getenv("ANYREF_SIDE_HEAD")) : 3;  // encoder blocks queued beside CLIP
  // 2 - 4 images per call (C3's per-GPU shape): the same feeding at a larger share over fewer steps (model.hip generate)
  int side_wgs_b_ = getenv("ANYREF_SIDE_WGS_B") ? atoi(getenv("ANYREF_SIDE_WGS_B")) : 160;
  int side_steps_b_ = getenv("ANYREF_SIDE_STEPS_B") ? atoi(getenv("ANYREF_SIDE_STEPS_B")) : 3;
  int side_cap_now_ = 0;  // the cap sam_feed applies during this call
};

std::unique_ptr<ModelBase> make_model(const anyref_config& cfg, int device);

}  // namespace anyref

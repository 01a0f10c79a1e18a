// extern "C" boundary of libanyref_hip.so (include/anyref_hip.h).  Exceptions never cross it:
// every entry point returns a status and stores the text for anyref_last_error().
#include <cstring>
#include <string>

#include "model.h"

using namespace anyref;

struct anyref_handle {
  std::unique_ptr<ModelBase> m;
  std::string err;
  std::vector<StampRow> stamp_rows;  // last anyref_stamps_collect()
};

static thread_local std::string g_create_err;

#define GUARD(h, body)                         \
  if (!(h) || !(h)->m) return 1;               \
  try {                                        \
    struct ProfInstall {                       \
      ProfInstall(Profiler* p, Stamper* t) { g_prof = p; g_stamp = t; } \
      ~ProfInstall() { g_prof = nullptr; g_stamp = nullptr; }            \
    } _pi(&(h)->m->prof, &(h)->m->stamp);      \
    body;                                      \
    hipError_t _e = hipGetLastError();         \
    if (_e != hipSuccess) {                    \
      (h)->err = std::string("HIP error: ") + hipGetErrorString(_e); \
      return 3;                                \
    }                                          \
    return 0;                                  \
  } catch (const std::exception& e) {          \
    (h)->err = e.what();                       \
    return 2;                                  \
  }

extern "C" {

int anyref_create(const anyref_config* cfg, int device, anyref_handle** out) {
  try {
    if (!cfg || !out) throw std::runtime_error("null argument");
    if (cfg->abi_version != ANYREF_ABI_VERSION) throw std::runtime_error("anyref_config ABI version mismatch");
    if (cfg->max_batch < 1 || cfg->max_seg < 1) throw std::runtime_error("max_batch / max_seg must be >= 1");
    if (cfg->llm_dim % cfg->llm_heads || cfg->clip_dim % cfg->clip_heads || cfg->sam_dim % cfg->sam_heads)
      throw std::runtime_error("hidden sizes must be divisible by the head counts");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
      throw std::runtime_error("no HIP device visible: the AnyRef HIP backend needs an MI355X (there is no CPU fallback)");
    if (device < 0 || device >= ndev) throw std::runtime_error("device index out of range");
    auto* h = new anyref_handle();
    h->m = make_model(*cfg, device);
    *out = h;
    return 0;
  } catch (const std::exception& e) {
    g_create_err = e.what();
    return 2;
  }
}

void anyref_destroy(anyref_handle* h) { delete h; }

const char* anyref_last_error(anyref_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int anyref_set_weight(anyref_handle* h, const char* name, const void* ptr, int is_device, int dtype,
                      const int64_t* shape, int ndim) {
  GUARD(h, h->m->set_weight(name, ptr, is_device, dtype, shape, ndim));
}

int anyref_finalize(anyref_handle* h) { GUARD(h, h->m->finalize()); }

int anyref_generate(anyref_handle* h, void* stream, const float* clip_images, const float* sam_images,
                    const int64_t* input_ids, const int32_t* lens, int B, int Lmax, const float* extra_embeds,
                    const int32_t* extra_slots, int n_extra, const int32_t* resized_hw, const int32_t* orig_hw,
                    int max_new_tokens, int eos_token_id, int64_t* out_ids, int32_t* out_lens, int32_t* out_nseg,
                    float* out_masks, int64_t out_masks_cap, int64_t* mask_offsets, float* out_low,
                    float* out_hidden) {
  GUARD(h, h->m->generate((hipStream_t)stream, clip_images, sam_images, input_ids, lens, B, Lmax, extra_embeds,
                          extra_slots, n_extra, resized_hw, orig_hw, max_new_tokens, eos_token_id, out_ids,
                          out_lens, out_nseg, out_masks, out_masks_cap, mask_offsets, out_low, out_hidden));
}

int anyref_forward_teacher(anyref_handle* h, void* stream, const float* clip_images, const float* sam_images,
                           const int64_t* input_ids, const int32_t* lens, int B, int Lmax,
                           const float* extra_embeds, const int32_t* extra_slots, int n_extra,
                           const int32_t* rephrase_start, const int32_t* resized_hw, const int32_t* orig_hw,
                           int32_t* out_nseg, float* out_masks, int64_t out_masks_cap, int64_t* mask_offsets,
                           float* out_low, float* out_hidden, float* out_logits) {
  GUARD(h, h->m->forward_teacher((hipStream_t)stream, clip_images, sam_images, input_ids, lens, B, Lmax,
                                 extra_embeds, extra_slots, n_extra, rephrase_start, resized_hw, orig_hw, out_nseg,
                                 out_masks, out_masks_cap, mask_offsets, out_low, out_hidden, out_logits));
}

int anyref_encode_images(anyref_handle* h, void* stream, const float* clip_images, int B, float* out,
                         float* clip_feat) {
  GUARD(h, h->m->encode_images((hipStream_t)stream, clip_images, B, out, clip_feat));
}

int anyref_sam_encode(anyref_handle* h, void* stream, const float* sam_images, int B, float* out) {
  GUARD(h, h->m->sam_encode((hipStream_t)stream, sam_images, B, out));
}

int anyref_mask_decode(anyref_handle* h, void* stream, const float* image_emb, const float* pred_emb, int n,
                       float* masks4, float* iou, const int32_t* resized_hw, const int32_t* orig_hw,
                       float* out_masks) {
  GUARD(h, h->m->mask_decode((hipStream_t)stream, image_emb, pred_emb, n, masks4, iou, resized_hw, orig_hw, out_masks));
}

int anyref_llm_forward(anyref_handle* h, void* stream, const float* embeds, const int32_t* lens, int B, int S,
                       float* hidden, float* logits, const int32_t* attn_q, float* attn_row) {
  GUARD(h, h->m->llm_forward((hipStream_t)stream, embeds, lens, B, S, hidden, logits, attn_q, attn_row));
}

int anyref_audio_encode(anyref_handle* h, void* stream, const float* mel, int n, float* emb) {
  GUARD(h, h->m->audio_encode((hipStream_t)stream, mel, n, emb));
}

int anyref_project_audio(anyref_handle* h, void* stream, const float* audio_emb, int n, float* out) {
  GUARD(h, h->m->project_audio((hipStream_t)stream, audio_emb, n, out));
}

int anyref_seg_tail(anyref_handle* h, void* stream, const float* sam_images, const int64_t* ids,
                    const int32_t* ids_lens, const int32_t* ref_pos, int B, int Lmax, int teacher, const float* hidden,
                    int hidden_rows, const float* attn_mean, const int32_t* resized_hw, const int32_t* orig_hw,
                    int32_t* out_nseg, float* out_masks, int64_t out_masks_cap, int64_t* mask_offsets, float* out_low) {
  GUARD(h, h->m->seg_tail((hipStream_t)stream, sam_images, ids, ids_lens, ref_pos, B, Lmax, teacher, hidden,
                          hidden_rows, attn_mean, resized_hw, orig_hw, out_nseg, out_masks, out_masks_cap,
                          mask_offsets, out_low));
}

int anyref_set_seg_range(anyref_handle* h, int lo, int hi) { GUARD(h, h->m->set_seg_range(lo, hi)); }

int anyref_set_overlap(anyref_handle* h, int on) { GUARD(h, h->m->set_overlap(on != 0)); }
int anyref_set_early_tail(anyref_handle* h, int on) { GUARD(h, h->m->set_early_tail(on != 0)); }
int anyref_set_extra_event(anyref_handle* h, void* event) { GUARD(h, h->m->set_extra_event(event)); }

int anyref_set_graphs(anyref_handle* h, int on) { GUARD(h, h->m->set_graphs(on != 0)); }
int anyref_set_side_share(anyref_handle* h, int wgs, int steps) { GUARD(h, h->m->set_side_share(wgs, steps)); }

int anyref_profile_enable(anyref_handle* h, int on) {
  GUARD(h, {
    h->m->prof.on = on != 0;
    if (on) h->m->prof.reset();
  });
}

int anyref_profile_config(anyref_handle* h, const char* only_tag, int sample_every) {
  GUARD(h, {
    h->m->prof.filter = only_tag ? only_tag : "";
    h->m->prof.sample_every = sample_every > 1 ? sample_every : 1;
  });
}

int anyref_profile_collect(anyref_handle* h) { GUARD(h, h->m->prof.collect()); }

int anyref_profile_calibrate(anyref_handle* h, void* stream, double* overhead_us) {
  GUARD(h, {
    h->m->prof.calibrate((hipStream_t)stream);
    if (overhead_us) *overhead_us = h->m->prof.null_ms * 1e3;
  });
}

int anyref_profile_read(anyref_handle* h, int idx, char* name, int cap, double* ms, int64_t* count, double* flops,
                        double* bytes) {
  if (!h || !h->m) return -1;
  auto st = h->m->prof.stats();
  if (idx < 0 || idx >= (int)st.size()) return -1;
  snprintf(name, cap, "%s", st[idx].first.c_str());
  *ms = st[idx].second.ms;
  *count = st[idx].second.count;
  *flops = st[idx].second.flops;
  *bytes = st[idx].second.bytes;
  return 0;
}

int anyref_stamps_enable(anyref_handle* h, int on) { GUARD(h, h->m->stamp.enable(on != 0)); }

int anyref_stamps_dropped(anyref_handle* h, int64_t* count) {
  GUARD(h, {
    if (!count) throw std::runtime_error("null argument");
    *count = h->m->stamp.dropped();
  });
}
int anyref_stamps_collect(anyref_handle* h, int64_t* count) {
  GUARD(h, {
    h->stamp_rows = h->m->stamp.collect();
    if (count) *count = (int64_t)h->stamp_rows.size();
  });
}

int anyref_stamps_read(anyref_handle* h, int64_t idx, char* name, int cap, double* t0_us, double* t1_us, double* bytes,
                       int* epoch) {
  if (!h || idx < 0 || idx >= (int64_t)h->stamp_rows.size()) return -1;
  const StampRow& r = h->stamp_rows[(size_t)idx];
  snprintf(name, cap, "%s", r.tag.c_str());
  *t0_us = r.t0_us;
  *t1_us = r.t1_us;
  *bytes = r.bytes;
  if (epoch) *epoch = r.epoch;
  return 0;
}

int anyref_stamps_spread(anyref_handle* h, int64_t idx, double* start_spread_us, double* end_spread_us, double* wg_median_us) {
  if (!h || idx < 0 || idx >= (int64_t)h->stamp_rows.size()) return -1;
  const StampRow& r = h->stamp_rows[(size_t)idx];
  *start_spread_us = r.start_spread_us;
  *end_spread_us = r.end_spread_us;
  *wg_median_us = r.wg_median_us;
  return 0;
}

int64_t anyref_device_bytes(anyref_handle* h) { return h && h->m ? h->m->device_bytes() : 0; }

const char* anyref_mode_name(anyref_handle* h) { return h && h->m ? h->m->mode_name() : ""; }

}  // extern "C"

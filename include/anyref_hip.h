/*
 * anyref_hip.h — C-ABI of the MI355X-native AnyRef inference backend (libanyref_hip.so).
 *
 * The reference (jwh97nn/AnyRef) is pure Python and has NO FFI / plugin layer
 * (SURVEY.md §8b); its boundary for this path is the Python surface of
 * `AnyRefForCausalLM` (model/anyref.py:182-237, :647-822) plus the state_dict
 * key names.  This library sits one level below a Python class that reproduces
 * that surface (anyref_amd/model.py); every entry point cites the reference
 * interface whose arithmetic it replaces.
 *
 * Conventions
 *   - plain pointers + sizes only; no torch / STL types cross the boundary
 *   - every `dev` pointer is HBM memory on the handle's device, owned by the caller
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it
 *   - return 0 on success, non-zero on error; text via anyref_last_error()
 *   - one handle per GPU per process; calls on one handle are serialised by the caller
 */
#ifndef ANYREF_HIP_H
#define ANYREF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANYREF_ABI_VERSION 2

/* dtype codes for anyref_set_weight */
#define ANYREF_F32 0
#define ANYREF_BF16 1
#define ANYREF_F16 2

/* compute modes */
#define ANYREF_MODE_PARITY 0 /* fp32 activations + fp32-input MFMA (exact fp32 accumulate) */
#define ANYREF_MODE_PERF 1   /* bf16 weights/activations on bf16 MFMA, fp32 accumulate + fp32 residual stream */
/* PERF with the LLaMA linear layers (q/k/v/o, gate/up/down, lm_head) held as fp8 e4m3 bytes + one f32 scale
 * per output row (weight-only, quantised at finalize: scale = max|row| / 448, RNE): half the weight bytes of
 * the HBM-bound decode; activations, the vision towers and SAM stay as in PERF (BASELINE config 5) */
#define ANYREF_MODE_PERF_FP8W 2
/* The tolerance-meeting mode at 16-bit MFMA rate (north_star: mask logits within 1e-3, identical greedy ids): weights stay in
 * their exact bf16 storage (never widened in HBM: a decode step streams the same bytes as PERF); every activation that feeds a
 * matrix product is f32 carried as a PAIR of bf16 terms (hi = bf16(a), lo = bf16(a - hi): |a - hi - lo| <= 2^-18 |a|), one bf16
 * MFMA pass per term into the same f32 accumulator; the decode GEMV multiplies the f32 activation row (kept in LDS as f32)
 * with the bf16 weights; attention operands (q, k, v, the KV cache), residual streams, norms and the mask decoder are f32 as
 * in PARITY.  Inputs must be weights that are exactly representable in bf16 (the synthetic workloads round once; a real
 * fp32 SAM checkpoint is rounded to bf16 at finalize, as the reference's own fp16 evaluation rounds it to fp16). */
#define ANYREF_MODE_PARITY16 3

typedef struct anyref_config {
  int32_t abi_version; /* = ANYREF_ABI_VERSION */
  int32_t mode;        /* ANYREF_MODE_* */
  /* CLIP vision tower (HF CLIPVisionModel under the absent model/llava; anyref.py:190-192) */
  int32_t clip_image, clip_patch, clip_dim, clip_heads, clip_layers_run, clip_mlp;
  float clip_eps;
  /* LLaMA decoder (HF LlamaModel; anyref.py:211-215) */
  int32_t llm_vocab, llm_dim, llm_heads, llm_layers, llm_mlp, llm_max_seq;
  float llm_rms_eps, llm_rope_theta;
  /* SAM (build_sam.py:48-108) */
  int32_t sam_img, sam_patch, sam_dim, sam_depth, sam_heads, sam_mlp_ratio, sam_window;
  int32_t sam_n_global;
  int32_t sam_global_idx[8];
  int32_t sam_out_chans, dec_heads, dec_mlp, dec_depth, num_mask_tokens;
  /* glue (anyref.py:116-127,161,197-209) */
  int32_t out_dim, audio_dim;
  int32_t seg_lo, seg_hi; /* inclusive [SEG] id range */
  float rephrase_weight;
  int32_t max_batch; /* images per call on this GPU */
  int32_t max_seg;   /* [SEG] tokens per image the workspaces are sized for */
  /* ImageBind audio trunk in HIP (imagebind_model.py:175-192,331-338,391-395,425-428).
   * aud_blocks = 0: no trunk in the handle (the PyTorch-ROCm module feeds anyref_project_audio instead). */
  int32_t aud_dim, aud_blocks, aud_heads, aud_mel, aud_len, aud_kernel, aud_stride, aud_clips;
} anyref_config;

typedef struct anyref_handle anyref_handle;

/* Build an empty model on HIP device `device`. */
int anyref_create(const anyref_config* cfg, int device, anyref_handle** out);
void anyref_destroy(anyref_handle* h);
const char* anyref_last_error(anyref_handle* h); /* h may be NULL: error of the last failed create */

/*
 * Hand over one tensor under its reference state_dict name (SURVEY.md §8b "Weight names":
 * `model.visual_model.*`, `model.text_hidden_fcs.0.{0,2}.*`, `model.audio_projector.*`,
 * `model.layers.*`, `lm_head.weight`, `model.vision_tower.vision_tower.vision_model.*`,
 * `model.mm_projector.*`).  Replaces `load_state_dict` (build_sam.py:104-107,
 * eval_referseg.py:70-88).  `ptr` may be host or device memory (`is_device`); the
 * library keeps its own copy.  Unknown names are ignored and counted.
 */
int anyref_set_weight(anyref_handle* h, const char* name, const void* ptr, int is_device,
                      int dtype, const int64_t* shape, int ndim);
/* Pack / fuse / convert the weights for the chosen mode; reports missing tensors. */
int anyref_finalize(anyref_handle* h);

/*
 * AnyRefForCausalLM.generate (model/anyref.py:647-822).
 *   clip_images  f32 [B,3,clip_image,clip_image]  dev
 *   sam_images   f32 [B,3,sam_img,sam_img]        dev
 *   input_ids    i64 [B,Lmax] host; negative ids are placeholders (-200 image: expands 1->n_patches)
 *   lens         i32 [B] host: true prompt length of each row (no padding semantics: every row
 *                is decoded exactly as the reference decodes a batch of one)
 *   extra_embeds f32 [n_extra, llm_dim] dev or NULL: rows that REPLACE token embeddings 1:1
 *                (projected audio / reference-image features, anyref.py:663-702)
 *   extra_slots  i32 [n_extra,2] host: (batch row, position in input_ids) of each extra row
 *   resized_hw / orig_hw  i32 [B,2] host: sam_resized_sizes and (height,width) (anyref.py:814-818)
 *   max_new_tokens, eos_token_id (<0 disables EOS)
 * outputs
 *   out_ids   i64 [B, Lmax+max_new_tokens] host, out_lens i32 [B] host
 *   out_nseg  i32 [B] host: number of [SEG] tokens found per image (anyref.py:723-726)
 *   out_masks f32 dev: mask logits; image b, seg j at offset mask_offsets[b] + j*H_b*W_b,
 *             capacity `out_masks_cap` floats (error if exceeded); mask_offsets i64 [B] host
 *   out_low   f32 dev or NULL: low-res logits [B, max_seg, 4g, 4g] (what a DP driver all-gathers)
 *   out_hidden f32 dev or NULL: [B, llm_max_seq, llm_dim] last-layer post-norm states
 *             (`outputs.hidden_states[-1]`, anyref.py:718)
 */
int anyref_generate(anyref_handle* h, void* stream, const float* clip_images, const float* sam_images,
                    const int64_t* input_ids, const int32_t* lens, int B, int Lmax,
                    const float* extra_embeds, const int32_t* extra_slots, int n_extra,
                    const int32_t* resized_hw, const int32_t* orig_hw, int max_new_tokens,
                    int eos_token_id, int64_t* out_ids, int32_t* out_lens, int32_t* out_nseg,
                    float* out_masks, int64_t out_masks_cap, int64_t* mask_offsets, float* out_low,
                    float* out_hidden);

/*
 * Teacher-forced path of `model_forward_new` (model/anyref.py:239-430): same arithmetic with the
 * ids given, no decode loop.  [SEG] positions are searched in input_ids itself and the hidden
 * state at pos-1+255 is used (anyref.py:273-282).  `out_logits` f32 dev or NULL:
 * [B, Smax, vocab] LM logits for the caller's CE loss (losses stay in Python, anyref.py:19-68).
 */
int anyref_forward_teacher(anyref_handle* h, void* stream, const float* clip_images,
                           const float* sam_images, const int64_t* input_ids, const int32_t* lens,
                           int B, int Lmax, const float* extra_embeds, const int32_t* extra_slots,
                           int n_extra, const int32_t* rephrase_start, const int32_t* resized_hw,
                           const int32_t* orig_hw, int32_t* out_nseg, float* out_masks,
                           int64_t out_masks_cap, int64_t* mask_offsets, float* out_low,
                           float* out_hidden, float* out_logits);

/* ---- stage entry points (each is one reference function; used by the parity tests) ---- */

/* LLaVA encode_images = CLIP hidden_states[-2][:,1:] -> mm_projector (anyref.py:334).
 * out f32 [B, n_patches, llm_dim] dev; clip_feat f32 [B,n_patches,clip_dim] dev or NULL. */
int anyref_encode_images(anyref_handle* h, void* stream, const float* clip_images, int B, float* out,
                         float* clip_feat);
/* ImageEncoderViT.forward (image_encoder.py:110-125). out f32 [B, g*g, out_chans] dev (NHWC tokens;
 * the reference's NCHW [B,256,g,g] is this transposed). */
int anyref_sam_encode(anyref_handle* h, void* stream, const float* sam_images, int B, float* out);
/* prompt_encoder(text) + mask_decoder.predict_masks + postprocess_masks for one image
 * (anyref.py:797-819).  image_emb f32 [g*g, out_chans] dev, pred_emb f32 [n,out_dim] dev.
 * masks4 f32 [n,4,4g,4g] dev or NULL, iou f32 [n,4] dev or NULL,
 * out_masks f32 [n,H,W] dev or NULL (needs resized_hw, orig_hw i32[2] host). */
int anyref_mask_decode(anyref_handle* h, void* stream, const float* image_emb, const float* pred_emb,
                       int n, float* masks4, float* iou, const int32_t* resized_hw,
                       const int32_t* orig_hw, float* out_masks);
/* LLaMA stack on given input embeddings (HF LlamaModel; call sites anyref.py:341-354).
 * embeds f32 [B,S,dim] dev, lens i32[B] host.  hidden f32 [B,S,dim] (post final norm),
 * logits f32 [B,S,vocab] or NULL, attn_row f32 [B,S] or NULL = head-mean last-layer attention of
 * query `attn_q[b]` (i32[B] host) over all keys (anyref.py:748-749). */
int anyref_llm_forward(anyref_handle* h, void* stream, const float* embeds, const int32_t* lens, int B,
                       int S, float* hidden, float* logits, const int32_t* attn_q, float* attn_row);

/* ImageBindModel.get_audio_feature, second return value (model/ImageBind/models/imagebind_model.py:477-511; call
 * sites anyref.py:313-315, :670-672), for handles created with aud_blocks > 0 and given the
 * `model.audio_encoder.*` weights: mel f32 dev [n, 1, aud_mel, aud_len] (n = clips, <= aud_clips * max_batch)
 * -> conv stem (kernel aud_kernel, stride aud_stride, no bias) + LayerNorm -> [CLS] + pos_embed -> aud_blocks
 * pre-LN blocks (MHA with add_bias_kv, GELU MLP x4) -> LayerNorm -> CLS -> Linear(aud_dim, audio_dim, no bias)
 * -> L2-normalise x min(exp(log_logit_scale), 100) -> emb f32 dev [n, audio_dim]. */
int anyref_audio_encode(anyref_handle* h, void* stream, const float* mel, int n, float* emb);

/* audio_projector = Linear(audio_dim, llm_dim) on ImageBind audio embeddings (anyref.py:161,673).
 * audio_emb f32 [n, audio_dim] dev (the ImageBind trunk itself stays a PyTorch-ROCm step) ->
 * out f32 [n, llm_dim] dev, ready to be passed as `extra_embeds`. */
int anyref_project_audio(anyref_handle* h, void* stream, const float* audio_emb, int n, float* out);

/*
 * The glue alone: everything `generate` does after `super().generate` returns (model/anyref.py:718-822) or
 * `model_forward_new` does around `super().forward` (:273-282, :356-430), on caller-provided LLM outputs --
 * [SEG] search, hidden-row gather with the reference's hard-coded image offset (+255 = n_patches - 1),
 * rephrase, text_hidden_fcs, SAM image encoder, per-image prompt encoder -> mask decoder -> postprocess.
 * This is the entry the parity tests hold against fixtures made by running the reference's own lines on
 * canned LLM outputs (tests/golden/make_golden_glue.py).
 *   ids        i64 host [B, Lmax]: `outputs.sequences` (teacher = 0) or `input_ids` (teacher = 1);
 *              a [SEG] at index p >= 1 takes hidden row p - 1 + 255 (which is `where(ids[:,1:]) + 255`
 *              of :723-726,:758 and `pos - 1 + 255` of :282)
 *   ids_lens   i32 host [B]
 *   ref_pos    i32 host [B] or NULL: rephrase start = ref_pos[b] - 1 + 255 (teacher = 0: the prompt length,
 *              :745; teacher = 1: `where(labels > 0)[0][0]`, :378)
 *   hidden     f32 dev [B, hidden_rows, llm_dim] = `hidden_states[-1]`
 *   attn_mean  f32 dev [B, hidden_rows, hidden_rows] head-mean `attentions[-1]`, needed iff rephrase_weight > 0
 *   outputs as anyref_generate.
 */
int anyref_seg_tail(anyref_handle* h, void* stream, const float* sam_images, const int64_t* ids,
                    const int32_t* ids_lens, const int32_t* ref_pos, int B, int Lmax, int teacher,
                    const float* hidden, int hidden_rows, const float* attn_mean, const int32_t* resized_hw,
                    const int32_t* orig_hw, int32_t* out_nseg, float* out_masks, int64_t out_masks_cap,
                    int64_t* mask_offsets, float* out_low);

/* Change the inclusive [SEG] id range after creation (`seg_token_idx` kwarg, anyref.py:197-200). */
int anyref_set_seg_range(anyref_handle* h, int lo, int hi);

/* Two-stream overlap of the SAM image encoder with the LLM decode (default 1).  0 keeps the whole
 * call on the caller's stream (used by bench.py to time kernels without a co-running stream). */
int anyref_set_overlap(anyref_handle* h, int on);

/* Early [SEG] masks (default 1; effective with overlap on, batch 1, rephrase_weight 0, no [SEG] in the prompt):
 * anyref_generate queues the hand-off MLP, mask decoder and postprocess of a generated [SEG] on the side stream the
 * moment the token is read, under the remaining decode steps (the reference runs them after generate() returns,
 * anyref.py:718-822; same arithmetic per prompt, one prompt per mask-decoder call).  0: all masks after the loop. */
int anyref_set_early_tail(anyref_handle* h, int on);

/* extra_embeds produced on ANOTHER stream (e.g. the ImageBind audio trunk + audio_projector, `anyref_audio_encode` /
 * `anyref_project_audio` on a stream of the caller's): the next anyref_generate / anyref_forward_teacher on this handle makes
 * its own stream wait for `event` (a hipEvent_t recorded behind that work) just before the splice -- the first and only
 * place the rows are read -- so the trunk runs beside the CLIP tower instead of in front of the call.  One-shot: cleared
 * by the call that consumes it; NULL cancels.  The event must stay alive until that call has been queued. */
int anyref_set_extra_event(anyref_handle* h, void* event);

/* hipGraph replay of the greedy decode step (default 1): one step is ~170 launches with fixed
 * arguments (position / next token live on the device), captured once per batch size.  0 launches
 * them eagerly; the per-kernel profiler below always runs eagerly. */
int anyref_set_graphs(anyref_handle* h, int on);

/* CU share of the side stream (generate, batch 1; default 128 workgroups over 6 steps): while the SAM encoder co-runs
 * with the decode loop, its GEMM / attention launches are capped at `wgs` workgroups (each walks several output
 * tiles), so that the decode GEMVs -- whose throughput is proportional to the CUs they get -- keep CUs of their own
 * instead of queueing behind 256 resident MFMA workgroups; the encoder's blocks are queued ceil(depth / steps) per
 * decode step, and what is left when the loop ends runs uncapped.  wgs = 0: uncapped, queued whole at the fork
 * (the behaviour for more than 4 images per call; 2 - 4 images are fed the same way at a share of their own, 160
 * workgroups over 3 steps, which this call does not change).  steps <= 0 keeps the current value.  Results are
 * bit-identical either way. */
int anyref_set_side_share(anyref_handle* h, int wgs, int steps);

/*
 * Per-kernel timing for the measurement harness (bench.py "roofline"): when enabled, every GEMM /
 * GEMV / attention launch is bracketed by a hipEvent pair on its launch stream.  After the caller
 * has synchronised the stream, anyref_profile_collect() books the elapsed times;
 * anyref_profile_read(idx) returns tag, summed ms, launch count and summed algorithmic FLOPs /
 * bytes (returns -1 past the last tag).
 */
int anyref_profile_enable(anyref_handle* h, int on);
/* Restrict timing to one tag (NULL = all) and to every `sample_every`-th launch of a tag, so the
 * event pairs do not perturb the timed region they measure. */
int anyref_profile_config(anyref_handle* h, const char* only_tag, int sample_every);
int anyref_profile_collect(anyref_handle* h);
/* Measures what an event bracket adds to the kernel inside it on `stream` (tiny-kernel differencing;
 * synchronises the stream) and takes it off every bracket booked afterwards, so a tag's time is the
 * kernel's own duration (checked against rocprofv3 --kernel-trace in profiles/).  *overhead_us
 * receives the value. */
int anyref_profile_calibrate(anyref_handle* h, void* stream, double* overhead_us);
int anyref_profile_read(anyref_handle* h, int idx, char* name, int cap, double* ms, int64_t* count,
                        double* flops, double* bytes);

/*
 * Kernel-side timestamps for the measurement harness (bench.py "roofline", in situ): hipEvent brackets cannot be
 * placed inside a replayed hipGraph and rocprofv3 serialises the call's two streams, so neither sees a decode GEMV
 * as it runs in production.  With stamps enabled every decode-GEMV workgroup writes {earliest wave start, latest
 * wave end} of the 100 MHz wall clock into its own slot (plain stores); a launch's duration is max(end) - min(start)
 * over its workgroups.  Works inside graph replay (the captured step is re-captured with stamp slots addressed by a
 * device-side replay counter) and beside the second stream.  Costs one uniform branch when off.
 * anyref_stamps_collect() (after the caller has synchronised the device) reduces the slots and returns the number of
 * launches recorded since enable / the last collect, in start-time order; anyref_stamps_read(idx) returns one launch:
 * tag (the rocprofv3-visible instantiation, as in anyref_profile_read), start / end in microseconds relative to the
 * first launch, algorithmic bytes, and the replay index of its captured step (-1: eager launch).
 */
int anyref_stamps_enable(anyref_handle* h, int on);
int anyref_stamps_collect(anyref_handle* h, int64_t* count);
/* launches that went UNSTAMPED since enable / the last collect -- a step larger than the per-step record, or graph replays past
 * the epoch capacity (48): read it BEFORE anyref_stamps_collect (which resets it); non-zero means per-step averages built on the
 * collected rows miss launches */
int anyref_stamps_dropped(anyref_handle* h, int64_t* count);
int anyref_stamps_read(anyref_handle* h, int64_t idx, char* name, int cap, double* t0_us, double* t1_us,
                       double* bytes, int* epoch);
/* the same launch's spread over its workgroups: last start - first start (dispatch ramp), last end - first end (tail),
 * and the median workgroup's own start-to-end span */
int anyref_stamps_spread(anyref_handle* h, int64_t idx, double* start_spread_us, double* end_spread_us,
                         double* wg_median_us);

/* Bytes of HBM the handle holds (weights + workspaces), for sizing reports. */
int64_t anyref_device_bytes(anyref_handle* h);
/* Name of the compute mode's arithmetic ("f32" / "bf16"). */
const char* anyref_mode_name(anyref_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* ANYREF_HIP_H */

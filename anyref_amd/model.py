"""Python surface of the backend: a mirror of the reference's `AnyRefForCausalLM`
(`model/anyref.py:182-237,647-822`) over the C-ABI library.

What is preserved: the `generate(...)` and `forward(**kwargs)` signatures and
argument meaning, the `[SEG]`-token -> SAM prompt hand-off, the state_dict weight names, the
`.config.{eos,bos,pad}_token_id` attributes and the no-op plumbing calls the eval scripts make
(`get_model()`, `initialize_vision_modules`, `initialize_anyref_modules`,
`resize_token_embeddings`, `.eval()`, `.cuda()`, `.to()`, `.half()`), so
`eval_referseg.py:137` / `eval_avs_object.py:137` can drive this class unchanged.

Deviations, both deliberate:
  * `generate` always returns the 3-tuple `(output_ids, pred_masks, (None, None, None))`; the
    reference returns a 2-tuple on its success path (`anyref.py:822`) but both north-star callers
    unpack three values (`eval_referseg.py:136`, `eval_avs_object.py:137`).
  * batched calls decode every row exactly as a batch of one (per-row lengths, no left-pad
    position shift); left-padded `input_ids` + `attention_masks` are accepted and un-padded here.

PyTorch is only the owner of device memory and streams; all arithmetic runs in libanyref_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import _lib
from .config import (AnyRefConfig, IMAGE_TOKEN_INDEX, AUDIO_REF_INDEX, IMG_REF_INDEX, AUDIO_REF_NUM)

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _c_config(cfg: AnyRefConfig, mode: int, max_batch: int, max_seg: int) -> _lib.AnyrefConfig:
    c = _lib.AnyrefConfig()
    c.abi_version = _lib.ABI_VERSION
    c.mode = mode
    cl, l, s = cfg.clip, cfg.llm, cfg.sam
    c.clip_image, c.clip_patch, c.clip_dim, c.clip_heads = cl.image_size, cl.patch, cl.dim, cl.heads
    c.clip_layers_run, c.clip_mlp, c.clip_eps = cl.layers_run, cl.mlp, cl.eps
    c.llm_vocab, c.llm_dim, c.llm_heads, c.llm_layers = l.vocab, l.dim, l.heads, l.layers
    c.llm_mlp, c.llm_max_seq, c.llm_rms_eps, c.llm_rope_theta = l.mlp, l.max_seq, l.rms_eps, l.rope_theta
    c.sam_img, c.sam_patch, c.sam_dim, c.sam_depth = s.img_size, s.patch, s.dim, s.depth
    c.sam_heads, c.sam_mlp_ratio, c.sam_window = s.heads, s.mlp_ratio, s.window
    assert len(s.global_idx) <= 8
    c.sam_n_global = len(s.global_idx)
    for i, g in enumerate(s.global_idx):
        c.sam_global_idx[i] = g
    c.sam_out_chans, c.dec_heads, c.dec_mlp, c.dec_depth = s.out_chans, s.dec_heads, s.dec_mlp, s.dec_depth
    c.num_mask_tokens = s.num_mask_tokens
    c.out_dim, c.audio_dim = cfg.out_dim, cfg.audio_dim
    c.seg_lo, c.seg_hi = cfg.seg_range()
    c.rephrase_weight = float(cfg.rephrase_weight)
    c.max_batch, c.max_seg = max_batch, max_seg
    at = cfg.audio_trunk
    if at is not None:
        c.aud_dim, c.aud_blocks, c.aud_heads, c.aud_mel = at.dim, at.blocks, at.heads, at.mel_bins
        c.aud_len, c.aud_kernel, c.aud_stride, c.aud_clips = at.target_len, at.kernel, at.stride, at.clips
    return c


_NEEDS_HANDLE = frozenset({
    "generate", "model_forward_new", "forward", "__call__", "encode_images", "sam_encode", "mask_decode", "llm_forward",
    "seg_tail", "postprocess", "audio_encode", "device_bytes", "set_overlap", "set_early_tail", "set_graphs", "profile_enable", "profile_read",
    "stamps_enable", "stamps_read", "set_side_share"})


class AnyRefForCausalLM:
    """MI355X-native stand-in for `model.anyref.AnyRefForCausalLM` (inference surface)."""

    def __init__(self, cfg: AnyRefConfig, mode: str = "perf", device: int = 0, max_batch: int = 1,
                 max_seg: int = 4, audio_encoder=None, defer: bool = False, **kwargs):
        # constructor kwargs of the reference (anyref.py:188-209) that shape the path
        if "seg_token_idx" in kwargs:
            cfg.seg_token_idx = kwargs.pop("seg_token_idx")
        if "rephrase_weight" in kwargs:
            cfg.rephrase_weight = kwargs.pop("rephrase_weight")
        if "out_dim" in kwargs:
            cfg.out_dim = kwargs.pop("out_dim")
        self.ce_loss_weight = kwargs.pop("ce_loss_weight", 1.0)
        self.dice_loss_weight = kwargs.pop("dice_loss_weight", 0.5)
        self.bce_loss_weight = kwargs.pop("bce_loss_weight", 2.0)
        self.vision_pretrained = kwargs.pop("vision_pretrained", None)
        self.add_audio_encoder = bool(kwargs.pop("add_audio_encoder", False))
        self.imagebind_ckpt = kwargs.pop("imagebind_ckpt", "model/ImageBind/imagebind_huge.pth")
        self.cfg = cfg
        self.mode = {"parity": _lib.MODE_PARITY, "perf": _lib.MODE_PERF, "perf_fp8w": _lib.MODE_PERF_FP8W,
                     "parity16": _lib.MODE_PARITY16}[mode]
        self.mode_name = mode
        self.device_index = device
        self.device = torch.device("cuda", device)
        self.max_batch, self.max_seg = max_batch, max_seg
        self.audio_encoder = audio_encoder      # PyTorch-ROCm ImageBind audio trunk (anyref_amd.audio)
        self.config = SimpleNamespace(eos_token_id=cfg.eos_token_id, bos_token_id=cfg.bos_token_id,
                                      pad_token_id=cfg.pad_token_id, hidden_size=cfg.llm.dim,
                                      vocab_size=cfg.llm.vocab, use_cache=True, out_dim=cfg.out_dim,
                                      mm_vision_tower=kwargs.pop("vision_tower", "openai/clip-vit-large-patch14"))
        self.h = None
        self._finalized = False
        # `generate` returns the 3-tuple the two north-star callers unpack (eval_referseg.py:136, eval_avs_object.py:137).
        # The reference's success path itself returns 2 values (anyref.py:822), which is what eval_refer_inv.py:135 and
        # eval_coco20i.py:142 unpack: set `model.success_arity = 2` for those scripts.
        self.success_arity = 3
        # `from_pretrained` flow: weights are gathered on the host first (base checkpoint, CLIP tower, SAM file,
        # token-row resize, LoRA merge all happen BEFORE `.cuda()` in the callers) and go to HBM in one build
        self._host_sd: Optional[Dict[str, torch.Tensor]] = {} if defer else None
        if not defer:
            self._create()

    def _create(self):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("AnyRefForCausalLM needs an MI355X: the HIP backend has no CPU fallback")
        h = C.c_void_p()
        cc = _c_config(self.cfg, self.mode, self.max_batch, self.max_seg)
        rc = self.lib.anyref_create(C.byref(cc), self.device_index, C.byref(h))
        if rc != 0:
            raise RuntimeError("anyref_create: " + self.lib.anyref_last_error(None).decode())
        self.h = h

    @property
    def n_img(self) -> int:
        return self.cfg.clip.n_patches

    def _ensure_built(self):
        """Deferred construction: create the handle for the FINAL config and move the gathered weights to HBM."""
        if self.h is not None:
            return
        from .checkpoint import missing_for
        sd = self._host_sd or {}
        audio = any(k.startswith("model.audio_projector.") for k in sd)
        miss = missing_for(self.cfg, sd.keys(), audio)
        if miss:
            raise RuntimeError(f"{len(miss)} weights of the inference path were never loaded, e.g. {miss[:4]} "
                               "(from_pretrained -> initialize_vision_modules -> initialize_anyref_modules -> adapter)")
        self.config.vocab_size = self.cfg.llm.vocab
        self._create()
        self.load_state_dict(sd)
        self._host_sd = None

    # ---- lifetime ------------------------------------------------------------------------
    def __del__(self):
        h = getattr(self, "h", None)
        if h:
            try:
                self.lib.anyref_destroy(h)
            except Exception:
                pass
            self.h = None

    def __getattribute__(self, name):
        # any method that talks to the C-ABI handle first makes sure a deferred build has happened
        if name in _NEEDS_HANDLE:
            object.__getattribute__(self, "_ensure_built")()
        return object.__getattribute__(self, name)

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.lib.anyref_last_error(self.h).decode()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- weights (reference state_dict names) --------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False):
        if self.h is None:                       # deferred build: keep gathering on the host
            self._host_sd.update(sd)
            return [], []
        for name, t in sd.items():
            if t.dtype not in _DT:
                t = t.float()
            t = t.contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            self._check(self.lib.anyref_set_weight(self.h, name.encode(), _ptr(t), int(t.is_cuda), _DT[t.dtype],
                                                   shape, t.dim()), f"set_weight({name})")
        self._check(self.lib.anyref_finalize(self.h), "finalize")
        self._finalized = True
        return [], []

    @classmethod
    def from_state_dict(cls, cfg: AnyRefConfig, sd, **kw) -> "AnyRefForCausalLM":
        if cfg.audio_trunk is None and any(k.startswith("model.audio_encoder.modality_trunks.audio.") for k in sd):
            from .config import AudioTrunkConfig          # the checkpoint carries the ImageBind trunk: run it in HIP
            cfg.audio_trunk = AudioTrunkConfig(
                dim=sd["model.audio_encoder.modality_heads.audio.0.weight"].shape[0],
                blocks=1 + max(int(k.split("blocks.")[1].split(".")[0]) for k in sd
                               if k.startswith("model.audio_encoder.modality_trunks.audio.blocks.")))
        m = cls(cfg, **kw)
        m.load_state_dict(sd)
        return m

    # ---- construction path of the reference's callers (eval_referseg.py:62-88) --------------
    @classmethod
    def from_pretrained(cls, model_version: str, torch_dtype=None, mode: Optional[str] = None, device: int = 0,
                        max_batch: int = 1, max_seg: int = 4, max_seq: int = 1024, **model_args):
        """`AnyRefForCausalLM.from_pretrained(model_version, torch_dtype=..., **model_args)` (eval_referseg.py:70-72)
        on an HF `save_pretrained` directory (sharded safetensors / bin): config.json -> LLM shape, every tensor ->
        host state dict under the reference's names.  `model_args` are the reference constructor's kwargs
        (anyref.py:188-209: seg_token_idx, out_dim, vision_pretrained, add_audio_encoder, rephrase_weight, ...).
        The handle is built at `.cuda()` / first use, after the caller's `initialize_*`, `resize_token_embeddings`
        and adapter merge.  `torch_dtype` is accepted for signature parity; the arithmetic type is `mode`
        ("perf" = bf16 storage, fp32 accumulate; "parity" = fp32), default perf."""
        import json
        import os
        from .checkpoint import read_hf_dir, llm_config_from_hf
        hf = json.load(open(os.path.join(model_version, "config.json")))
        cfg = AnyRefConfig(llm=llm_config_from_hf(hf, max_seq))
        for k in ("eos_token_id", "bos_token_id", "pad_token_id"):
            if hf.get(k) is not None:
                setattr(cfg, k, int(hf[k]))
        model_args.pop("train_mask_decoder", None)
        if "mm_vision_tower" in hf:
            model_args.setdefault("vision_tower", hf["mm_vision_tower"])
        m = cls(cfg, mode=mode or "perf", device=device, max_batch=max_batch, max_seg=max_seg, defer=True, **model_args)
        m._host_sd.update(read_hf_dir(model_version))
        # a merged checkpoint (merge_lora.py:62 `save_pretrained`) already carries the towers: size them from it
        if "train_mask_decoder" in hf and m.vision_pretrained:
            m._adopt_sam_shape()
        return m

    def _adopt_sam_shape(self):
        from .checkpoint import sam_config_for
        s = self.cfg.sam
        self.cfg.sam = sam_config_for(self.vision_pretrained, img_size=s.img_size, patch=s.patch, window=s.window,
                                      out_chans=s.out_chans)

    def get_model(self):
        return self

    def get_vision_tower(self):
        return self

    def initialize_vision_modules(self, model_args=None, *_a, **_k):
        """LLaVA's `initialize_vision_modules` (eval_referseg.py:77): load the CLIP tower named by
        `config.mm_vision_tower` (a local HF directory) unless the checkpoint already carries it."""
        if self._host_sd is None:
            return None
        from .synth import CLIP_PREFIX
        if any(k.startswith(CLIP_PREFIX) for k in self._host_sd):
            return None
        from .checkpoint import load_clip_tower
        path = getattr(model_args, "mm_vision_tower", None) or self.config.mm_vision_tower
        self.cfg.clip, sd = load_clip_tower(path)
        self._host_sd.update(sd)
        return None

    def initialize_anyref_modules(self, config=None, *_a, **_k):
        """`initialize_anyref_modules` (anyref.py:96-161): SAM from `vision_pretrained` (variant by substring),
        `text_hidden_fcs` and, with `add_audio_encoder`, `audio_projector` freshly initialised as `nn.Linear` does
        (their trained values come with the adapter's `modules_to_save`), ImageBind trunk attached if its file exists."""
        if self._host_sd is None:
            return None
        import os
        from .checkpoint import load_sam
        from .synth import SAM_PREFIX
        sd = self._host_sd
        if self.vision_pretrained is None:
            raise ValueError("initialize_anyref_modules needs the `vision_pretrained` constructor kwarg (anyref.py:98-105)")
        self._adopt_sam_shape()
        if not any(k.startswith(SAM_PREFIX) for k in sd):
            sd.update(load_sam(self.vision_pretrained))
        H, out = self.cfg.llm.dim, self.cfg.out_dim

        def fresh(name, n_out, n_in):
            if name + ".weight" not in sd:
                lin = torch.nn.Linear(n_in, n_out)
                sd[name + ".weight"], sd[name + ".bias"] = lin.weight.detach(), lin.bias.detach()

        fresh("model.text_hidden_fcs.0.0", H, H)
        fresh("model.text_hidden_fcs.0.2", out, H)
        if self.add_audio_encoder:
            fresh("model.audio_projector", H, self.cfg.audio_dim)
            if self.audio_encoder is None:
                from .audio import ImageBindAudio
                self.audio_encoder = ImageBindAudio()
                if os.path.exists(self.imagebind_ckpt):
                    from .checkpoint import read_tensors
                    self.audio_encoder.load_reference_state_dict(read_tensors(self.imagebind_ckpt))
                else:
                    print("ImageBind audio encoder ckpt not found!")
        return None

    def resize_token_embeddings(self, n: int):
        if self._host_sd is None:
            if n != self.cfg.llm.vocab:
                raise ValueError(f"tokenizer has {n} tokens but the backend was built for vocab {self.cfg.llm.vocab}")
            return
        from .checkpoint import resize_token_rows
        resize_token_rows(self._host_sd, n)
        self.cfg.llm.vocab = n
        self.config.vocab_size = n

    def merge_adapter(self, adapter_dir: str):
        """`PeftModel.from_pretrained(model, dir).merge_and_unload()` (anyref_amd.peft_compat)."""
        if self._host_sd is None:
            raise RuntimeError("the LoRA merge happens on the host state dict: call it before .cuda() / first use")
        from .checkpoint import merge_lora
        self.adapter_stats = merge_lora(self._host_sd, adapter_dir)
        return self

    def host_state_dict(self):
        """The gathered weights before the build (deferred construction only)."""
        return self._host_sd

    def eval(self):
        return self

    def cuda(self, *_a):
        self._ensure_built()
        return self

    def half(self):
        return self

    def to(self, *_a, **_k):
        return self

    @property
    def device_bytes(self) -> int:
        return int(self.lib.anyref_device_bytes(self.h))

    def set_seg_token_idx(self, seg_token_idx):
        """`seg_token_idx` kwarg of the reference constructor (anyref.py:197-200), changeable later."""
        self.cfg.seg_token_idx = seg_token_idx
        if self.h is None:
            return
        lo, hi = self.cfg.seg_range()
        self._check(self.lib.anyref_set_seg_range(self.h, lo, hi), "set_seg_range")

    def set_overlap(self, on: bool):
        """SAM encoder on a second stream under the LLM decode (default) or everything on one stream."""
        self._check(self.lib.anyref_set_overlap(self.h, int(on)), "set_overlap")

    def set_early_tail(self, on: bool):
        """Masks of generated [SEG]s decoded on the side stream while the greedy loop goes on (default; batch 1)."""
        self._check(self.lib.anyref_set_early_tail(self.h, int(on)), "set_early_tail")

    def set_graphs(self, on: bool):
        """hipGraph replay of the greedy decode step (default) or eager launches."""
        self._check(self.lib.anyref_set_graphs(self.h, int(on)), "set_graphs")

    def set_side_share(self, wgs: int, steps: int = 0):
        """Workgroup cap of the SAM encoder's GEMM / attention launches while it co-runs with the decode loop (0: none),
        and the number of decode steps its blocks are spread over (0: keep)."""
        self._check(self.lib.anyref_set_side_share(self.h, int(wgs), int(steps)), "set_side_share")

    # ---- per-kernel timing for bench.py ------------------------------------------------------
    def profile_enable(self, on: bool, only_tag: Optional[str] = None, sample_every: int = 1):
        self._check(self.lib.anyref_profile_config(self.h, only_tag.encode() if only_tag else None, sample_every),
                    "profile_config")
        if on:  # what an empty event pair reads on this stream is taken off every bracket
            ov = C.c_double()
            self._check(self.lib.anyref_profile_calibrate(self.h, self._stream(), C.byref(ov)), "profile_calibrate")
            self.profile_overhead_us = ov.value
        self._check(self.lib.anyref_profile_enable(self.h, int(on)), "profile_enable")

    def profile_read(self):
        """-> {tag: dict(ms, count, flops, bytes)} of everything timed since profile_enable(True)."""
        torch.cuda.synchronize(self.device)
        self._check(self.lib.anyref_profile_collect(self.h), "profile_collect")
        out, i = {}, 0
        name = C.create_string_buffer(128)
        ms, cnt, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        while self.lib.anyref_profile_read(self.h, i, name, 128, C.byref(ms), C.byref(cnt), C.byref(fl),
                                           C.byref(by)) == 0:
            out[name.value.decode()] = dict(ms=ms.value, count=cnt.value, flops=fl.value, bytes=by.value)
            i += 1
        return out

    def stamps_enable(self, on: bool):
        """Kernel-side timestamps of the decode GEMVs (include/anyref_hip.h `anyref_stamps_*`): what bench.py's in-situ
        roofline reads -- production launch path (hipGraph replay, SAM encoder co-running), no event brackets."""
        torch.cuda.synchronize(self.device)
        self._check(self.lib.anyref_stamps_enable(self.h, int(on)), "stamps_enable")

    def stamps_read(self):
        """-> [dict(tag, t0_us, t1_us, bytes, epoch)] of every stamped launch since enable / the last read, by start time."""
        torch.cuda.synchronize(self.device)
        n = _lib._L()
        self._check(self.lib.anyref_stamps_dropped(self.h, C.byref(n)), "stamps_dropped")
        self.stamps_dropped = int(n.value)      # launches of this pass that went unstamped (record / epoch capacity): 0 or the rows lie
        self._check(self.lib.anyref_stamps_collect(self.h, C.byref(n)), "stamps_collect")
        name = C.create_string_buffer(128)
        t0, t1, by, ep = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        rows = []
        for i in range(n.value):
            if self.lib.anyref_stamps_read(self.h, i, name, 128, C.byref(t0), C.byref(t1), C.byref(by), C.byref(ep)) != 0:
                break
            a, b, c = C.c_double(), C.c_double(), C.c_double()
            self.lib.anyref_stamps_spread(self.h, i, C.byref(a), C.byref(b), C.byref(c))
            rows.append(dict(tag=name.value.decode(), t0_us=t0.value, t1_us=t1.value, bytes=by.value, epoch=ep.value,
                             start_spread_us=a.value, end_spread_us=b.value, wg_median_us=c.value))
        return rows

    def postprocess(self, low: torch.Tensor, resized_size, original_size) -> torch.Tensor:
        """`Sam.postprocess_masks` (sam.py:137-172) on low-res logits [n, 4g, 4g] -> [n, H, W]."""
        low = low.to(self.device, torch.float32).contiguous()
        n, lh, lw = low.shape
        out = torch.empty(n, int(original_size[0]), int(original_size[1]), device=self.device, dtype=torch.float32)
        if n:
            rc = self.lib.anyref_op_postprocess(self._stream(), _ptr(low), n, lh, lw, self.cfg.sam.img_size,
                                                int(resized_size[0]), int(resized_size[1]), int(original_size[0]),
                                                int(original_size[1]), _ptr(out))
            if rc != 0:
                raise RuntimeError("postprocess: " + self.lib.anyref_op_last_error().decode())
        return out

    # ---- helpers ---------------------------------------------------------------------------
    def _rows(self, input_ids: torch.Tensor, attention_masks: Optional[torch.Tensor], keep_out: Optional[list] = None,
              keep_in: Optional[list] = None):
        """-> (ids int64 host [B,Lmax] right-aligned rows, lens int32 [B]).

        `keep_out` receives the per-row index tensors that were kept; `keep_in` applies such a list instead of
        deriving one (labels must be cut exactly where their `input_ids` row was: a label row holds -100 / ids and
        never equals the pad id, so the pad rule below cannot be evaluated on it)."""
        ids = input_ids.detach().to("cpu", torch.long)
        if ids.dim() == 1:
            ids = ids[None]
        B, Lm = ids.shape
        rows = []
        pad = self.config.pad_token_id
        for b in range(B):
            r = ids[b]
            keep = torch.arange(Lm)
            if keep_in is not None:
                keep = keep_in[b]
            elif attention_masks is not None:
                keep = attention_masks[b].to("cpu").bool().nonzero().flatten()
            elif B > 1 and pad is not None:
                # the callers' batched path passes left-padded ids and NO mask (eval_referseg.py:124-137); the
                # collator's own mask is `input_ids.ne(pad_token_id)` (utils/coco_instance.py:142): strip the pad
                # runs at both ends of the row (pad = unk never opens or closes a prompt: BOS ... "ASSISTANT:")
                nz = (r != int(pad)).nonzero().flatten()
                keep = torch.arange(int(nz[0]), int(nz[-1]) + 1) if len(nz) else torch.arange(1)
            if keep_out is not None:
                keep_out.append(keep)
            rows.append(r[keep])
        lens = torch.tensor([len(r) for r in rows], dtype=torch.int32)
        out = torch.zeros(B, int(lens.max()), dtype=torch.long)
        for b, r in enumerate(rows):
            out[b, : len(r)] = r
        return out.contiguous(), lens

    def _extra_overlapped(self, ids, lens, audios, ref_feats):
        """`_extra` with the audio trunk + projector (launch-bound: ~1.3 ms of small kernels) on a stream of their own: the
        next C call waits for them only at the splice, so they run beside the CLIP tower instead of in front of it
        (anyref_set_extra_event).  Same kernels, same results."""
        # (HIP trunk inside the handle only: C4 41.1 -> 40.4 ms per image; the PyTorch-ROCm module's ~3 ms of host-side launches sit in
        #  front of the C call whatever stream they go to: 42.2 vs 42.5 ms)
        if audios is None or self.cfg.audio_trunk is None or self.device.type != "cuda" or os.environ.get("ANYREF_AUDIO_OVERLAP", "1") == "0":
            return self._extra(ids, lens, audios, ref_feats)
        main = torch.cuda.current_stream(self.device)
        if getattr(self, "_audio_stream", None) is None:
            self._audio_stream = torch.cuda.Stream(self.device)
            self._audio_event = torch.cuda.Event()
        side = self._audio_stream
        side.wait_stream(main)                      # the mel clips / features the caller queued on its stream
        with torch.cuda.stream(side):
            out = self._extra(ids, lens, audios, ref_feats)
            self._audio_event.record(side)
        if out[0] is not None:
            out[0].record_stream(main)
        self._check(self.lib.anyref_set_extra_event(self.h, C.c_void_p(self._audio_event.cuda_event)), "set_extra_event")
        return out

    def _extra(self, ids: torch.Tensor, lens, audios, ref_feats):
        """Projected audio / reference features -> (extra_embeds dev [n,H], slots host [n,2])."""
        embeds, slots = [], []
        B = ids.shape[0]
        if audios is not None:
            for b in range(B):
                a = audios[b] if isinstance(audios, (list, tuple)) else audios[b:b + 1]
                if a is None:
                    continue
                if a.dim() >= 4:     # raw mel clips [1,3,1,128,204] -> ImageBind embedding [3,1024]
                    if self.cfg.audio_trunk is not None:          # trunk inside the handle (HIP, f-4)
                        a = self.audio_encode(a)
                    elif self.audio_encoder is not None:           # PyTorch-ROCm module (north_star's default)
                        _, emb = self.audio_encoder.get_audio_feature(a.to(self.device).float())
                        a = emb[0]
                    else:
                        raise RuntimeError("raw audio given but neither a HIP trunk (cfg.audio_trunk + model.audio_encoder.* "
                                           "weights) nor an audio_encoder module (anyref_amd.audio) is attached")
                a = a.to(self.device, torch.float32).reshape(-1, self.cfg.audio_dim).contiguous()
                out = torch.empty(a.shape[0], self.cfg.llm.dim, device=self.device, dtype=torch.float32)
                self._check(self.lib.anyref_project_audio(self.h, self._stream(), _ptr(a), a.shape[0], _ptr(out)),
                            "project_audio")
                pos = (ids[b, : lens[b]] == AUDIO_REF_INDEX).nonzero().flatten().tolist()
                if len(pos) != out.shape[0]:
                    raise ValueError(f"row {b}: {len(pos)} audio placeholders but {out.shape[0]} audio features")
                embeds.append(out)
                slots += [(b, p) for p in pos]
        if ref_feats is not None:
            for b in range(B):
                r = ref_feats[b]
                if r is None:
                    continue
                r = r.to(self.device, torch.float32).reshape(-1, self.cfg.llm.dim).contiguous()
                pos = (ids[b, : lens[b]] == IMG_REF_INDEX).nonzero().flatten().tolist()
                if len(pos) != r.shape[0]:
                    raise ValueError(f"row {b}: {len(pos)} image-ref placeholders but {r.shape[0]} features")
                embeds.append(r)
                slots += [(b, p) for p in pos]
        if not embeds:
            return None, None, 0
        e = torch.cat(embeds, 0).contiguous()
        s = torch.tensor(slots, dtype=torch.int32).contiguous()
        return e, s, e.shape[0]

    def _ref_features(self, ref_images, B: int):
        """`ref_images` branch of generate (anyref.py:681-702) / forward (:319-339): every reference image goes
        through `encode_images` (a second CLIP pass) and ends as IMG_REF_NUM rows that replace the prompt's
        `<img_ref>` placeholders.  The reference pools a TENSOR batch 256 -> 16 -> 4 in the glue (:695-700) but hands
        LIST items to the absent llava layer unpooled (:691-692); this backend's reading of that layer (INFERRED,
        documented in DESIGN.md) pools them the way `model_forward_new` does (:335-338), so both forms take the
        same kernel (`anyref_op_pool_ref_tokens`).  1-D items are RoI coordinates (:688-689): need the llava layer."""
        if ref_images is None:
            return None
        from .config import IMG_REF_NUM
        if isinstance(ref_images, (list, tuple)):
            items = list(ref_images)
        else:
            if ref_images.dim() != 4 or ref_images.shape[0] != B:          # anyref.py:695,701-702
                raise NotImplementedError("a ref_images tensor must be [batch, 3, S, S]")
            items = list(ref_images)
        feats = []
        for r in items:
            if r is None:
                feats.append(None)
                continue
            if r.dim() == 1:
                raise NotImplementedError("RoI-coordinate reference (anyref.py:688-689) needs the absent llava layer")
            f = self.encode_images(r[None].float())                      # [1, 256, H]
            out = torch.empty(1, IMG_REF_NUM, f.shape[2], device=self.device, dtype=torch.float32)
            rc = self.lib.anyref_op_pool_ref_tokens(self._stream(), _ptr(f), 1, f.shape[1], f.shape[2], IMG_REF_NUM, _ptr(out))
            if rc != 0:
                raise RuntimeError("pool_ref_tokens: " + self.lib.anyref_op_last_error().decode())
            feats.append(out[0])
        return feats

    # ---- stage calls (used by tests and by callers that want the pieces) -------------------
    def encode_images(self, clip_images: torch.Tensor, return_clip: bool = False):
        x = clip_images.to(self.device, torch.float32).contiguous()
        B = x.shape[0]
        out = torch.empty(B, self.n_img, self.cfg.llm.dim, device=self.device, dtype=torch.float32)
        cf = torch.empty(B, self.n_img, self.cfg.clip.dim, device=self.device, dtype=torch.float32) if return_clip else None
        self._check(self.lib.anyref_encode_images(self.h, self._stream(), _ptr(x), B, _ptr(out), _ptr(cf)),
                    "encode_images")
        return (out, cf) if return_clip else out

    def audio_encode(self, mel: torch.Tensor) -> torch.Tensor:
        """`get_audio_feature(...)[1]` (imagebind_model.py:477-511) in HIP: mel [..., 1, mel_bins, target_len] (any
        leading clip / batch dims) -> embedding [n_clips, audio_dim], L2-normalised x logit scale."""
        at = self.cfg.audio_trunk
        if at is None:
            raise RuntimeError("the handle was created without an audio trunk (cfg.audio_trunk is None)")
        x = mel.to(self.device, torch.float32).reshape(-1, 1, at.mel_bins, at.target_len).contiguous()
        out = torch.empty(x.shape[0], self.cfg.audio_dim, device=self.device, dtype=torch.float32)
        self._check(self.lib.anyref_audio_encode(self.h, self._stream(), _ptr(x), x.shape[0], _ptr(out)), "audio_encode")
        return out

    def sam_encode(self, sam_images: torch.Tensor) -> torch.Tensor:
        """-> [B, 256, g, g] (the reference's NCHW layout)."""
        x = sam_images.to(self.device, torch.float32).contiguous()
        B, g = x.shape[0], self.cfg.sam.grid
        out = torch.empty(B, g * g, self.cfg.sam.out_chans, device=self.device, dtype=torch.float32)
        self._check(self.lib.anyref_sam_encode(self.h, self._stream(), _ptr(x), B, _ptr(out)), "sam_encode")
        return out.view(B, g, g, -1).permute(0, 3, 1, 2)

    def mask_decode(self, image_embedding: torch.Tensor, pred_embeddings: torch.Tensor, resized_size=None,
                    original_size=None):
        """image_embedding [256,g,g] or [1,256,g,g]; pred_embeddings [n,256] ->
        dict(masks4 [n,4,4g,4g], iou [n,4], masks [n,H,W] if sizes given)."""
        g, Cc = self.cfg.sam.grid, self.cfg.sam.out_chans
        e = image_embedding.reshape(Cc, g * g).t().to(self.device, torch.float32).contiguous()
        p = pred_embeddings.to(self.device, torch.float32).reshape(-1, self.cfg.out_dim).contiguous()
        n, nt = p.shape[0], self.cfg.sam.num_mask_tokens
        m4 = torch.empty(n, nt, 4 * g, 4 * g, device=self.device, dtype=torch.float32)
        iou = torch.empty(n, nt, device=self.device, dtype=torch.float32)
        out = rs = os_ = None
        if original_size is not None:
            out = torch.empty(n, int(original_size[0]), int(original_size[1]), device=self.device, dtype=torch.float32)
            rs = (C.c_int32 * 2)(int(resized_size[0]), int(resized_size[1]))
            os_ = (C.c_int32 * 2)(int(original_size[0]), int(original_size[1]))
        self._check(self.lib.anyref_mask_decode(self.h, self._stream(), _ptr(e), _ptr(p), n, _ptr(m4), _ptr(iou),
                                                rs, os_, _ptr(out)), "mask_decode")
        return dict(masks4=m4, iou=iou, masks=out)

    def llm_forward(self, embeds: torch.Tensor, lens: Optional[Sequence[int]] = None, want_logits: bool = False,
                    attn_q: Optional[Sequence[int]] = None):
        x = embeds.to(self.device, torch.float32).contiguous()
        B, S, H = x.shape
        ln = (C.c_int32 * B)(*([S] * B if lens is None else [int(v) for v in lens]))
        hidden = torch.empty(B, S, H, device=self.device, dtype=torch.float32)
        logits = torch.empty(B, S, self.cfg.llm.vocab, device=self.device, dtype=torch.float32) if want_logits else None
        aq = ar = None
        if attn_q is not None:
            aq = (C.c_int32 * B)(*[int(v) for v in attn_q])
            ar = torch.zeros(B, S, device=self.device, dtype=torch.float32)
        self._check(self.lib.anyref_llm_forward(self.h, self._stream(), _ptr(x), ln, B, S, _ptr(hidden), _ptr(logits),
                                                aq, _ptr(ar)), "llm_forward")
        return dict(hidden=hidden, logits=logits, attn_row=ar)

    def seg_tail(self, sam_images, ids, ids_lens, ref_pos, hidden, attn_mean, sam_resized_sizes, height, width,
                 teacher: bool = False):
        """The glue alone (anyref.py:718-822 / :273-282,:356-430) on caller-provided LLM outputs:
        ids [B,L] (`outputs.sequences` or, teacher=True, `input_ids`), hidden [B,rows,H] = `hidden_states[-1]`,
        attn_mean [B,rows,rows] head-mean `attentions[-1]` (only with rephrase_weight > 0), ref_pos [B] = prompt
        length (generate) or `where(labels > 0)[0][0]` (forward).  -> (pred_masks list | None, nseg)."""
        ids = ids.detach().to("cpu", torch.long).contiguous()
        B, Lmax = ids.shape
        lens = torch.tensor([int(v) for v in ids_lens], dtype=torch.int32)
        rp = None if ref_pos is None else torch.tensor([int(v) for v in ref_pos], dtype=torch.int32)
        hid = hidden.to(self.device, torch.float32).contiguous()
        att = None if attn_mean is None else attn_mean.to(self.device, torch.float32).contiguous()
        sam = sam_images.to(self.device, torch.float32).contiguous()
        height, width = [int(h) for h in height], [int(w) for w in width]
        rs = torch.tensor([[int(a), int(b)] for a, b in sam_resized_sizes], dtype=torch.int32).contiguous()
        os_ = torch.tensor([[h, w] for h, w in zip(height, width)], dtype=torch.int32).contiguous()
        nseg = torch.zeros(B, dtype=torch.int32)
        offs = torch.zeros(B, dtype=torch.long)
        cap = sum(self.max_seg * h * w for h, w in zip(height, width))
        out_masks = torch.empty(cap, device=self.device, dtype=torch.float32)
        self._check(self.lib.anyref_seg_tail(self.h, self._stream(), _ptr(sam), _ptr(ids), _ptr(lens), _ptr(rp), B, Lmax,
                                             int(teacher), _ptr(hid), hid.shape[1], _ptr(att), _ptr(rs), _ptr(os_),
                                             _ptr(nseg), _ptr(out_masks), cap, _ptr(offs), None), "seg_tail")
        if int(nseg.sum()) == 0:
            return None, nseg
        return [out_masks[int(offs[b]): int(offs[b]) + int(nseg[b]) * height[b] * width[b]].view(int(nseg[b]), height[b], width[b])
                for b in range(B)], nseg

    # ---- the reference surface -------------------------------------------------------------
    @torch.no_grad()
    def generate(self, clip_images, input_ids, sam_images, sam_resized_sizes, height, width, audios=None,
                 ref_images=None, output_hidden_states=True, return_dict_in_generate=True, max_new_tokens=128,
                 attention_masks=None, _return_extras: bool = False):
        """`AnyRefForCausalLM.generate` (model/anyref.py:647-822)."""
        ids, lens = self._rows(input_ids, attention_masks)
        B, Lmax = ids.shape
        if B > self.max_batch:
            raise ValueError(f"batch {B} > max_batch {self.max_batch} the handle was created for")
        clip = clip_images.to(self.device, torch.float32).contiguous()
        sam = sam_images.to(self.device, torch.float32).contiguous()
        extra, slots, n_extra = self._extra_overlapped(ids, lens, audios, self._ref_features(ref_images, B))
        height = [int(h) for h in height]
        width = [int(w) for w in width]
        rs = torch.tensor([[int(a), int(b)] for a, b in sam_resized_sizes], dtype=torch.int32).contiguous()
        os_ = torch.tensor([[h, w] for h, w in zip(height, width)], dtype=torch.int32).contiguous()
        Lout = Lmax + max_new_tokens
        out_ids = torch.zeros(B, Lout, dtype=torch.long)
        out_lens = torch.zeros(B, dtype=torch.int32)
        out_nseg = torch.zeros(B, dtype=torch.int32)
        offs = torch.zeros(B, dtype=torch.long)
        cap = sum(self.max_seg * h * w for h, w in zip(height, width))
        out_masks = torch.empty(cap, device=self.device, dtype=torch.float32)
        out_low = hid = None
        if _return_extras:       # low-res logits (what a DP driver all-gathers) and, unless "low", the hidden states
            L = 4 * self.cfg.sam.grid
            out_low = torch.zeros(B, self.max_seg, L, L, device=self.device, dtype=torch.float32)
            if _return_extras != "low":
                hid = torch.empty(B, self.cfg.llm.max_seq, self.cfg.llm.dim, device=self.device, dtype=torch.float32)
        eos = self.config.eos_token_id if self.config.eos_token_id is not None else -1
        self._check(self.lib.anyref_generate(
            self.h, self._stream(), _ptr(clip), _ptr(sam), _ptr(ids), _ptr(lens), B, Lmax, _ptr(extra),
            _ptr(slots), n_extra, _ptr(rs), _ptr(os_), int(max_new_tokens), int(eos), _ptr(out_ids), _ptr(out_lens),
            _ptr(out_nseg), _ptr(out_masks), cap, _ptr(offs), _ptr(out_low), _ptr(hid)), "generate")
        # HF pads finished rows with pad_token_id
        full = torch.full((B, int(out_lens.max())), int(self.config.pad_token_id or 0), dtype=torch.long)
        for b in range(B):
            full[b, : out_lens[b]] = out_ids[b, : out_lens[b]]
        output_ids = full.to(self.device)
        total = int(out_nseg.sum())
        if total == 0:                                    # anyref.py:729-730
            res = (output_ids, None, (None, None, None))
        elif self.cfg.rephrase_weight > 0 and total < B:  # (3-tuple in the reference too)
            # anyref.py:739-744,763-765: with rephrase on, the reference indexes the flattened [SEG] list by sample; fewer
            # [SEG] tokens than samples is its IndexError -> `no_mask` path: one all-zero [1, height[0], width[0]] mask per
            # sample (the same tensor object bs times), whatever the other rows produced
            z = torch.zeros((1, height[0], width[0]), device=self.device, dtype=torch.float32)
            res = (output_ids, [z] * B, (None, None, None))
        else:
            pred_masks = []
            for b in range(B):
                n, h, w = int(out_nseg[b]), height[b], width[b]
                o = int(offs[b])
                pred_masks.append(out_masks[o: o + n * h * w].view(n, h, w))
            res = (output_ids, pred_masks, (None, None, None))
        if self.success_arity == 2 and res[1] is not None and not (self.cfg.rephrase_weight > 0 and total < B):
            res = res[:2]                                 # the reference's own success path (anyref.py:822): any [SEG], no `no_mask`
        if _return_extras:
            return res, dict(out_lens=out_lens, nseg=out_nseg, low_res=out_low, hidden=hid)
        return res

    def forward(self, **kwargs):
        """`AnyRefForCausalLM.forward` (anyref.py:233-237): dispatch as the reference does."""
        if "past_key_values" in kwargs:
            raise NotImplementedError("plain LM forward with past_key_values is handled inside the HIP decode loop")
        return self.model_forward_new(**kwargs)

    __call__ = forward

    @torch.no_grad()
    def model_forward_new(self, clip_images, sam_images, input_ids, labels, attention_masks, sam_resized_sizes,
                          gt_masks, height, width, audios=None, ref_images=None, _return_extras: bool = False,
                          **kwargs):
        """Teacher-forced `model_forward_new` (anyref.py:239-466), inference arithmetic on the GPU,
        losses (`anyref.py:19-68,432-450`) evaluated with torch on the returned logits."""
        keeps: list = []
        ids, lens = self._rows(input_ids, attention_masks, keep_out=keeps)
        B, Lmax = ids.shape
        lab_rows, _ = self._rows(labels, None, keep_in=keeps)   # the same positions as the ids, whatever chose them
        clip = clip_images.to(self.device, torch.float32).contiguous()
        sam = sam_images.to(self.device, torch.float32).contiguous()
        extra, slots, n_extra = self._extra(ids, lens, audios, self._ref_features(ref_images, B))
        height = [int(h) for h in height]
        width = [int(w) for w in width]
        rs = torch.tensor([[int(a), int(b)] for a, b in sam_resized_sizes], dtype=torch.int32).contiguous()
        os_ = torch.tensor([[h, w] for h, w in zip(height, width)], dtype=torch.int32).contiguous()
        out_nseg = torch.zeros(B, dtype=torch.int32)
        offs = torch.zeros(B, dtype=torch.long)
        cap = sum(self.max_seg * h * w for h, w in zip(height, width))
        out_masks = torch.empty(cap, device=self.device, dtype=torch.float32)
        reph = torch.zeros(B, dtype=torch.int32)
        for b in range(B):
            pos = (lab_rows[b, : lens[b]] > 0).nonzero().flatten()
            reph[b] = int(pos[0]) if len(pos) else 0
        has_img = [(ids[b, : lens[b]] == IMAGE_TOKEN_INDEX).any().item() for b in range(B)]
        Sp = max(int(lens[b]) + (self.n_img - 1 if has_img[b] else 0) for b in range(B))
        logits = torch.empty(B, Sp, self.cfg.llm.vocab, device=self.device, dtype=torch.float32)
        hid = torch.empty(B, self.cfg.llm.max_seq, self.cfg.llm.dim, device=self.device,
                          dtype=torch.float32) if _return_extras else None
        self._check(self.lib.anyref_forward_teacher(
            self.h, self._stream(), _ptr(clip), _ptr(sam), _ptr(ids), _ptr(lens), B, Lmax, _ptr(extra), _ptr(slots),
            n_extra, _ptr(reph), _ptr(rs), _ptr(os_), _ptr(out_nseg), _ptr(out_masks), cap, _ptr(offs), None,
            _ptr(hid), _ptr(logits)), "forward_teacher")
        # LM loss (HF shift-by-one CE; the image span carries IGNORE labels)
        num, den = torch.zeros((), device=self.device), 0
        for b in range(B):
            lab = lab_rows[b, : lens[b]]
            if has_img[b]:
                ip = int((ids[b, : lens[b]] == IMAGE_TOKEN_INDEX).nonzero()[0])
                lab = torch.cat([lab[:ip], torch.full((self.n_img,), -100, dtype=torch.long), lab[ip + 1:]])
            lab = lab.to(self.device)
            valid = lab[1:] != -100
            if valid.any():
                lg = logits[b, : lab.shape[0] - 1][valid]
                num = num + F.cross_entropy(lg, lab[1:][valid], reduction="sum")
                den += int(valid.sum())
        lm_loss = num / max(den, 1)
        if int(out_nseg.sum()) == 0:                      # anyref.py:356-365
            return {"loss": lm_loss, "lm_loss": lm_loss}
        pred_masks = []
        for b in range(B):
            n, h, w = int(out_nseg[b]), height[b], width[b]
            o = int(offs[b])
            pred_masks.append(out_masks[o: o + n * h * w].view(n, h, w))
        out = {"lm_loss": lm_loss}
        if gt_masks is not None:
            ce = dice = torch.zeros((), device=self.device)
            nm = 0
            for b in range(B):
                pm, gt = pred_masks[b], gt_masks[b].to(pred_masks[b])
                if pm.shape[-2:] != gt.shape[-2:]:
                    pm = F.interpolate(pm[None], size=gt.shape[-2:], mode="bilinear", align_corners=False)[0]
                ce = ce + sigmoid_ce_loss(pm, gt, gt.shape[0]) * gt.shape[0]
                dice = dice + dice_loss(pm, gt, gt.shape[0]) * gt.shape[0]
                nm += gt.shape[0]
            ce = self.bce_loss_weight * ce / (nm + 1e-8)
            dice = self.dice_loss_weight * dice / (nm + 1e-8)
            out.update({"loss": lm_loss + ce + dice, "ce_loss": ce, "dice_loss": dice, "mask_loss": ce + dice})
        if _return_extras:
            out.update(pred_masks=pred_masks, hidden=hid, logits=logits)
        return out


def dice_loss(inputs, targets, num_masks, scale=1000, eps=1e-6):
    """`dice_loss` (model/anyref.py:19-47): the live body adds 1 to numerator and denominator and divides by
    `num_masks`; `scale` / `eps` are dead arguments there (the scaled variant is commented out, :38-42)."""
    inputs = inputs.sigmoid().flatten(1, 2)
    targets = targets.flatten(1, 2)
    numerator = 2 * (inputs * targets).sum(-1)
    denominator = inputs.sum(-1) + targets.sum(-1)
    loss = 1 - (numerator + 1) / (denominator + 1)
    return loss.sum() / num_masks


def sigmoid_ce_loss(inputs, targets, num_masks):
    """`sigmoid_ce_loss` (model/anyref.py:51-68)."""
    loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    return loss.flatten(1, 2).mean(1).sum() / (num_masks + 1e-8)

"""Shape/config description of the AnyRef inference path.

The reference spreads these numbers over three places: the (absent) LLaVA/HF
config objects (`model/anyref.py:171-179,188-215`), the SAM builders
(`model/segment_anything/build_sam.py:15-108`) and literal kwargs in the eval
scripts (`eval_referseg.py:62-69`).  Here they are one plain dataclass that is
handed to the C-ABI library as a flat struct (`include/anyref_hip.h`).
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Tuple, Union, List

# Placeholder ids in `input_ids`.  -200 is upstream LLaVA's IMAGE_TOKEN_INDEX; the
# audio / reference-image placeholder ids live in the absent `model/llava/constants.py`
# (`model/anyref.py:13`), so the two values below are this build's choice: callers
# only ever see them through `tokenizer_image_token(..., placehold=True)` and filter
# them out with `ids > 0` (`eval_referseg.py:150`).
IMAGE_TOKEN_INDEX = -200
AUDIO_REF_INDEX = -300
IMG_REF_INDEX = -400
IMG_REF_NUM = 4          # model/anyref.py:337-338 (16 -> 4 pooling)
AUDIO_REF_NUM = 3        # utils/avsbench.py:256-259 (3 clips)


@dataclass
class ClipConfig:
    image_size: int = 224
    patch: int = 14
    dim: int = 1024
    heads: int = 16
    layers: int = 24
    mlp: int = 4096
    eps: float = 1e-5
    select_layer: int = -2           # hidden_states[-2], CLS dropped ("patch")

    @property
    def grid(self) -> int:
        return self.image_size // self.patch

    @property
    def n_patches(self) -> int:
        return self.grid * self.grid

    @property
    def layers_run(self) -> int:
        # hidden_states has layers+1 entries; [-2] is the output of layer (layers-2)
        return self.layers + 1 + self.select_layer


@dataclass
class LlmConfig:
    vocab: int = 32007
    dim: int = 4096
    heads: int = 32
    layers: int = 32
    mlp: int = 11008
    rms_eps: float = 1e-6
    rope_theta: float = 10000.0
    max_seq: int = 1024              # KV-cache rows per sequence (>= 255 + L + max_new)

    @property
    def head_dim(self) -> int:
        return self.dim // self.heads


@dataclass
class SamConfig:
    img_size: int = 1024
    patch: int = 16
    dim: int = 1280
    depth: int = 32
    heads: int = 16
    mlp_ratio: int = 4
    window: int = 14
    global_idx: Tuple[int, ...] = (7, 15, 23, 31)
    out_chans: int = 256
    # mask decoder (build_sam.py:84-99)
    dec_heads: int = 8
    dec_mlp: int = 2048
    dec_depth: int = 2
    num_mask_tokens: int = 4

    @property
    def grid(self) -> int:
        return self.img_size // self.patch


@dataclass
class AudioTrunkConfig:
    """ImageBind audio trunk (model/ImageBind/models/imagebind_model.py:36-110,175-192 defaults of `imagebind_huge`)."""
    dim: int = 768
    blocks: int = 12
    heads: int = 12
    mel_bins: int = 128
    target_len: int = 204
    kernel: int = 16
    stride: int = 10
    clips: int = 3                   # clips per sample (utils/avsbench.py:256-259)

    @property
    def n_patches(self) -> int:
        return ((self.mel_bins - self.kernel) // self.stride + 1) * ((self.target_len - self.kernel) // self.stride + 1)


@dataclass
class AnyRefConfig:
    clip: ClipConfig = field(default_factory=ClipConfig)
    llm: LlmConfig = field(default_factory=LlmConfig)
    sam: SamConfig = field(default_factory=SamConfig)
    out_dim: int = 256
    seg_token_idx: Union[int, List[int]] = 32000
    rephrase_weight: float = 0.0
    eos_token_id: int = 2
    bos_token_id: int = 1
    pad_token_id: int = 0
    audio_dim: int = 1024            # ImageBind audio embedding width (imagebind_model.py:425-428)
    # None: the ImageBind trunk stays a PyTorch-ROCm module outside the handle (north_star's default);
    # an AudioTrunkConfig: the trunk runs in HIP inside the handle, given `model.audio_encoder.*`
    audio_trunk: "AudioTrunkConfig" = None

    def seg_range(self) -> Tuple[int, int]:
        """[lo, hi] inclusive id range that counts as a [SEG] token (anyref.py:197-200,723-726)."""
        if isinstance(self.seg_token_idx, (list, tuple)):
            return int(self.seg_token_idx[0]), int(self.seg_token_idx[-1])
        return int(self.seg_token_idx), int(self.seg_token_idx)

    def to_dict(self):
        return asdict(self)


def config_7b() -> AnyRefConfig:
    """C2: LLaVA-7B + CLIP ViT-L/14 + SAM-H (BASELINE.json configs[1])."""
    return AnyRefConfig()


def config_13b() -> AnyRefConfig:
    """C5: 13B LLM + ViT-L + SAM-H."""
    c = AnyRefConfig()
    c.llm = LlmConfig(dim=5120, heads=40, layers=40, mlp=13824)
    return c


def config_tiny(window: int = 14, sam_dim: int = 192, sam_heads: int = 3,
                llm_layers: int = 2) -> AnyRefConfig:
    """C1: tiny random-init plumbing config: ViT-Tiny + 2-layer 256-d LLM +
    stock mask decoder on a 224² image (14×14 SAM embedding)."""
    return AnyRefConfig(
        clip=ClipConfig(image_size=224, patch=14, dim=192, heads=3, layers=3, mlp=768),
        llm=LlmConfig(vocab=1000, dim=256, heads=4, layers=llm_layers, mlp=688, max_seq=512),
        sam=SamConfig(img_size=224, patch=16, dim=sam_dim, depth=2, heads=sam_heads,
                      window=window, global_idx=(1,)),
        seg_token_idx=999, eos_token_id=2,
    )

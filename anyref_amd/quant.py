"""fp8 (OCP e4m3fn) weight-only quantisation of the LLaMA linear layers: the torch statement of what
`ANYREF_MODE_PERF_FP8W` does at `finalize` (csrc/ops.hip `quant_fp8_rows_kernel`).

    scale[n] = max_k |W[n, k]| / 448        (1 for an all-zero row)
    q[n, k]  = round-to-nearest-even_e4m3fn(W[n, k] / scale[n])
    W'[n, k] = float(q[n, k]) * scale[n]    <- what the fp8 path multiplies by

`dequantized_state_dict` gives a checker (the oracle) the weights the device computes with.
"""
import re
from typing import Dict, Tuple

import torch

FP8_MAX = 448.0
_LLM_LINEAR = re.compile(r"^(model\.layers\.\d+\.(self_attn\.[qkvo]_proj|mlp\.(gate|up|down)_proj)\.weight|lm_head\.weight)$")


def quantize_rows_fp8(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """w [N, K] float -> (q uint8 [N, K] (e4m3fn bit patterns), scale f32 [N])."""
    w = w.detach().to(torch.float32)
    amax = w.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / FP8_MAX, torch.ones_like(amax))
    q = (w / scale[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale


def dequantize_rows_fp8(q_u8: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return q_u8.view(torch.float8_e4m3fn).to(torch.float32) * scale[:, None]


def is_fp8_weight(name: str) -> bool:
    """The tensors ANYREF_MODE_PERF_FP8W holds in fp8: q/k/v/o, gate/up/down of every layer, lm_head."""
    return _LLM_LINEAR.match(name) is not None


def dequantized_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        if is_fp8_weight(k):
            q, s = quantize_rows_fp8(v.cpu())
            out[k] = dequantize_rows_fp8(q, s).to(v.dtype)
        else:
            out[k] = v
    return out

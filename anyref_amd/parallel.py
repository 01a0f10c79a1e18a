"""Batch-sharded data parallel inference across the GPUs of one node.

The reference has no multi-GPU inference at all (single process, batch 1:
`eval_referseg.py:101-106,255`); images are independent (`anyref.py:797-819`), so the path shards
as pure DP: one process + one `anyref_handle` per GPU, weights replicated, inputs sharded on the
host, and two collectives per call — an all-gather of the low-resolution mask logits
(`[B_local, max_seg, 4g, 4g]` fp32 = 1 MiB per image at SAM-H) and one of a small int64 record per
image (token ids, [SEG] count, length).  Full-resolution masks are re-created from the gathered low-res logits
by the same bilinear postprocess (bit-identical: the resize is per mask), so the 4 MB/mask
full-res tensors never cross xGMI.  `backend="nccl"` is RCCL on ROCm; the CPU tests drive the
same code over gloo.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of a global batch of n owned by `rank` (rank r gets [lo, hi))."""
    per, rem = divmod(n, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def gather_results(low: torch.Tensor, nseg: torch.Tensor, ids: torch.Tensor, ids_len: torch.Tensor,
                   n_global: int, group=None):
    """All-gather one rank's results into global-batch order.

    low [b_local, max_seg, L, L] f32, nseg [b_local] i32, ids [b_local, Lout] i64, ids_len [b_local] i32
    (all on the communication device).  Ranks may own different b_local (ragged tail): rows are
    padded to ceil(n/world) for the collective and dropped afterwards.
    """
    world = dist.get_world_size(group)
    per = -(-n_global // world)
    # RCCL ("nccl" on ROCm) gathers device tensors in place; gloo (CPU tests / rehearsal) wants host tensors
    dev = low.device
    on_host = dist.get_backend(group) == "gloo" and dev.type != "cpu"

    def padrows(t):
        if t.shape[0] == per:
            return t.contiguous()
        pad = torch.zeros((per - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        return torch.cat([t, pad], 0).contiguous()

    # two collectives per call: the low-res logits, and one int64 record per image = [ids | nseg | ids_len]
    meta = torch.cat([ids.to(torch.int64), nseg.to(torch.int64)[:, None], ids_len.to(torch.int64)[:, None]], 1)
    gathered = []
    for t in (low, meta):
        t = padrows(t.cpu() if on_host else t)
        g = torch.empty((world * per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(g, t, group=group)
        gathered.append(g.to(dev) if on_host else g)
    gm = gathered[1]
    L = ids.shape[1]
    outs = [gathered[0], gm[:, L].to(nseg.dtype), gm[:, :L].contiguous(), gm[:, L + 1].to(ids_len.dtype)]
    keep = []
    for r in range(world):
        lo, hi = shard_range(n_global, r, world)
        keep += list(range(r * per, r * per + (hi - lo)))
    keep = torch.tensor(keep, device=dev)
    return tuple(o.index_select(0, keep) for o in outs)


class DataParallelAnyRef:
    """Wraps one per-GPU `AnyRefForCausalLM`; `generate` takes the GLOBAL batch on every rank, runs
    the local shard and returns the gathered global result (same return convention)."""

    def __init__(self, model, group=None):
        self.model = model
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    @torch.no_grad()
    def generate(self, clip_images, input_ids, sam_images, sam_resized_sizes, height, width, audios=None,
                 ref_images=None, max_new_tokens=128, attention_masks=None, **kw):
        n = clip_images.shape[0]
        lo, hi = shard_range(n, self.rank, self.world)
        sl = slice(lo, hi)
        m = self.model
        pick = lambda x: None if x is None else x[sl]
        Lout = input_ids.shape[1] + max_new_tokens
        if hi > lo:
            res, ex = m.generate(
                clip_images[sl], input_ids[sl], sam_images[sl], sam_resized_sizes[sl], height[sl], width[sl],
                audios=pick(audios), ref_images=pick(ref_images), max_new_tokens=max_new_tokens,
                attention_masks=pick(attention_masks), _return_extras="low")
            if self.world == 1:
                return res
            ids = res[0]                       # (ids, masks[, rest]): 2 or 3 values by `model.success_arity`
            low, nseg, lens = ex["low_res"], ex["nseg"].to(m.device), ex["out_lens"].to(m.device)
        else:
            # a global batch smaller than the world leaves this rank without images: it still takes part in both
            # collectives (with zero rows), or the other ranks would wait in all_gather forever
            L = 4 * m.cfg.sam.grid
            ids = torch.zeros(0, Lout, dtype=torch.long, device=m.device)
            low = torch.zeros(0, m.max_seg, L, L, dtype=torch.float32, device=m.device)
            nseg = torch.zeros(0, dtype=torch.int32, device=m.device)
            lens = torch.zeros(0, dtype=torch.int32, device=m.device)
        idp = torch.zeros(hi - lo, Lout, dtype=torch.long, device=m.device)
        idp[:, : ids.shape[1]] = ids
        low, nseg, gids, glen = gather_results(low, nseg, idp, lens, n, self.group)
        out_ids = gids[:, : int(glen.max())]
        total = int(nseg.sum())
        if total == 0:
            return out_ids, None, (None, None, None)
        if getattr(m.cfg, "rephrase_weight", 0) > 0 and total < n:
            # anyref.py:739-744,763-765 on the GLOBAL batch (what a single process would have seen): fewer [SEG]
            # tokens than samples with rephrasing on is the reference's `no_mask` return
            z = torch.zeros((1, int(height[0]), int(width[0])), device=m.device, dtype=torch.float32)
            return out_ids, [z] * n, (None, None, None)
        # full-resolution logits from the gathered low-res ones (per-mask bilinear, sam.py:137-172)
        pred = []
        for b in range(n):
            k = int(nseg[b])
            pred.append(m.postprocess(low[b, :k], sam_resized_sizes[b], (int(height[b]), int(width[b]))))
        if getattr(m, "success_arity", 3) == 2:
            return out_ids, pred               # the reference's own success path (anyref.py:822)
        return out_ids, pred, (None, None, None)

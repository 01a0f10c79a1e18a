#!/usr/bin/env python3
"""Headline benchmark: images/sec of the AnyRef refer-seg forward (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one `generate()` over one batch of synthetic (image, instruction) pairs per GPU:
CLIP ViT-L/14 -> LLaVA splice -> LLaMA-7B prefill (S=320) + greedy decode (10 new tokens, KV
cache) -> [SEG] hand-off -> SAM-H image encoder -> mask decoder -> 1024^2 mask logits, with the
inputs already resident in HBM.  N=1 is BASELINE.json configs[1] (batch 1).  For N>1 (launched by
torch.distributed.run, one rank per GPU) every rank runs the same per-GPU workload on its own
images (weak scaling) and the step ends with the RCCL all-gather of the low-res mask logits +
token ids.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dominant kernel of the step (by summed device time), timed live with hipEvent
                 pairs on the launch stream over the timed region
  cpu_baseline : the CPU oracle (oracle/anyref_oracle.py, fp32, KV cache on) timed on this host on
                 one image of the same workload (N=1 only)
  parity       : mask-logit max-abs-err and greedy-id identity of both arithmetic modes vs that
                 CPU forward on the same (image, instruction) pair
"""
import argparse
import json
import os
import sys
import time

# the host driver only supports dmabuf IPC: RCCL / cross-process device-memory sharing need this before HIP starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from anyref_amd.config import config_7b, config_tiny, IMAGE_TOKEN_INDEX  # noqa: E402
from anyref_amd.synth import synth_state_dict  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0       # HBM3E spec peak


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def make_inputs(cfg, B, seed, L_text=63):
    """SURVEY.md §8d synthetic inputs: BOS, one image placeholder, 63 random ids (L=65, S=320)."""
    g = torch.Generator().manual_seed(seed)
    clip = torch.randn(B, 3, cfg.clip.image_size, cfg.clip.image_size, generator=g)
    sam = torch.randn(B, 3, cfg.sam.img_size, cfg.sam.img_size, generator=g)
    hi = min(32000, cfg.llm.vocab - 8)
    ids = torch.stack([torch.cat([torch.tensor([1, IMAGE_TOKEN_INDEX]), torch.randint(3, hi, (L_text,), generator=g)])
                       for _ in range(B)])
    return clip, sam, ids


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=["c2", "c5", "tiny"],
                    help="c2: BASELINE configs[1] (the headline); c5: 13B LLM, fp8 weights, batch 8 (configs[4]); tiny: plumbing")
    ap.add_argument("--mode", default=None, choices=["perf", "perf_fp8w"], help="default: perf (bf16); c5: perf_fp8w")
    ap.add_argument("--batch-per-gpu", type=int, default=1)
    ap.add_argument("--max-new-tokens", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI) or gloo (rehearsal)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    if args.same_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    from anyref_amd.model import AnyRefForCausalLM
    from anyref_amd.parallel import gather_results

    B, T = args.batch_per_gpu, args.max_new_tokens
    mode = args.mode or ("perf_fp8w" if args.config == "c5" else "perf")
    if args.config == "c2":
        cfg = config_7b()
        cfg.llm.max_seq = 512
        S_img = 1024
    elif args.config == "c5":
        from anyref_amd.config import config_13b
        cfg = config_13b()
        cfg.llm.max_seq = 512
        S_img = 1024
        if B == 1:
            B = 8
        args.no_cpu_baseline = True      # the 13B fp32 oracle needs ~55 GB and minutes per image: not timed here
    else:
        cfg = config_tiny()
        S_img = cfg.sam.img_size
    t0 = time.time()
    sd = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.bfloat16)        # identical on every rank
    clip, sam, ids = make_inputs(cfg, B, seed=1 + rank)
    clip, sam = clip.to(dev), sam.to(dev)                                          # resident in HBM
    sizes, H, W = [(S_img, S_img)] * B, [S_img] * B, [S_img] * B
    model = AnyRefForCausalLM.from_state_dict(cfg, sd, mode=mode, device=local, max_batch=B, max_seg=2)
    model.config.eos_token_id = None                                               # fixed work: T new tokens
    torch.cuda.synchronize()
    log(f"[bench] weights + perf model ready in {time.time() - t0:.1f}s, {model.device_bytes / 2**30:.1f} GiB on device")

    # name the id the random model emits at decode step 3 as [SEG]
    out_ids, _, _ = model.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T)
    seg_id = int(out_ids[0, ids.shape[1] + 2])
    model.set_seg_token_idx(seg_id)

    n_global = B * world
    Lout = ids.shape[1] + T

    def step():
        (oids, masks, _), ex = model.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T, _return_extras=True)
        if world > 1:
            idp = torch.zeros(B, Lout, dtype=torch.long, device=dev)
            idp[:, : oids.shape[1]] = oids
            gather_results(ex["low_res"], ex["nseg"].to(dev), idp, ex["out_lens"].to(dev), n_global)
        return oids, masks

    for _ in range(args.warmup):
        step()
    # pre-pass: which kernel dominates the step?  Tags are per kernel INSTANTIATION (what rocprofv3 lists);
    # the dominant kernel is the kernel template whose instantiations sum to the most device time.
    def family(tag):
        return "_".join(tag.split("_")[:2])              # gemv_bf16_swiglu_x8 -> gemv_bf16

    model.profile_enable(True)
    step()
    table = model.profile_read()
    model.profile_enable(False)
    fam_ms = {}
    for k, v in table.items():
        fam_ms[family(k)] = fam_ms.get(family(k), 0.0) + v["ms"]
    dom = max(fam_ms, key=fam_ms.get) if fam_ms else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # hipEvent pairs around every 16th launch of the dominant kernel during the timed region (timing every
    # launch of every kernel costs ~4 us of dispatch gap each, 10 ms per step at 2300 launches)
    model.profile_enable(True, only_tag=dom, sample_every=16)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        oids, masks = step()
    barrier()
    dt = time.perf_counter() - t0
    timed = {k: v for k, v in model.profile_read().items() if v["count"]}
    model.profile_enable(False)
    # the same kernel without a co-running stream (one extra, untimed step): the timed region overlaps the
    # SAM encoder with the decode loop, which inflates every co-running kernel's duration
    iso = {}
    if dom:
        model.set_overlap(False)
        step()
        model.profile_enable(True, only_tag=dom, sample_every=1)
        step()
        iso = {k: v for k, v in model.profile_read().items() if v["count"]}
        model.profile_enable(False)
        model.set_overlap(True)
    tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    ips = n_global * args.steps / dt

    roofline = None
    if dom and timed:
        compute = dom.startswith(("gemm", "attn"))
        peak = (PEAK_BF16_TFLOPS if "bf16" in dom else PEAK_F32_TFLOPS) if compute else PEAK_HBM_GBS
        unit = "TFLOP/s" if compute else "GB/s"

        def rate(v):  # algorithmic FLOPs (or bytes) of the timed launches / their summed duration
            return (v["flops"] / 1e12 if compute else v["bytes"] / 1e9) / (v["ms"] / 1e3)

        def total(d):
            return {f: sum(v[f] for v in d.values()) for f in ("ms", "count", "flops", "bytes")}

        tot = total(timed)
        ach = rate(tot)
        roofline = dict(kernel=dom, bound="mfma" if compute else "hbm", achieved=round(ach, 2 if compute else 1), peak=peak,
                        unit=unit, frac=round(ach / peak, 4), traffic=None)
        per_step = sum(v["count"] for k, v in table.items() if family(k) == dom)
        roofline.update(event_pair_overhead_us=round(getattr(model, "profile_overhead_us", 0.0), 2),
                        launches_per_step=per_step, timed_launches=tot["count"],
                        avg_launch_us=round(tot["ms"] * 1e3 / tot["count"], 2),
                        algorithmic_bytes_per_launch=round(tot["bytes"] / tot["count"]),
                        share_of_step=round(tot["ms"] / tot["count"] * per_step * args.steps / 1e3 / dt, 3))
        # per instantiation: what to hold against rocprofv3 --kernel-trace --stats (profiles/)
        roofline["instantiations"] = {
            k: dict(launches_per_step=table[k]["count"], avg_launch_us=round(v["ms"] * 1e3 / v["count"], 2),
                    achieved=round(rate(v), 1), algorithmic_bytes_per_launch=round(v["bytes"] / v["count"]))
            for k, v in sorted(timed.items())}
        if iso:
            ti = total(iso)
            roofline["isolated"] = {"achieved": round(rate(ti), 1), "frac": round(rate(ti) / peak, 4),
                                    "avg_launch_us": round(ti["ms"] * 1e3 / ti["count"], 2),
                                    "instantiations": {k: dict(avg_launch_us=round(v["ms"] * 1e3 / v["count"], 2),
                                                               achieved=round(rate(v), 1)) for k, v in sorted(iso.items())},
                                    "note": "same kernels, SAM-encoder overlap off (no co-running stream)"}
        # HBM traffic from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command
        # (tools/pmc_summary.py -> profiles/pmc_traffic.json; counters cannot be read inside a timed run),
        # averaged over the kernel's launches of one step
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if args.config == "c2" and all(k in pmc for k in timed):
                n = sum(table[k]["count"] for k in timed)
                roofline["traffic"] = round(sum(pmc[k]["hbm_bytes_per_launch"] * table[k]["count"] for k in timed) / n)
        except (OSError, ValueError, KeyError):
            pass
    breakdown = {k: dict(ms_per_step=round(v["ms"], 3), launches=v["count"],
                         tflops=round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1),
                         gbs=round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)) for k, v in sorted(
        table.items(), key=lambda kv: -kv[1]["ms"])}

    res = {
        "metric": {"c2": "images/sec (1024^2, 7B LLM+ViT-L+SAM-H)", "c5": "images/sec (1024^2, 13B LLM+ViT-L+SAM-H, fp8 weights)",
                   "tiny": "images/sec (tiny plumbing config)"}[args.config],
        "value": round(ips, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if mode == "perf" else "bf16 (LLM linear weights fp8 e4m3, weight-only)",
        "data": "synthetic",
        "config": {"workload": {"c2": "C2: LLaVA-7B + CLIP ViT-L/14 + SAM-H refer-seg forward, 1024x1024, S=320 prompt, "
                                      f"{T} new tokens, KV cache",
                                "c5": "C5: 13B LLM + CLIP ViT-L/14 + SAM-H refer-seg forward, fp8-weight LLM, 1024x1024, S=320 "
                                      f"prompt, {T} new tokens, KV cache",
                                "tiny": "C1 tiny plumbing config"}[args.config],
                   "batch_per_gpu": B, "global_batch": n_global, "parallelism": f"dp{world}",
                   "max_new_tokens": T, "weights": "random-init N(0,0.02^2) rounded to bf16"},
        "roofline": roofline,
        "kernel_breakdown": breakdown,
    }

    # ---------------- CPU baseline + parity against it (rank 0, N=1 only) ----------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import anyref_oracle as O
        # the GPU box gives one GPU a share of 16 host cores (of 256 visible); oversubscribing them
        # makes the fp32 oracle many times slower
        cores = int(os.environ.get("ANYREF_CPU_THREADS", min(16, os.cpu_count() or 1)))
        torch.set_num_threads(cores)
        log("[bench] GPU leg: " + json.dumps({k: res[k] for k in ("value", "ms_per_step", "roofline")}))
        one = dict(clip=clip[:1], ids=ids[:1], sam=sam[:1], sizes=sizes[:1], H=H[:1], W=W[:1])

        def gen(m):
            o, mk, _ = m.generate(one["clip"], one["ids"], one["sam"], one["sizes"], one["H"], one["W"], max_new_tokens=T)
            torch.cuda.synchronize()
            return o[0].cpu().tolist(), mk

        # parity-mode model first: its greedy ids equal the oracle's, so it names the [SEG] id
        # (the id emitted at decode step 3) for every leg below
        parity = {}
        pm = None
        if not args.no_parity:
            pm = AnyRefForCausalLM.from_state_dict(cfg, sd, mode="parity", device=local, max_batch=1, max_seg=2)
            pm.config.eos_token_id = None
            p_ids, _ = gen(pm)
            seg_ref = p_ids[ids.shape[1] + 2]
            pm.set_seg_token_idx(seg_ref)
        else:
            seg_ref = seg_id
        cfg.seg_token_idx = seg_ref
        t0 = time.time()
        sd_cpu = {k: v.float().cpu() for k, v in sd.items()}
        log(f"[bench] weights on host in {time.time() - t0:.1f}s; timing the CPU oracle on 1 image, {cores} threads")
        t0 = time.time()
        with torch.no_grad():
            ref = O.anyref_generate(sd_cpu, cfg, clip[:1].cpu(), [ids[0]], sam[:1].cpu(), sizes[:1], H[:1], W[:1],
                                    max_new_tokens=T, eos=False, use_cache=True)
        cpu_s = time.time() - t0
        res["cpu_baseline"] = {"value": round(1.0 / cpu_s, 5), "unit": "images/sec", "cores": cores, "kind": "port",
                               "sample": f"1 image of the same workload (full forward, fp32, KV cache on): {cpu_s:.1f}s"}
        want = ref["output_ids"][0].tolist()
        ref_mask = ref["pred_masks"][0] if ref["pred_masks"] is not None else None
        if pm is not None:
            t1 = time.perf_counter()
            p_ids, p_masks = gen(pm)
            parity["parity"] = {"greedy_ids_identical": p_ids == want, "ms_per_image": round((time.perf_counter() - t1) * 1e3, 2)}
            if p_ids == want and p_masks is not None and ref_mask is not None:
                parity["parity"]["mask_logit_max_abs_err"] = float((p_masks[0].cpu() - ref_mask).abs().max())
                parity["parity"]["logit_range"] = float(ref_mask.abs().max())
            del pm
            torch.cuda.empty_cache()
        model.set_seg_token_idx(seg_ref)
        got, g_masks = gen(model)
        parity["perf"] = {"greedy_ids_identical": got == want}
        if got == want and g_masks is not None and ref_mask is not None:
            parity["perf"]["mask_logit_max_abs_err"] = float((g_masks[0].cpu() - ref_mask).abs().max())
        else:
            # bf16 flipped a near-tied argmax: say where and how tied, then isolate the NUMERICAL error by
            # teacher-forcing the oracle's ids through the bf16 path (anyref.py:239-430 semantics)
            L0 = ids.shape[1]
            k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), len(want) - 1)
            parity["perf"]["first_divergence_new_token"] = k - L0
            with torch.no_grad():
                hrow = ref["hidden"][0][k - 1 + cfg.clip.n_patches - 1]
                lg = torch.nn.functional.linear(hrow, sd_cpu["lm_head.weight"])
                top2 = torch.topk(lg, 2).values
            parity["perf"]["oracle_top2_logit_gap_there"] = float(top2[0] - top2[1])
            parity["perf"]["oracle_logit_std"] = float(lg.std())
            full = ref["output_ids"][0]
            if ref_mask is not None:
                out = model.model_forward_new(clip[:1], sam[:1], full[None], full[None].clone(), None, sizes[:1], None,
                                              H[:1], W[:1], _return_extras=True)
                if "pred_masks" in out:
                    parity["perf"]["teacher_forced_mask_logit_max_abs_err"] = float(
                        (out["pred_masks"][0].cpu() - ref_mask).abs().max())
        if ref_mask is not None:
            parity["logit_range"] = float(ref_mask.abs().max())
        res["parity"] = parity
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: images/sec of the AnyRef refer-seg forward (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one `generate()` over one batch of synthetic (image, instruction) pairs per GPU:
CLIP ViT-L/14 -> LLaVA splice -> LLaMA-7B prefill (S=320) + greedy decode (10 new tokens, KV
cache) -> [SEG] hand-off -> SAM-H image encoder -> mask decoder -> 1024^2 mask logits, with the
inputs already resident in HBM.  N=1 is BASELINE.json configs[1] (C2: batch 1).  For N>1 (launched by
torch.distributed.run, one rank per GPU) the default is configs[2] (C3: the same model, 4 images per GPU
= global batch 4 N; 32 on 8 GPUs): every rank runs the same per-GPU workload on its own images (weak
scaling) and the step ends with the RCCL all-gather of the low-res mask logits + token ids.

The timed region runs what ships: hipGraph replay of the decode step, SAM encoder on the second stream,
no profiler.  Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dominant kernel of the step (by summed device time), measured in a separate untimed
                 pass of the same step from KERNEL-SIDE timestamps (every workgroup stamps the 100 MHz wall
                 clock at entry / exit; include/anyref_hip.h anyref_stamps_*): in situ (frac) and with the
                 co-running stream off (isolated, what rocprofv3 --kernel-trace --stats can check)
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
# SAM-H encoder 5.961e12 + CLIP 1.553e11 + projector 2.1e9 + LLM prefill 4.20e12 + decode 1.34e11 +
# hand-off 3.6e7 + mask decoder 3.6e9 (2 x MAC, S = 320 prompt, 10 new tokens)
FLOP_PER_IMAGE_C2 = 1.045e13


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
    ap.add_argument("--config", default=None, choices=["c2", "c3", "c4", "c5", "tiny"],
                    help="c2: BASELINE configs[1] (the headline; default at --gpus 1); c3: configs[2] = c2 at 4 images per GPU "
                         "(default at --gpus N > 1: global batch 4 N); c4: c2 + audio reference (raw mel clips through the "
                         "ImageBind trunk, configs[3]); c5: 13B LLM, fp8 weights, batch 8 (configs[4]); tiny: plumbing")
    ap.add_argument("--audio-trunk", default="hip", choices=["hip", "torch"],
                    help="c4: ImageBind audio trunk inside the HIP handle (f-4) or as the PyTorch-ROCm module")
    ap.add_argument("--mode", default=None, choices=["perf", "perf_fp8w", "parity16"],
                    help="default: perf (bf16 LLaMA / CLIP, f16 SAM encoder); c5: perf_fp8w; parity16: the tolerance-meeting mode "
                         "(f32 activations as bf16 pairs x exactly stored bf16 weights) as the timed headline")
    ap.add_argument("--batch-per-gpu", type=int, default=None, help="default: 1 (c2 / c4), 4 (c3), 8 (c5)")
    ap.add_argument("--roofline-steps", type=int, default=3, help="steps of the untimed kernel-timestamp passes")
    ap.add_argument("--stamps-out", default=None,
                    help="write the per-launch kernel-timestamp table (CSV) here; default gpurun_out/stamps_<config>.csv "
                         "when gpurun_out/ exists")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N = 1, c2 only: skip the short c3-shape / c4 / c5 sub-measurements appended to the line")
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
    if args.config is None:
        args.config = "c2" if world == 1 else "c3"
    # one process per GPU under torch.distributed.run: nothing may have touched HIP before the rank picks its device
    assert not torch.cuda.is_initialized(), "HIP was initialised before the rank selected its device"
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

    T = args.max_new_tokens

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def family(tag):
        return "_".join(tag.split("_")[:2])              # gemv_bf16_swiglu_x8 -> gemv_bf16

    def measure(config, steps, warmup, B=None, mode=None, roofline_steps=3, stamps_out=None, full=True):
        """Build `config`, time `steps` steps of it on the production path, then (untimed) measure its dominant
        kernel.  -> (result dict, context for the CPU / parity legs)."""
        mode = mode or args.mode or ("perf_fp8w" if config == "c5" else "perf")
        B = B or {"c3": 4, "c5": 8}.get(config, 1)
        if config in ("c2", "c3", "c4"):
            cfg = config_7b()
            cfg.llm.max_seq = 512
            S_img = 1024
            if config == "c4" and args.audio_trunk == "hip":
                from anyref_amd.config import AudioTrunkConfig
                cfg.audio_trunk = AudioTrunkConfig()
        elif config == "c5":
            from anyref_amd.config import config_13b
            cfg = config_13b()
            cfg.llm.max_seq = 512
            S_img = 1024
        else:
            cfg = config_tiny()
            S_img = cfg.sam.img_size
        t0 = time.time()
        sd = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.bfloat16)        # identical on every rank
        clip, sam, ids = make_inputs(cfg, B, seed=1 + rank)
        clip, sam = clip.to(dev), sam.to(dev)                                          # resident in HBM
        sizes, H, W = [(S_img, S_img)] * B, [S_img] * B, [S_img] * B
        gen_kw = {}
        if config == "c4":
            # AVSBench-style prompt: 3 <audio_ref> placeholders (utils/avsbench.py:256-259), raw mel clips [1,3,1,128,204]
            from anyref_amd.config import AUDIO_REF_INDEX
            ids = torch.cat([ids[:, :5], torch.full((B, 3), AUDIO_REF_INDEX), ids[:, 5:-3]], 1)
            gen_kw["audios"] = [torch.randn(1, 3, 1, 128, 204, generator=torch.Generator().manual_seed(9 + b)).to(dev)
                                for b in range(B)]
        model = AnyRefForCausalLM.from_state_dict(cfg, sd, mode=mode, device=local, max_batch=B, max_seg=2)
        model.config.eos_token_id = None                                               # fixed work: T new tokens
        if config == "c4" and args.audio_trunk == "torch":
            from anyref_amd.audio import ImageBindAudio
            torch.manual_seed(0)
            model.audio_encoder = ImageBindAudio().eval().to(dev)
        torch.cuda.synchronize()
        log(f"[bench:{config}] weights + {mode} model ready in {time.time() - t0:.1f}s, {model.device_bytes / 2**30:.1f} GiB on device")

        # name the id the random model emits at decode step 3 as [SEG]
        out_ids, _, _ = model.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T, **gen_kw)
        seg_id = int(out_ids[0, ids.shape[1] + 2])
        model.set_seg_token_idx(seg_id)
        n_global = B * world
        Lout = ids.shape[1] + T

        def step():
            if world == 1:
                oids, masks, _ = model.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T, **gen_kw)
                return oids, masks
            (oids, masks, _), ex = model.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T, _return_extras="low", **gen_kw)
            idp = torch.zeros(B, Lout, dtype=torch.long, device=dev)
            idp[:, : oids.shape[1]] = oids
            gather_results(ex["low_res"], ex["nseg"].to(dev), idp, ex["out_lens"].to(dev), n_global)
            return oids, masks

        for _ in range(warmup):
            step()
        # ---- the timed region: what ships (hipGraph decode step, two streams, no profiler, no stamps) ----
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ips = n_global * steps / dt
        # ---- N > 1: who took part, and what the per-step exchange costs (untimed, after the timed region) ----
        ranks_info = collective = None
        if world > 1:
            # every rank reports the GPU it ran on: the first SCALE record shows N ranks on N DISTINCT devices
            pr = torch.cuda.get_device_properties(dev)
            me = {"rank": rank, "local_rank": local, "device_index": dev.index, "device_name": pr.name,
                  "device_uuid": str(getattr(pr, "uuid", "")), "pci_bus": f"{getattr(pr, 'pci_domain_id', 0):04x}:"
                  f"{getattr(pr, 'pci_bus_id', 0):02x}:{getattr(pr, 'pci_device_id', 0):02x}", "pid": os.getpid()}
            ranks_info = [None] * world
            dist.all_gather_object(ranks_info, me)
            # the step's exchange alone: the two all-gathers of gather_results on this rank's real result shapes
            (o_, _, _), ex_ = model.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T, _return_extras="low", **gen_kw)
            idp_ = torch.zeros(B, Lout, dtype=torch.long, device=dev)
            idp_[:, : o_.shape[1]] = o_
            low_, nseg_, len_ = ex_["low_res"], ex_["nseg"].to(dev), ex_["out_lens"].to(dev)
            for _ in range(2):
                gather_results(low_, nseg_, idp_, len_, n_global)
            barrier()
            t0 = time.perf_counter()
            n_coll = 10
            for _ in range(n_coll):
                gather_results(low_, nseg_, idp_, len_, n_global)
            barrier()
            ct = torch.tensor([(time.perf_counter() - t0) / n_coll], dtype=torch.float64,
                              device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(ct, op=dist.ReduceOp.MAX)
            bytes_rank = low_.numel() * 4 + B * (Lout + 2) * 8
            collective = {"backend": "rccl (torch.distributed nccl)" if args.dist_backend == "nccl" else args.dist_backend,
                          "what": "all_gather_into_tensor of the low-res mask logits [B, max_seg, 256, 256] f32 + one int64 record "
                                  "per image (ids, [SEG] count, length); full-res masks are re-created per rank",
                          "bytes_per_rank": int(bytes_rank), "bytes_gathered_per_rank": int(bytes_rank * world),
                          "ms": round(float(ct.item()) * 1e3, 4), "share_of_step": round(float(ct.item()) / (dt / steps), 5),
                          "timed": f"{n_coll} exchanges after 2 warm-ups, max over ranks, outside the timed region"}

        # ---- untimed: per-kernel table of one step (hipEvent brackets, eager launches) -> the dominant kernel.
        # Tags are per kernel INSTANTIATION (what rocprofv3 lists); the dominant kernel is the kernel template whose
        # instantiations sum to the most device time.
        model.profile_enable(True)
        step()
        table = model.profile_read()
        model.profile_enable(False)
        fam_ms = {}
        for k, v in table.items():
            fam_ms[family(k)] = fam_ms.get(family(k), 0.0) + v["ms"]
        dom = max(fam_ms, key=fam_ms.get) if fam_ms else None

        # ---- untimed: the dominant kernel's duration from KERNEL-SIDE timestamps on the production launch path
        # (graph replay, encoder co-running): frac; then the same with the co-running stream off: isolated
        def stamp_pass(n):
            model.stamps_enable(True)
            step()                                   # first stamped step captures the stamped graph: not recorded
            model.stamps_read()
            for _ in range(n):
                step()
            rows = model.stamps_read()
            model.stamps_enable(False)
            # a pass longer than the stamp record (48 graph replays, 256 launches x 512 workgroups per step) would silently
            # drop launches while the per-step figures still divide by `n`
            assert model.stamps_dropped == 0, f"{model.stamps_dropped} launches went unstamped: lower --roofline-steps / --max-new-tokens"
            return rows

        def by_tag(rows):
            out = {}
            for r in rows:
                d = out.setdefault(r["tag"], dict(us=0.0, count=0, bytes=0.0))
                d["us"] += r["t1_us"] - r["t0_us"]
                d["count"] += 1
                d["bytes"] += r["bytes"]
            return out

        roofline = None
        # (every rank runs the same passes whatever it finds dominant: the steps contain collectives)
        rows, iso_rows = [], []
        roofline_steps = min(roofline_steps, max(1, 48 // max(1, T - 1)))     # graph replays per pass <= the stamp record's epochs
        if roofline_steps > 0:
            rows = stamp_pass(roofline_steps)
            model.set_overlap(False)
            iso_rows = stamp_pass(1)
            model.set_overlap(True)
        if rows and iso_rows and dom and dom.startswith("gemv"):
            situ = by_tag(rows)
            tot_us, tot_n, tot_b = (sum(v[f] for v in situ.values()) for f in ("us", "count", "bytes"))
            ach = tot_b / 1e9 / (tot_us / 1e6)
            per_step = sum(v["count"] for k, v in table.items() if family(k) == dom)
            roofline = dict(kernel=dom, bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                            frac=round(ach / PEAK_HBM_GBS, 4), traffic=None, traffic_source=None,
                            source="kernel-side timestamps (100 MHz wall clock stamped by every workgroup; launch = "
                                   "max(end) - min(start)) over an untimed pass of the production path: hipGraph replay, "
                                   "SAM encoder co-running on the second stream",
                            launches_per_step=per_step, timed_launches=tot_n, avg_launch_us=round(tot_us / tot_n, 2),
                            algorithmic_bytes_per_launch=round(tot_b / tot_n),
                            share_of_step=round(tot_us / roofline_steps / 1e3 / (dt / steps * 1e3), 3))
            roofline["instantiations"] = {
                k: dict(launches_per_step=round(v["count"] / roofline_steps), avg_launch_us=round(v["us"] / v["count"], 2),
                        achieved=round(v["bytes"] / 1e9 / (v["us"] / 1e6), 1),
                        algorithmic_bytes_per_launch=round(v["bytes"] / v["count"])) for k, v in sorted(situ.items())}
            # gaps between consecutive stamped launches of one captured step (what a kernel boundary + the attention
            # launch between two GEMVs cost): reported, not part of `achieved`
            gaps = sorted(b["t0_us"] - a["t1_us"] for a, b in zip(rows, rows[1:]) if a["epoch"] == b["epoch"] and a["epoch"] >= 0)
            if gaps:
                roofline["gap_between_launches_us"] = dict(median=round(gaps[len(gaps) // 2], 2),
                                                           p10=round(gaps[len(gaps) // 10], 2),
                                                           p90=round(gaps[len(gaps) * 9 // 10], 2))
            iso = by_tag(iso_rows)
            iu, inn, ib = (sum(v[f] for v in iso.values()) for f in ("us", "count", "bytes"))
            roofline["isolated"] = {
                "achieved": round(ib / 1e9 / (iu / 1e6), 1), "frac": round(ib / 1e9 / (iu / 1e6) / PEAK_HBM_GBS, 4),
                "avg_launch_us": round(iu / inn, 2),
                "instantiations": {k: dict(avg_launch_us=round(v["us"] / v["count"], 2),
                                           achieved=round(v["bytes"] / 1e9 / (v["us"] / 1e6), 1)) for k, v in sorted(iso.items())},
                "note": "same kernels, same launch path, SAM-encoder overlap off (no co-running stream): the figure "
                        "rocprofv3 --kernel-trace --stats can check (profiles/r03_*kernel_stats.csv); rocprofv3's duration "
                        "includes the dispatch ramp in front of the first wave, the stamps do not"}
            # HBM traffic from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command
            # (tools/pmc_summary.py -> profiles/pmc_traffic.json; counters cannot be read inside a timed run),
            # averaged over the kernel's launches of one step
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                if config == "c2" and B == 1 and mode == "perf" and all(k in pmc for k in situ):
                    n = sum(table[k]["count"] for k in situ if k in table)
                    roofline["traffic"] = round(sum(pmc[k]["hbm_bytes_per_launch"] * table[k]["count"] for k in situ if k in table) / n)
                    roofline["traffic_source"] = ("committed file profiles/pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / "
                                                  "WRITE_SIZE passes of this command, FETCH_SIZE doubled per the guide's gfx950 "
                                                  "correction); not collected by this run")
            except (OSError, ValueError, KeyError):
                pass
            out_path = stamps_out
            if out_path is None and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
                out_path = os.path.join(ROOT, "gpurun_out", f"stamps_{config}.csv")
            if out_path and rank == 0:
                with open(out_path, "w") as f:
                    f.write("pass,step_replay,tag,start_us,end_us,duration_us,algorithmic_bytes,GBps\n")
                    for name, rr in (("in_situ", rows), ("isolated", iso_rows)):
                        for r in rr:
                            d = r["t1_us"] - r["t0_us"]
                            f.write(f"{name},{r['epoch']},{r['tag']},{r['t0_us']:.2f},{r['t1_us']:.2f},{d:.2f},{r['bytes']:.0f},"
                                    f"{r['bytes'] / 1e3 / max(d, 1e-9):.1f}\n")
                roofline["per_launch_table"] = os.path.relpath(out_path, ROOT)
        elif dom:
            # a compute-bound dominant kernel (large batches): hipEvent brackets of an eager pass, overlap off
            compute = dom.startswith(("gemm", "attn"))
            peak = (PEAK_BF16_TFLOPS if any(t in dom for t in ("bf16", "f16", "sp16")) else PEAK_F32_TFLOPS) if compute else PEAK_HBM_GBS
            sel = {k: v for k, v in table.items() if family(k) == dom and v["count"]}

            def part(rows, as_compute, what):
                ms_ = sum(v["ms"] for v in rows.values())
                work_ = sum(v["flops"] / 1e12 if as_compute else v["bytes"] / 1e9 for v in rows.values())
                ach_ = work_ / (ms_ / 1e3)
                pk = peak if as_compute else PEAK_HBM_GBS
                return dict(kernel=dom, launches=what, bound="mfma" if as_compute else "hbm", achieved=round(ach_, 1), peak=pk,
                            unit="TFLOP/s" if as_compute else "GB/s", frac=round(ach_ / pk, 4), ms_per_step=round(ms_, 3),
                            traffic=None, traffic_source=None, source="hipEvent brackets over one eager, untimed step",
                            launches_per_step=sum(v["count"] for v in rows.values()))
            # a GEMM template serves two regimes in one step: the decode launches (<= 16 rows, tags "..._dec": every weight byte
            # streamed once per step -> HBM roofline) and the prefill launches (MFMA roofline) -- never one lumped figure
            dec = {k: v for k, v in sel.items() if k.endswith("_dec")} if compute else {}
            rest = {k: v for k, v in sel.items() if k not in dec}
            parts = ([part(rest, compute, "prefill / encoder launches (M > 16 rows)" if dec else "all")] if rest else []) + \
                    ([part(dec, False, "decode launches (M <= 16 rows: weight streaming)")] if dec else [])
            roofline = dict(max(parts, key=lambda r: r["ms_per_step"]))
            if len(parts) > 1:
                roofline["parts"] = parts
            # the same launches with the co-running SAM stream off (event brackets of two busy streams include the time a
            # launch waits for CUs the other stream holds): what rocprofv3 --kernel-trace of an overlap-off run can check
            model.set_overlap(False)
            model.profile_enable(True)
            step()
            iso_table = model.profile_read()
            model.profile_enable(False)
            model.set_overlap(True)
            iso_sel = {k: v for k, v in iso_table.items() if family(k) == dom and v["count"]}
            iso_dec = {k: v for k, v in iso_sel.items() if k.endswith("_dec")} if compute else {}
            iso_rest = {k: v for k, v in iso_sel.items() if k not in iso_dec}
            iso_parts = ([part(iso_rest, compute, "prefill / encoder launches (M > 16 rows)" if iso_dec else "all")] if iso_rest else []) + \
                        ([part(iso_dec, False, "decode launches (M <= 16 rows: weight streaming)")] if iso_dec else [])
            if iso_parts:
                iso_main = max(iso_parts, key=lambda r: r["ms_per_step"])
                roofline["isolated"] = {k: iso_main[k] for k in ("launches", "bound", "achieved", "peak", "unit", "frac", "ms_per_step")}
                roofline["isolated"]["note"] = "same eager step, SAM-encoder overlap off (no co-running stream)"
        breakdown = {k: dict(ms_per_step=round(v["ms"], 3), launches=v["count"],
                             tflops=round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1),
                             gbs=round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)) for k, v in sorted(
            table.items(), key=lambda kv: -kv[1]["ms"])}
        workload = {"c2": "C2: LLaVA-7B + CLIP ViT-L/14 + SAM-H refer-seg forward, 1024x1024, S=320 prompt, "
                          f"{T} new tokens, KV cache",
                    "c3": f"C3: C2 at {B} images per GPU, batch-sharded data parallel: global batch {B}*N = {n_global} "
                          f"({B * 8} on 8 GPUs), all-gather of mask logits + ids per step; S=320 prompts, {T} new tokens, KV cache",
                    "c4": "C4: C2 + audio reference: 3 raw mel clips [1,3,1,128,204] -> ImageBind audio trunk "
                          f"({'HIP, inside the handle' if args.audio_trunk == 'hip' else 'PyTorch-ROCm module'}) -> "
                          f"audio_projector -> 3 <audio_ref> slots; S=320 prompt, {T} new tokens, KV cache",
                    "c5": "C5: 13B LLM + CLIP ViT-L/14 + SAM-H refer-seg forward, fp8-weight LLM, 1024x1024, S=320 "
                          f"prompt, {T} new tokens, KV cache",
                    "tiny": "C1 tiny plumbing config"}[config]
        res = {
            "metric": {"c2": "images/sec (1024^2, 7B LLM+ViT-L+SAM-H)", "c3": "images/sec (1024^2, 7B LLM+ViT-L+SAM-H)",
                       "c4": "images/sec (1024^2, 7B LLM+ViT-L+SAM-H, audio-referred: ImageBind audio trunk + 3 audio tokens)",
                       "c5": "images/sec (1024^2, 13B LLM+ViT-L+SAM-H, fp8 weights)",
                       "tiny": "images/sec (tiny plumbing config)"}[config],
            "value": round(ips, 3), "unit": "images/sec", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"perf": "bf16 (LLaMA / CLIP) + f16 (SAM encoder)",
                      "perf_fp8w": "bf16 (LLaMA / CLIP; LLM linear weights fp8 e4m3, weight-only) + f16 (SAM encoder)",
                      "parity16": "f32 activations as bf16 pairs x bf16 weights (bf16 MFMA, f32 accumulate); f32 attention"}[mode],
            "data": "synthetic",
            "config": {"workload": workload, "batch_per_gpu": B, "global_batch": n_global, "parallelism": f"dp{world}",
                       "max_new_tokens": T, "weights": "random-init N(0,0.02^2) rounded to bf16",
                       "launch_path": "hipGraph decode step, SAM encoder on a second stream, profiler off"},
            "roofline": roofline,
        }
        if ranks_info is not None:
            res["ranks"] = ranks_info
            res["distinct_devices"] = len({(r["device_uuid"], r["pci_bus"]) for r in ranks_info})
            res["collective"] = collective
        if full:
            res["kernel_breakdown"] = breakdown
        if config == "c2" and T == 10 and mode == "perf" and ids.shape[1] == 65:
            # algorithmic FLOPs of one image (S = 320, T = 10, bf16) x images/s over the dense bf16 MFMA peak
            res["mfma_frac_e2e"] = round(FLOP_PER_IMAGE_C2 * ips / world / (PEAK_BF16_TFLOPS * 1e12), 4)
            res["flop_per_image"] = FLOP_PER_IMAGE_C2
        if full and config == "c2" and world == 1 and B == 1 and mode == "perf":
            # untimed: the MFMA-bound stages alone (no co-running stream), against the dense bf16 / f16 MFMA peak;
            # algorithmic FLOPs from SURVEY.md §8d (S = 320 prompt)
            def stage_ms(f, n=5):
                for _ in range(2):
                    f()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    f()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e3
            emb = torch.randn(1, 320, cfg.llm.dim, device=dev) * 0.02
            st = {"sam_encoder": (stage_ms(lambda: model.sam_encode(sam)), 5.961e12),
                  "llm_prefill_S320": (stage_ms(lambda: model.llm_forward(emb)), 4.20e12),
                  "clip_tower_projector": (stage_ms(lambda: model.encode_images(clip)), 1.574e11)}
            res["mfma_stages"] = {k: dict(ms=round(ms, 3), tflops=round(fl / ms / 1e9, 1),
                                          frac_of_mfma_peak=round(fl / ms / 1e9 / PEAK_BF16_TFLOPS, 4)) for k, (ms, fl) in st.items()}
            res["mfma_stages"]["note"] = ("each stage alone on the chip through its C-ABI entry (sam_encode / llm_forward / encode_images, "
                                          "5 calls after 2 warm-ups, host-timed); SAM encoder operands f16, LLaMA / CLIP bf16")
        ctx = dict(cfg=cfg, sd=sd, model=model, clip=clip, sam=sam, ids=ids, sizes=sizes, H=H, W=W, B=B, mode=mode)
        return res, ctx

    res, ctx = measure(args.config, args.steps, args.warmup, B=args.batch_per_gpu, roofline_steps=args.roofline_steps,
                       stamps_out=args.stamps_out)
    if args.config in ("c4", "c5", "c3"):
        args.no_cpu_baseline = True      # the CPU baseline / parity legs are quoted on C2 (13B fp32 needs ~55 GB of host memory)
    cfg, sd, model, clip, sam, ids, sizes, H, W = (ctx[k] for k in ("cfg", "sd", "model", "clip", "sam", "ids", "sizes", "H", "W"))
    mode = ctx["mode"]

    # ---- driver-visible secondary lines (N = 1, headline config only): the other single-GPU configurations, short ----
    if world == 1 and args.config == "c2" and not args.no_secondary and os.environ.get("ANYREF_BENCH_SECONDARY", "1") != "0":
        log("[bench] GPU leg: " + json.dumps({k: res[k] for k in ("value", "ms_per_step", "roofline")}))
        secondary = {}
        import gc
        for name, conf in (("c3_shape", "c3"), ("c4", "c4"), ("c5", "c5")):
            try:
                r2, c2 = measure(conf, 5, 2, mode="perf_fp8w" if conf == "c5" else "perf", roofline_steps=1, stamps_out="", full=False)
                del c2
                secondary[name] = {k: r2[k] for k in ("metric", "value", "ms_per_step", "dtype", "config", "roofline")}
            except Exception as e:          # a secondary line must never cost the headline
                secondary[name] = {"error": repr(e)}
            gc.collect()
            torch.cuda.empty_cache()
        secondary["note"] = ("5 timed steps each on this GPU, same production launch path; c3_shape = configs[2]'s per-GPU shape "
                             "(4 images per call) on one GPU; parity of these shapes: tests/test_gpu_e2e.py, test_gpu_audio.py, "
                             "test_gpu_fp8w.py")
        res["secondary"] = secondary

    # ---- the tolerance-meeting mode, timed in the same run on the same workload and launch path (N = 1, C2) ----
    # north_star's bar (mask logits within 1e-3, identical greedy ids) is met by `parity16`: weights in their exact bf16
    # storage, every activation that feeds a matrix product carried as a pair of bf16 terms (2^-18 relative), f32
    # attention operands; its parity numbers (fan-in workload, where the bar bites) are attached below from `parity`
    tol = None
    if world == 1 and args.config == "c2" and mode != "parity16" and os.environ.get("ANYREF_BENCH_TOLERANCE", "1") != "0":
        try:
            m16 = AnyRefForCausalLM.from_state_dict(cfg, sd, mode="parity16", device=local, max_batch=ctx["B"], max_seg=2)
            m16.config.eos_token_id = None
            o16, _, _ = m16.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T)
            m16.set_seg_token_idx(int(o16[0, ids.shape[1] + 2]))
            for _ in range(args.warmup):
                m16.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                m16.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T)
            barrier()
            dt16 = (time.perf_counter() - t0) / args.steps
            tol = {"mode": "parity16", "value": round(ctx["B"] / dt16, 3), "unit": "images/sec", "ms_per_step": round(dt16 * 1e3, 3),
                   "steps": args.steps, "warmup": args.warmup, "device_gib": round(m16.device_bytes / 2 ** 30, 2),
                   "arithmetic": "bf16 weights exactly as stored (never widened in HBM); activations f32 carried as two bf16 terms "
                                 "(hi + lo, 2^-18 relative), one bf16 MFMA pass per term; decode GEMV: f32 activation row x bf16 "
                                 "weights; attention operands, KV cache, norms, residual streams, mask decoder f32",
                   "mask_logit_max_abs_err": None, "ids_match_rate": None}
            del m16
            torch.cuda.empty_cache()
        except Exception as e:              # must never cost the headline
            tol = {"mode": "parity16", "error": repr(e)}
        res["tolerance_mode"] = tol
        log("[bench] tolerance mode: " + json.dumps(tol))

    # ---------------- CPU baseline + parity against it (rank 0, N=1 only) ----------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import anyref_oracle as O
        from oracle.check import compare_generate, summarize
        # the GPU box gives one GPU a share of 16 host cores (of 256 visible); oversubscribing them
        # makes the fp32 oracle many times slower
        cores = int(os.environ.get("ANYREF_CPU_THREADS", min(16, os.cpu_count() or 1)))
        torch.set_num_threads(cores)
        one = (clip[:1], ids[0], sam[:1], sizes[:1], H[:1], W[:1])

        def mode_model(sd_, cfg_, mode_):
            m_ = AnyRefForCausalLM.from_state_dict(cfg_, sd_, mode=mode_, device=local, max_batch=1, max_seg=2)
            m_.config.eos_token_id = None
            return m_

        # the oracle's own greedy ids name the [SEG] id (the id emitted at decode step 3)
        t0 = time.time()
        sd_cpu = {k: v.float().cpu() for k, v in sd.items()}
        log(f"[bench] weights on host in {time.time() - t0:.1f}s; timing the CPU oracle on 1 image, {cores} threads")
        with torch.no_grad():
            img_feats = O.encode_images(sd_cpu, cfg, clip[:1].cpu())
            first = O.greedy_generate(sd_cpu, cfg, O.splice_embeddings(sd_cpu, cfg, ids[0], img_feats[0]), T, None)[0]
        cfg.seg_token_idx = int(first[2])          # (that run doubles as the warm-up of the timed ones below)

        def oracle_image(use_cache=True, max_new=T):
            with torch.no_grad():
                return O.anyref_generate(sd_cpu, cfg, clip[:1].cpu(), [ids[0]], sam[:1].cpu(), sizes[:1], H[:1], W[:1],
                                         max_new_tokens=max_new, eos=False, use_cache=use_cache)

        reps = int(os.environ.get("ANYREF_CPU_REPS", 3))
        times, ref = [], None
        for _ in range(reps):                      # warm-up, then the median of the timed runs
            t0 = time.time()
            ref = oracle_image()
            times.append(time.time() - t0)
        cpu_s = sorted(times)[len(times) // 2]
        res["cpu_baseline"] = {"value": round(1.0 / cpu_s, 5), "unit": "images/sec", "cores": cores, "kind": "port",
                               "sample": f"1 image of the same workload (full forward, fp32, KV cache on): median of {reps} "
                                         f"runs after a warm-up, {cpu_s:.1f}s ({', '.join(f'{t:.1f}' for t in times)})"}
        # the reference as written re-runs the whole prefix for every new token (use_cache=False, anyref.py:171):
        # timed on a bounded sample -- 2 new tokens (2 full-prefix LLM passes + CLIP + SAM + decoder) -- and
        # extended to T tokens with the measured cost of one full-prefix pass
        if not os.environ.get("ANYREF_NO_NOCACHE"):
            with torch.no_grad():
                emb0 = O.splice_embeddings(sd_cpu, cfg, ids[0], img_feats[0])
                t0 = time.time()
                O.llama_layers(sd_cpu, cfg, emb0)
                pass_s = time.time() - t0
            t0 = time.time()
            oracle_image(use_cache=False, max_new=2)
            two = time.time() - t0
            nocache_s = two + (T - 2) * pass_s
            res["cpu_baseline"]["no_cache"] = {
                "value": round(1.0 / nocache_s, 5), "unit": "images/sec",
                "sample": f"reference semantics (use_cache=False): 2 new tokens timed ({two:.1f}s) + {T - 2} x one full-prefix "
                          f"LLM pass ({pass_s:.1f}s each, timed once) = {nocache_s:.1f}s per image"}
        lm = sd_cpu["lm_head.weight"]
        n_img = cfg.clip.n_patches
        parity = {"workload_normal": {}}
        model.set_seg_token_idx(cfg.seg_token_idx)
        parity["workload_normal"]["perf"] = summarize([compare_generate(model, ref, *one, T, lm, n_img)])
        if not args.no_parity:
            pm = mode_model(sd, cfg, "parity")
            t1 = time.perf_counter()
            parity["workload_normal"]["parity"] = summarize([compare_generate(pm, ref, *one, T, lm, n_img)])
            parity["workload_normal"]["parity"]["ms_per_image"] = round((time.perf_counter() - t1) * 1e3, 2)
            del pm
            torch.cuda.empty_cache()
            pm = mode_model(sd, cfg, "parity16")
            t1 = time.perf_counter()
            parity["workload_normal"]["parity16"] = summarize([compare_generate(pm, ref, *one, T, lm, n_img)])
            parity["workload_normal"]["parity16"]["ms_per_image"] = round((time.perf_counter() - t1) * 1e3, 2)
            del pm
            torch.cuda.empty_cache()
        parity["workload_normal"]["note"] = ("the timed workload (SURVEY.md §8d, every matrix N(0,0.02^2)): logits are nearly "
                                             "flat, so read the errors relative to logit_range")
        # ---- the PARITY workload: fan-in-scaled weights (O(1) activations, peaked LM logits, mask logits of several
        # units; anyref_amd.synth init="fan_in"), several prompts over one image -> ids-match rate + relative error
        n_prompts = int(os.environ.get("ANYREF_PARITY_PROMPTS", 4))
        if not args.no_parity and args.config == "c2" and n_prompts > 0:
            del sd_cpu, ref             # 27 GB of host fp32 weights: make room for the second workload's
            t0 = time.time()
            sd2 = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.bfloat16, init="fan_in")
            sd2_cpu = {k: v.float().cpu() for k, v in sd2.items()}
            g2 = torch.Generator().manual_seed(7)
            prompts = [torch.cat([torch.tensor([1, IMAGE_TOKEN_INDEX]), torch.randint(3, 32000, (63,), generator=g2)])
                       for _ in range(n_prompts)]
            clip1, sam1 = clip[:1].cpu(), sam[:1].cpu()
            with torch.no_grad():
                feats = O.encode_images(sd2_cpu, cfg, clip1)
                outs = [O.greedy_generate(sd2_cpu, cfg, O.splice_embeddings(sd2_cpu, cfg, p_, feats[0]), T, None)
                        for p_ in prompts]
                cfg.seg_token_idx = int(outs[0][0][2])
                fulls = [torch.cat([p_, torch.tensor(o[0])]) for p_, o in zip(prompts, outs)]
                tail = O.generate_tail(sd2_cpu, cfg, [fulls[0]], [len(prompts[0])], [outs[0][1]], None, sam1, sizes[:1],
                                       H[:1], W[:1])
            refs = [dict(output_ids=[f], hidden=[o[1]], pred_masks=tail["pred_masks"] if i == 0 else None)
                    for i, (f, o) in enumerate(zip(fulls, outs))]
            log(f"[bench] parity workload: CPU oracle on {n_prompts} prompts (1 with masks) in {time.time() - t0:.0f}s")
            lm2 = sd2_cpu["lm_head.weight"]
            parity["workload_fan_in"] = {}
            for mode_ in ("parity", "parity16", "perf"):
                m2 = mode_model(sd2, cfg, mode_)
                rows = [compare_generate(m2, refs[i], clip[:1], prompts[i], sam[:1], sizes[:1], H[:1], W[:1], T, lm2, n_img)
                        for i in range(n_prompts)]
                parity["workload_fan_in"][mode_] = summarize(rows)
                del m2
                torch.cuda.empty_cache()
            parity["workload_fan_in"]["note"] = ("fan-in-scaled weights: hidden states O(1), LM logit std ~4, mask logits of "
                                                 "several units; a flipped greedy id is followed by a teacher-forced comparison")
        # the headline parity numbers (north_star: identical greedy ids, mask logits within 1e-3) per arithmetic mode,
        # taken from the workload where they bite
        src = parity.get("workload_fan_in", parity["workload_normal"])
        for mode_ in ("parity", "parity16", "perf"):
            if mode_ in src:
                parity[mode_] = {k: src[mode_][k] for k in ("ids_match_rate", "mask_logit_max_abs_err", "mask_logit_rel_err",
                                                             "logit_range", "first_divergence_new_token") if k in src[mode_]}
                parity[mode_]["rel_err"] = src[mode_].get("mask_logit_rel_err")
        res["parity"] = parity
        if tol and "error" not in tol and "parity16" in parity:
            tol["mask_logit_max_abs_err"] = parity["parity16"].get("mask_logit_max_abs_err")
            tol["ids_match_rate"] = parity["parity16"].get("ids_match_rate")
            tol["logit_range"] = parity["parity16"].get("logit_range")
            tol["parity_workload"] = "fan_in" if "workload_fan_in" in parity else "normal"
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

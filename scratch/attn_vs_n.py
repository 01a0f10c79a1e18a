"""Lab: decode step span (kernel-side stamps) vs context length: how much of the decode attention is fixed cost?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b, IMAGE_TOKEN_INDEX
from anyref_amd.model import AnyRefForCausalLM
from anyref_amd.synth import synth_state_dict
from bench import make_inputs
cfg = config_7b(); cfg.llm.max_seq = 512
sd = synth_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16)
clip, sam, ids = make_inputs(cfg, 1, seed=1); clip, sam = clip.cuda(), sam.cuda()
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode="perf", max_batch=1, max_seg=2); m.config.eos_token_id = None
sizes, H, W = [(1024, 1024)], [1024], [1024]
m.set_overlap(False)
def run(p):
    m.stamps_enable(True); m.generate(clip, p, sam, sizes, H, W, max_new_tokens=10); m.stamps_read()
    m.generate(clip, p, sam, sizes, H, W, max_new_tokens=10); rows = m.stamps_read(); m.stamps_enable(False)
    ep = {}
    for r in rows:
        if r["epoch"] >= 0: ep.setdefault(r["epoch"], []).append(r)
    spans = sorted(max(x["t1_us"] for x in v) - min(x["t0_us"] for x in v) for v in ep.values())
    busy = sorted(sum(x["t1_us"] - x["t0_us"] for x in v) for v in ep.values())
    g = sorted(b["t0_us"] - a["t1_us"] for v in ep.values() for a, b in zip(v, v[1:]))
    return spans[len(spans) // 2], busy[len(busy) // 2], g[len(g) // 2], g[len(g) * 9 // 10]
for name, p in (("S=320 (image + 64 ids)", ids), ("S=259 (image + 3 ids)", ids[:, :4]), ("S=3 (no image)", torch.tensor([[1, 5, 6]])),
                ("S=64 (no image)", torch.cat([torch.tensor([[1]]), ids[:, 2:]], 1))):
    s, b, g50, g90 = run(p)
    print(f"{name:26s}: step span {s:7.1f} us, GEMV busy {b:7.1f}, non-GEMV {s - b:6.1f} us; gap median {g50:.2f} p90 {g90:.2f}", flush=True)

"""Lab: per-launch workgroup spread of the decode GEMVs (kernel-side stamps), overlap off."""
import os, sys, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b
from anyref_amd.model import AnyRefForCausalLM
from anyref_amd.synth import synth_state_dict
from bench import make_inputs
cfg = config_7b(); cfg.llm.max_seq = 512
sd = synth_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16)
clip, sam, ids = make_inputs(cfg, 1, seed=1); clip, sam = clip.cuda(), sam.cuda()
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode="perf", max_batch=1, max_seg=2); m.config.eos_token_id = None
sizes, H, W = [(1024, 1024)], [1024], [1024]
m.set_overlap(os.environ.get("OVL", "0") == "1")
m.stamps_enable(True); m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10); m.stamps_read()
m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10); rows = m.stamps_read(); m.stamps_enable(False)
rows = [r for r in rows if 2 <= r["epoch"] <= 5]
by = collections.defaultdict(list)
for r in rows: by[(r["tag"], round(r["bytes"] / 1e6))].append(r)
med = lambda v: sorted(v)[len(v) // 2]
for (tag, mb), v in sorted(by.items()):
    L = med([r['t1_us'] - r['t0_us'] for r in v])
    print(f"{tag:22s} {mb:4d} MB n={len(v):4d}: launch {L:6.2f} us, start spread "
          f"{med([r['start_spread_us'] for r in v]):5.2f}, end spread (tail) {med([r['end_spread_us'] for r in v]):5.2f}, median WG span "
          f"{med([r['wg_median_us'] for r in v]):6.2f}  -> {mb / L * 1e3:.0f} GB/s")
gaps = [b["t0_us"] - a["t1_us"] for a, b in zip(rows, rows[1:]) if a["epoch"] == b["epoch"]]
print("gap median", med(gaps), "p10", sorted(gaps)[len(gaps) // 10], "p90", sorted(gaps)[len(gaps) * 9 // 10])

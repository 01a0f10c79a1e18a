"""Lab: C2 image time and per-step decode spans vs the side stream's workgroup cap (set_side_share).
python scratch/side_share.py 0 96 128 160 192"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b
from anyref_amd.model import AnyRefForCausalLM
from anyref_amd.synth import synth_state_dict
from bench import make_inputs
caps = [tuple(int(v) for v in x.split(":")) for x in sys.argv[1:]] or [(0, 6), (128, 6)]
cfg = config_7b(); cfg.llm.max_seq = 512
sd = synth_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16)
clip, sam, ids = make_inputs(cfg, 1, seed=1); clip, sam = clip.cuda(), sam.cuda()
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode=os.environ.get("ANYREF_LAB_MODE", "perf"), max_batch=1, max_seg=2); m.config.eos_token_id = None
sizes, H, W = [(1024, 1024)], [1024], [1024]
o, _, _ = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10)
m.set_seg_token_idx(int(o[0, ids.shape[1] + 2]))
ref = None
def timed(n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, out
for rnd in range(2):
    for cap in caps:
        m.set_side_share(cap[0], cap[1])
        timed(3)
        ms, out = timed(10)
        if ref is None: ref = (out[0].clone(), out[1][0].clone())
        same = torch.equal(out[0], ref[0]) and torch.equal(out[1][0], ref[1])
        m.stamps_enable(True); m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10); m.stamps_read()
        m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10); rows = m.stamps_read(); m.stamps_enable(False)
        ep = {}
        for r in rows:
            if r["epoch"] >= 0: ep.setdefault(r["epoch"], []).append(r)
        spans = [max(x["t1_us"] for x in v) - min(x["t0_us"] for x in v) for _, v in sorted(ep.items())]
        busy = [sum(x["t1_us"] - x["t0_us"] for x in v) for _, v in sorted(ep.items())]
        print(f"round {rnd} cap {cap[0]:3d} steps {cap[1]}: {ms:6.2f} ms/image, outputs identical {same}, decode step spans (ms): " +
              " ".join(f"{s / 1e3:.2f}" for s in spans) + " | GEMV busy: " + " ".join(f"{s / 1e3:.2f}" for s in busy), flush=True)

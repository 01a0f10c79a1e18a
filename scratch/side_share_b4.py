"""Lab: C3's per-GPU shape (4 images per call) vs the side stream's CU share (ANYREF_SIDE_ANYB=1 lets batches > 1 take it)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b
from anyref_amd.model import AnyRefForCausalLM
from anyref_amd.synth import synth_state_dict
from bench import make_inputs
B = 4
cfg = config_7b(); cfg.llm.max_seq = 512
sd = synth_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16)
clip, sam, ids = make_inputs(cfg, B, seed=1); clip, sam = clip.cuda(), sam.cuda()
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode="perf", max_batch=B, max_seg=2); m.config.eos_token_id = None
sizes, H, W = [(1024, 1024)] * B, [1024] * B, [1024] * B
o, _, _ = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10)
m.set_seg_token_idx(int(o[0, ids.shape[1] + 2]))
def timed(n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rnd in range(2):
    for cap, steps in [(0, 6)] + [tuple(int(v) for v in x.split(":")) for x in sys.argv[1:]]:
        m.set_side_share(cap, steps); timed(2)
        print(f"round {rnd} cap {cap:3d} steps {steps}: {timed():7.2f} ms per call of {B} images = {B / timed() * 1e3:.1f} images/s", flush=True)

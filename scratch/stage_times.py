"""Per-stage times (alone, no overlap) through the Python mirror: clip / prefill / sam / mask; and generate with overlap on/off."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b
from anyref_amd.synth import synth_state_dict
from anyref_amd.model import AnyRefForCausalLM
from bench import make_inputs
cfg = config_7b(); cfg.llm.max_seq = 512
dev = torch.device('cuda', 0)
sd = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.bfloat16)
clip, sam, ids = make_inputs(cfg, 1, seed=1); clip, sam = clip.to(dev), sam.to(dev)
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode='perf', max_batch=1, max_seg=2)
emb = torch.randn(1, 320, 4096, device=dev) * 0.02
e = m.sam_encode(sam)[0]; pp = torch.randn(1, 256, device=dev)
fs = {'mask': lambda: m.mask_decode(e, pp, (1024, 1024), (1024, 1024)), 'sam': lambda: m.sam_encode(sam),
      'prefill': lambda: m.llm_forward(emb), 'clip': lambda: m.encode_images(clip)}
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
out = {k: round(t(f), 3) for k, f in fs.items()}
m.config.eos_token_id = -1
g = lambda: m.generate(clip, ids, sam, [(1024, 1024)], [1024], [1024], max_new_tokens=10)
out['generate'] = round(t(g), 3)
m.set_overlap(False); out['generate_no_overlap'] = round(t(g), 3)
print(out)

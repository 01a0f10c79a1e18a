"""Per-stage times at batch B (alone, no overlap): python scratch/stage_times_b.py 4"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b
from anyref_amd.synth import synth_state_dict
from anyref_amd.model import AnyRefForCausalLM
from bench import make_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = config_7b(); cfg.llm.max_seq = 512
dev = torch.device('cuda', 0)
sd = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.bfloat16)
clip, sam, ids = make_inputs(cfg, B, seed=1); clip, sam = clip.to(dev), sam.to(dev)
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode='perf', max_batch=B, max_seg=2)
emb = torch.randn(B, 320, 4096, device=dev) * 0.02
fs = {'sam': lambda: m.sam_encode(sam), 'prefill': lambda: m.llm_forward(emb), 'clip': lambda: m.encode_images(clip)}
def t(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
out = {k: round(t(f), 3) for k, f in fs.items()}
m.config.eos_token_id = -1
sz = [(1024, 1024)] * B
g = lambda T: (lambda: m.generate(clip, ids, sam, sz, [1024] * B, [1024] * B, max_new_tokens=T))
m.set_overlap(False)
t10, t2 = t(g(10)), t(g(2))
out['generate_no_sam_10tok'] = round(t10, 3); out['decode_step'] = round((t10 - t2) / 8, 3)
m.set_overlap(True)
o, _, _ = m.generate(clip, ids, sam, sz, [1024] * B, [1024] * B, max_new_tokens=10)
m.set_seg_token_idx(int(o[0, ids.shape[1] + 2]))
out['generate_with_masks'] = round(t(g(10)), 3)
print(B, out)

"""generate() at C2 with the decode step as a hipGraph vs eager launches (same process, alternated)."""
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
m.config.eos_token_id = None
o, _, _ = m.generate(clip, ids, sam, [(1024, 1024)], [1024], [1024], max_new_tokens=10)
m.set_seg_token_idx(int(o[0, ids.shape[1] + 2]))
g = lambda: m.generate(clip, ids, sam, [(1024, 1024)], [1024], [1024], max_new_tokens=10)
def t(n=10):
    for _ in range(2): g()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for r in range(3):
    for on in (True, False):
        m.set_graphs(on)
        print("graphs", on, round(t(), 3), flush=True)

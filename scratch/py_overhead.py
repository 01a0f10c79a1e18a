"""How much of a C2 generate() is Python-side work around the C-ABI call?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b
from anyref_amd.model import AnyRefForCausalLM
from anyref_amd.synth import synth_state_dict
from bench import make_inputs
cfg = config_7b(); cfg.llm.max_seq = 512
sd = synth_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16)
clip, sam, ids = make_inputs(cfg, 1, seed=1)
clip, sam = clip.cuda(), sam.cuda()
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode="perf", max_batch=1, max_seg=2)
m.config.eos_token_id = None
o, _, _ = m.generate(clip, ids, sam, [(1024, 1024)], [1024], [1024], max_new_tokens=10)
m.set_seg_token_idx(int(o[0, ids.shape[1] + 2]))
real = m.lib.anyref_generate
acc = [0.0]
class Wrap:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = real(*a); acc[0] += time.perf_counter() - t0; return r
m.lib.anyref_generate = Wrap()
for _ in range(3):
    m.generate(clip, ids, sam, [(1024, 1024)], [1024], [1024], max_new_tokens=10)
torch.cuda.synchronize()
acc[0] = 0.0
N = 10
t0 = time.perf_counter()
for _ in range(N):
    m.generate(clip, ids, sam, [(1024, 1024)], [1024], [1024], max_new_tokens=10)
    torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"per image: total {tot / N * 1e3:.3f} ms, inside anyref_generate {acc[0] / N * 1e3:.3f} ms, python around it {(tot - acc[0]) / N * 1e3:.3f} ms")

import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    m.generate(clip, ids, sam, [(1024, 1024)], [1024], [1024], max_new_tokens=10)
torch.cuda.synchronize()
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(14)
print(st.getvalue()[:3500])

import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b
from anyref_amd.synth import synth_state_dict
from anyref_amd.model import AnyRefForCausalLM
from bench import make_inputs
dev = torch.device('cuda', 0)
def t(f, n=8):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
cfg = config_7b(); cfg.llm.max_seq = 512
sd = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.bfloat16)
clip, sam, ids = make_inputs(cfg, 1, seed=1); clip, sam = clip.to(dev), sam.to(dev)
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode='perf', max_batch=1, max_seg=2); m.config.eos_token_id = None
sizes, H, W = [(1024, 1024)], [1024], [1024]
out_ids, _, _ = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10)
m.set_seg_token_idx(int(out_ids[0, ids.shape[1] + 2]))
print('prio', os.environ.get('ANYREF_SAM_PRIO'), 'T=10 %.2f ms  T=20 %.2f ms' % (
    t(lambda: m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10)), t(lambda: m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=20))), flush=True)

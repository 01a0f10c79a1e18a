import sys, time, torch
sys.path.insert(0, '.')
from anyref_amd.config import config_7b
from anyref_amd.synth import synth_state_dict
from anyref_amd.model import AnyRefForCausalLM
from bench import make_inputs
cfg = config_7b(); cfg.llm.max_seq = 512
dev = torch.device('cuda', 0)
sd = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.bfloat16)
clip, sam, ids = make_inputs(cfg, 1, seed=1); clip, sam = clip.to(dev), sam.to(dev)
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode='perf', max_batch=1, max_seg=2); m.config.eos_token_id = None
sizes, H, W = [(1024, 1024)], [1024], [1024]
out_ids, _, _ = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10)
m.set_seg_token_idx(int(out_ids[0, ids.shape[1] + 2]))
def t(f, n=8):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
ref = None
for ov in (1, 0):
    m.set_overlap(bool(ov))
    for g in (1, 0, 1, 0):
        m.set_graphs(bool(g))
        for T in (10, 40):
            o = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T)
            if T == 10:
                if ref is None: ref = o
                assert torch.equal(o[0], ref[0]) and torch.equal(o[1][0], ref[1][0]), 'graph/eager mismatch'
            print('overlap=%d graphs=%d T=%2d  %.2f ms' % (ov, g, T, t(lambda: m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T))), flush=True)

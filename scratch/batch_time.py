import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b
from anyref_amd.synth import synth_state_dict
from anyref_amd.model import AnyRefForCausalLM
from bench import make_inputs
dev = torch.device('cuda', 0)
def t(f, n=3):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
cfg = config_7b(); cfg.llm.max_seq = 512
sd = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.bfloat16)
for mode in ('perf', 'perf_fp8w'):
    for B in (1, 2, 4, 8):
        clip, sam, ids = make_inputs(cfg, B, seed=1); clip, sam = clip.to(dev), sam.to(dev)
        sizes, H, W = [(1024, 1024)] * B, [1024] * B, [1024] * B
        m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode=mode, max_batch=B, max_seg=2); m.config.eos_token_id = None
        out_ids, _, _ = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10)
        m.set_seg_token_idx(int(out_ids[0, ids.shape[1] + 2]))
        a = t(lambda: m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10)); b = t(lambda: m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=20))
        print(f'7B {mode:10s} B={B}  T=10 {a:7.1f} ms ({B*1e3/a:5.1f} img/s)  decode step {(b-a)/10:6.2f} ms', flush=True)
        del m; torch.cuda.empty_cache()

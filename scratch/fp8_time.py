import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b, config_13b
from anyref_amd.synth import synth_state_dict
from anyref_amd.model import AnyRefForCausalLM
from bench import make_inputs
dev = torch.device('cuda', 0)
def t(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
which = sys.argv[1]
if which == '7b':
    cfg = config_7b(); cfg.llm.max_seq = 512
    sd = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.bfloat16)
    clip, sam, ids = make_inputs(cfg, 1, seed=1); clip, sam = clip.to(dev), sam.to(dev)
    sizes, H, W = [(1024, 1024)], [1024], [1024]
    for mode in ('perf', 'perf_fp8w'):
        m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode=mode, max_batch=1, max_seg=2); m.config.eos_token_id = None
        out_ids, _, _ = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10)
        m.set_seg_token_idx(int(out_ids[0, ids.shape[1] + 2]))
        a = t(lambda: m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10)); b = t(lambda: m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=20))
        print(f'7B {mode:10s} {m.device_bytes/2**30:5.1f} GiB  T=10 {a:.2f} ms  T=20 {b:.2f} ms  -> {(b-a)/10:.3f} ms/step', flush=True)
        del m; torch.cuda.empty_cache()
else:
    B = 8
    cfg = config_13b(); cfg.llm.max_seq = 512
    sd = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.bfloat16)
    clip, sam, ids = make_inputs(cfg, B, seed=1); clip, sam = clip.to(dev), sam.to(dev)
    sizes, H, W = [(1024, 1024)] * B, [1024] * B, [1024] * B
    for mode in ('perf_fp8w', 'perf'):
        m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode=mode, max_batch=B, max_seg=2); m.config.eos_token_id = None
        out_ids, _, _ = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10)
        m.set_seg_token_idx(int(out_ids[0, ids.shape[1] + 2]))
        a = t(lambda: m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10), n=3)
        print(f'13B batch {B} {mode:10s} {m.device_bytes/2**30:5.1f} GiB  T=10 {a:.1f} ms per batch = {B*1e3/a:.2f} images/s', flush=True)
        del m; torch.cuda.empty_cache()

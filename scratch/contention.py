import os, sys, time, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b
from anyref_amd.synth import synth_state_dict
from anyref_amd.model import AnyRefForCausalLM
from bench import make_inputs
dev = torch.device('cuda', 0)
cfg = config_7b(); cfg.llm.max_seq = 512
sd = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.bfloat16)
clip, sam, ids = make_inputs(cfg, 1, seed=1); clip, sam = clip.to(dev), sam.to(dev)
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode='perf', max_batch=1, max_seg=2); m.config.eos_token_id = None
sizes, H, W = [(1024, 1024)], [1024], [1024]
m.set_overlap(False)
gen = lambda T: m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T)
gen(10)
side = torch.cuda.Stream()
A = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16); Bm = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
small = torch.randn(512, 512, device=dev, dtype=torch.bfloat16)
big1 = torch.empty(64 * 2**20, device=dev, dtype=torch.float32); big2 = torch.empty_like(big1)
l2a = torch.empty(2 * 2**20, device=dev, dtype=torch.float32); l2b = torch.empty_like(l2a)
def run_side(kind, stop):
    with torch.cuda.stream(side):
        while not stop.is_set():
            for _ in range(20):
                if kind == 'mm_big': torch.mm(A, Bm)
                elif kind == 'mm_small': torch.mm(small, small)
                elif kind == 'copy_hbm': big2.copy_(big1)
                elif kind == 'copy_l2': l2b.copy_(l2a)
            side.synchronize()
def step_ms():
    for _ in range(2): gen(10)
    torch.cuda.synchronize(); t0 = time.perf_counter(); gen(10); torch.cuda.synchronize(); a = time.perf_counter() - t0
    t0 = time.perf_counter(); gen(40); torch.cuda.synchronize(); b = time.perf_counter() - t0
    return (b - a) / 30 * 1e3
print('no side work           : decode step %.2f ms' % step_ms(), flush=True)
for kind in ('mm_big', 'mm_small', 'copy_hbm', 'copy_l2'):
    stop = threading.Event(); th = threading.Thread(target=run_side, args=(kind, stop)); th.start()
    time.sleep(0.2)
    s = step_ms()
    stop.set(); th.join(); torch.cuda.synchronize()
    print('side stream %-10s : decode step %.2f ms' % (kind, s), flush=True)

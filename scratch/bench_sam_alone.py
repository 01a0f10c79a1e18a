"""Lab: the SAM-H encoder alone (f16 perf build), host-timed: median of 5 x 10 calls."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b
from anyref_amd.synth import synth_state_dict
from anyref_amd.model import AnyRefForCausalLM
from bench import make_inputs
cfg = config_7b(); cfg.llm.max_seq = 512
dev = torch.device('cuda', 0)
sd = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.bfloat16)
clip, sam, ids = make_inputs(cfg, 1, seed=1); sam = sam.to(dev)
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode='perf', max_batch=1, max_seg=2)
for _ in range(3): e = m.sam_encode(sam)
ts = []
for r in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): e = m.sam_encode(sam)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10 * 1e3)
print(f"{os.environ.get('TAG','')} sam encoder alone: {sorted(ts)[2]:.3f} ms  checksum {e[0].float().abs().mean().item():.6f}", flush=True)

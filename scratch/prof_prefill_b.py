"""LLaMA prefill alone at B sequences of 320 rows (run under rocprofv3 --kernel-trace --stats): python scratch/prof_prefill_b.py 4"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b
from anyref_amd.synth import synth_state_dict
from anyref_amd.model import AnyRefForCausalLM
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = config_7b(); cfg.llm.max_seq = 512
dev = torch.device('cuda', 0)
sd = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.bfloat16)
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode='perf', max_batch=B, max_seg=2)
del sd
emb = torch.randn(B, 320, 4096, device=dev) * 0.02
for _ in range(2): m.llm_forward(emb)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): m.llm_forward(emb)
torch.cuda.synchronize()
print(f"prefill B={B}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")

"""Lab: SAM-H encoder on B images in one call vs B calls of one image (cache residency of the activations):
python scratch/sam_batch_split.py 4"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b
from anyref_amd.synth import synth_state_dict
from anyref_amd.model import AnyRefForCausalLM
from bench import make_inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mode = os.environ.get("ANYREF_LAB_MODE", "perf")
cfg = config_7b(); cfg.llm.max_seq = 512
dev = torch.device('cuda', 0)
sd = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.bfloat16)
clip, sam, ids = make_inputs(cfg, B, seed=1); sam = sam.to(dev)
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode=mode, max_batch=B, max_seg=2)
def t(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
whole = t(lambda: m.sam_encode(sam))
def split():
    return [m.sam_encode(sam[i:i + 1]) for i in range(B)]
one = t(split)
a = m.sam_encode(sam); b = torch.cat(split())
print(f"{mode} B={B}: one call {whole:.2f} ms ({whole / B:.2f} per image), {B} calls {one:.2f} ms ({one / B:.2f} per image), identical {torch.equal(a, b)}")

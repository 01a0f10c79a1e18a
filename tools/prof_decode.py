"""41-token C2 generate, overlap and graphs off, for rocprofv3 --kernel-trace --stats: the decode GEMVs alone on the chip
(the `isolated` figure of bench.py's roofline; tools/collect_profiles.sh -> profiles/r0N_decode_isolated_kernel_stats.csv)."""
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
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode='perf', max_batch=1, max_seg=2); m.config.eos_token_id = None
sizes, H, W = [(1024, 1024)], [1024], [1024]
m.set_overlap(False); m.set_graphs(False)
for _ in range(3):
    m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=41)
torch.cuda.synchronize()

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
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode='perf', max_batch=1, max_seg=2)
emb = torch.randn(1, 320, 4096, device=dev) * 0.02
which = sys.argv[1]
e = m.sam_encode(sam)[0]; pp = torch.randn(1, 256, device=dev)
f = {'mask': lambda: m.mask_decode(e, pp, (1024, 1024), (1024, 1024)), 'sam': lambda: m.sam_encode(sam), 'prefill': lambda: m.llm_forward(emb), 'clip': lambda: m.encode_images(clip)}[which]
for _ in range(4): f()
torch.cuda.synchronize()

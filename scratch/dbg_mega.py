import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b, config_tiny
from anyref_amd.synth import synth_state_dict
from anyref_amd.model import AnyRefForCausalLM
from bench import make_inputs
dev = torch.device('cuda', 0)
cfg = config_tiny()
sd = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.float32)
for mode in ('parity', 'perf'):
  for B in (1, 2):
    clip, sam, ids = make_inputs(cfg, B, seed=1); clip, sam = clip.to(dev), sam.to(dev)
    m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode=mode, max_batch=B, max_seg=2); m.config.eos_token_id = None
    sizes, H, W = [(224, 224)] * B, [224] * B, [224] * B
    outs = {}
    for g in (0, 1):
      for pd in (0, 1):
        m.set_graphs(bool(g)); m.set_persistent_decode(bool(pd))
        outs[(g, pd)] = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=12)
        torch.cuda.synchronize()
    r = outs[(0, 0)]
    for k, o in outs.items():
        print(mode, 'B', B, 'graphs,persistent', k, 'ids equal', torch.equal(o[0], r[0]), o[0][0, -6:].tolist(), flush=True)

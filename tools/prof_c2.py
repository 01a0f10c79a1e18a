"""One C2 generate per iteration for rocprofv3 --kernel-trace: overlap off by default (clean per-kernel durations),
`--overlap` to see what co-running does.  python tools/prof_c2.py [--overlap] [--iters N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b
from anyref_amd.model import AnyRefForCausalLM
from anyref_amd.synth import synth_state_dict
from bench import make_inputs
it = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 3
cfg = config_7b(); cfg.llm.max_seq = 512
sd = synth_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16)
clip, sam, ids = make_inputs(cfg, 1, seed=1)
clip, sam = clip.cuda(), sam.cuda()
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode="perf", max_batch=1, max_seg=2)
m.config.eos_token_id = None
o, _, _ = m.generate(clip, ids, sam, [(1024, 1024)], [1024], [1024], max_new_tokens=10)
m.set_seg_token_idx(int(o[0, ids.shape[1] + 2]))
m.set_overlap("--overlap" in sys.argv)
for _ in range(it + 1):
    m.generate(clip, ids, sam, [(1024, 1024)], [1024], [1024], max_new_tokens=10)
torch.cuda.synchronize()

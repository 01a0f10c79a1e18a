"""One C2 generate per iteration in a given arithmetic mode, overlap off (clean per-kernel durations under rocprofv3).
python tools/prof_mode.py MODE [--iters N] [--fan-in]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b
from anyref_amd.model import AnyRefForCausalLM
from anyref_amd.synth import synth_state_dict
from bench import make_inputs
mode = sys.argv[1]
it = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 2
cfg = config_7b(); cfg.llm.max_seq = 512
sd = synth_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16)
clip, sam, ids = make_inputs(cfg, 1, seed=1)
clip, sam = clip.cuda(), sam.cuda()
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode=mode, max_batch=1, max_seg=2)
del sd
m.config.eos_token_id = None
o, _, _ = m.generate(clip, ids, sam, [(1024, 1024)], [1024], [1024], max_new_tokens=10)
m.set_seg_token_idx(int(o[0, ids.shape[1] + 2]))
m.set_overlap("--overlap" in sys.argv)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(it):
    m.generate(clip, ids, sam, [(1024, 1024)], [1024], [1024], max_new_tokens=10)
torch.cuda.synchronize()
print(mode, "ms per generate", (time.perf_counter() - t0) / it * 1e3, "device GiB", m.device_bytes / 2**30)

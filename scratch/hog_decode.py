"""Lab: decode-step time (kernel-side stamps) while `n` CUs are held by a memory-silent hog kernel on another stream.
python scratch/hog_decode.py   (ANYREF_GEMV_GRID=n to change the GEMV grid; read once per process)"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b
from anyref_amd.model import AnyRefForCausalLM
from anyref_amd.synth import synth_state_dict
from bench import make_inputs
hog = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libhog.so"))
hog.hog_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_ulonglong, C.c_int]
cfg = config_7b(); cfg.llm.max_seq = 512
sd = synth_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16)
clip, sam, ids = make_inputs(cfg, 1, seed=1); clip, sam = clip.cuda(), sam.cuda()
m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode="perf", max_batch=1, max_seg=2); m.config.eos_token_id = None
sizes, H, W = [(1024, 1024)], [1024], [1024]
m.set_overlap(False)
side = torch.cuda.Stream()
sink = torch.zeros(4, device="cuda")
T = 12
big = torch.zeros(1 << 30, dtype=torch.uint8, device="cuda")
def run(n_hog, mode=0, pause=0):
    m.stamps_enable(True)
    m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T); m.stamps_read()
    torch.cuda.synchronize()
    if n_hog:
        rc = hog.hog_launch(C.c_void_p(side.cuda_stream), n_hog, 156 * 1024, 60.0, C.c_void_p(sink.data_ptr()), mode, C.c_void_p(big.data_ptr()), big.numel(), pause)
        assert rc == 0
    m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T)
    rows = m.stamps_read(); m.stamps_enable(False)
    torch.cuda.synchronize()
    ep = {}
    for r in rows:
        if r["epoch"] >= 0: ep.setdefault(r["epoch"], []).append(r)
    spans = sorted(max(x["t1_us"] for x in v) - min(x["t0_us"] for x in v) for v in ep.values())
    busy = sorted(sum(x["t1_us"] - x["t0_us"] for x in v) for v in ep.values())
    return spans[len(spans) // 2], busy[len(busy) // 2]
for mode, pause, n in ((0, 0, 0), (0, 0, 64), (1, 0, 64), (1, 0, 128), (2, 0, 64), (2, 4, 64), (2, 16, 64), (2, 64, 64), (2, 16, 128)):
    s, b = run(n, mode, pause)
    print(f"hog mode {mode} pause {pause:3d} CUs {n:3d}: decode step span {s:7.1f} us, GEMV busy {b:7.1f} us", flush=True)

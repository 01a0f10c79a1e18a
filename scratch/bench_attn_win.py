"""Lab: SAM-H window attention as the encoder launches it (25 windows x 16 heads, 14 x 14 tokens, hd 80, bias from the tables), f16."""
import ctypes as C, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
B, H, size, hd, ld = 25, 16, 14, 80, 128
S = size * size
q, k, v = (torch.randn(B, S, H, hd, device="cuda").half() for _ in range(3))
tab = (torch.randn(2, 2 * size, ld, device="cuda") * 0.3).half()
o = torch.empty(B, S, H, hd, device="cuda", dtype=torch.float16)
run = lambda: lib.anyref_op_attention_tab(2, None, P(q), P(k), P(v), P(o), B, H, S, hd, hd ** -0.5, P(tab[0]), P(tab[1]), ld, size, size)
assert run() == 0, lib.anyref_op_last_error()
for _ in range(3): run()
torch.cuda.synchronize()
ts = []
for r in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
t = sorted(ts)[3]
print(f"window attention: {t:.1f} us  checksum {o.float().abs().mean().item():.6f}", flush=True)

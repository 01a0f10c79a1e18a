"""Lab: SAM-H global attention (4096 tokens, 16 heads, hd 80, bias from the P buffer) as the encoder launches it, f16.
A/B: ANYREF_ATTN_G2=0 (general kernel, 8 waves x 16 queries) / 4 / 8 (two query blocks per wave, 4 or 8 waves)."""
import ctypes as C, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
B, H, size, hd = int(os.environ.get("B", "1")), 16, 64, 80
S = size * size
q, k, v = (torch.randn(B, S, H, hd, device="cuda").half() for _ in range(3))
p = torch.randn(H, B * S, 256, device="cuda") * 0.5
o = torch.empty(B, S, H, hd, device="cuda", dtype=torch.float16)
run = lambda: lib.anyref_op_attention_relp(2, None, P(q), P(k), P(v), P(o), B, H, S, hd, hd ** -0.5, P(p), 256, size, size)
assert run() == 0, lib.anyref_op_last_error()
for _ in range(3): run()
torch.cuda.synchronize()
ts = []
for r in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10 * 1e3)
t = sorted(ts)[2]
print(f"G2={os.environ.get('ANYREF_ATTN_G2', 'default')} B={B}: {t:.1f} us  {4.0 * B * H * S * S * hd / t / 1e6:.0f} TFLOP/s  checksum {o.float().abs().mean().item():.6f}", flush=True)

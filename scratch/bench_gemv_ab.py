"""Lab: decode GEMV shapes (7B) with padded weight rows (as the model lays them out), cold weights (rotating copies
beyond the Infinity Cache); prints the median of `rounds` timed blocks.  A/B by running it once per build / knob."""
import ctypes as C, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ANYREF_OPTEST_LDW_PAD", "64")
from anyref_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
pad = int(os.environ["ANYREF_OPTEST_LDW_PAD"])
shapes = [(12288, 4096, 0, 1), (4096, 4096, 0, 0), (11008, 4096, 1, 1), (4096, 11008, 0, 0), (32000, 4096, 0, 1)]
B = int(os.environ.get("B", "1"))
out = []
for N, K, dual, norm in shapes:
    nb = max(2, int(600e6 // (N * K * 2 * (2 if dual else 1))) + 1)
    Ws = [(torch.randn(N, K + pad, device='cuda') * 0.05).bfloat16() for _ in range(nb)]
    W2s = [(torch.randn(N, K + pad, device='cuda') * 0.05).bfloat16() for _ in range(nb)] if dual else None
    x = torch.randn(B, K, device='cuda'); gain = torch.ones(K, device='cuda'); y = torch.empty(B, N, device='cuda')
    def run(i):
        lib.anyref_op_gemv(1, None, P(x), P(gain) if norm else None, 1e-6, P(Ws[i % nb]), P(W2s[i % nb]) if dual else None, None, P(y), None, B, N, K, 0)
    for i in range(nb): run(i)
    torch.cuda.synchronize()
    ts = []
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 40
        e0.record()
        for i in range(n): run(i)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    ts.sort()
    out.append(f"{ts[len(ts)//2]:6.2f}")
print(os.environ.get("TAG", ""), "qkv o gate/up down lm_head (us, median of 7):", " ".join(out), flush=True)

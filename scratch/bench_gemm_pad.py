"""Lab: do the SAM GEMMs care about the weight row stride (power-of-two-ish strides aliasing onto few channels)?
Run with ANYREF_OPTEST_LDW_PAD=0 / 64: W is allocated [N, K + pad] either way."""
import ctypes as C, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pad = int(os.environ.get("ANYREF_OPTEST_LDW_PAD", "0"))
from anyref_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
def timeit(fn, n=32):
    for i in range(4): fn(i)
    torch.cuda.synchronize()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(i)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[2]
out = []
for name, M, N, K in (("qkv", 4096, 3840, 1280), ("fc1", 4096, 5120, 1280), ("fc2", 4096, 1280, 5120), ("proj", 4096, 1280, 1280)):
    NW = 24
    A = torch.randn(M, K, device='cuda').half()
    W = [(torch.randn(N, K + pad, device='cuda') * 0.05).half() for _ in range(NW)]
    bias = torch.randn(N, device='cuda')
    Cc = torch.empty(M, N, device='cuda', dtype=torch.float16)
    t = timeit(lambda i: lib.anyref_op_gemm(2, None, P(A), P(W[i % NW]), P(bias), P(Cc), None, None, M, N, K, 0, 0))
    out.append(f"{name} {t:.1f}")
print(f"ldw pad {pad}:", " ".join(out), flush=True)

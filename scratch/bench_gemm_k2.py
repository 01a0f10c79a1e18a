"""Lab: SAM qkv-shaped GEMM (4096 x 3840, f16, bias) over K: the K -> 0 intercept is prologue + epilogue, the slope the K tile."""
import ctypes as C, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for M, N in ((4096, 3840), (4096, 5120), (4096, 1280)):
    out = []
    for K in (64, 128, 256, 640, 1280, 2560, 5120):
        NW = 16
        A = torch.randn(M, K, device='cuda').half()
        W = [(torch.randn(N, K, device='cuda') * 0.05).half() for _ in range(NW)]
        bias = torch.randn(N, device='cuda')
        Cc = torch.empty(M, N, device='cuda', dtype=torch.float16)
        t = timeit(lambda i: lib.anyref_op_gemm(2, None, P(A), P(W[i % NW]), P(bias), P(Cc), None, None, M, N, K, 0, 0))
        out.append(f"K={K}: {t:.1f}")
    print(M, N, " ".join(out), flush=True)

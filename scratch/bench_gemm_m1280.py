"""Lab: LLaMA-7B prefill GEMM shapes at M = 1280 (C3: 4 sequences of 320) and M = 320, bf16, cold weights."""
import ctypes as C, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
def timeit(fn, n=16):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(i)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[2]
for M in (320, 1280):
    for name, N, K in (("qkv", 12288, 4096), ("o", 4096, 4096), ("gate/up", 22016, 4096), ("down", 4096, 11008)):
        NW = max(2, int(600e6 // (N * K * 2)) + 1)
        A = torch.randn(M, K, device='cuda').bfloat16()
        W = [(torch.randn(N, K, device='cuda') * 0.05).bfloat16() for _ in range(NW)]
        Cc = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        t = timeit(lambda i: lib.anyref_op_gemm(1, None, P(A), P(W[i % NW]), None, P(Cc), None, None, M, N, K, 0, 0))
        print(f"M={M} {name} {N}x{K}: {t:.1f} us  {2.0*M*N*K/t/1e6:.0f} TF", flush=True)

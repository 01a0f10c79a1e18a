"""Lab: what the store tail of the 256^2 GEMM costs: K = 64 with every row dropped (row_map = -1) vs stored."""
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
M, N = 4096, 3840
for K in (64, 1280):
    A = torch.randn(M, K, device='cuda').half()
    W = (torch.randn(N, K, device='cuda') * 0.05).half()
    bias = torch.randn(N, device='cuda')
    Cc = torch.empty(M, N, device='cuda', dtype=torch.float16)
    Cf = torch.empty(M, N, device='cuda', dtype=torch.float32)
    drop = torch.full((M,), -1, device='cuda', dtype=torch.int32)
    ident = torch.arange(M, device='cuda', dtype=torch.int32)
    t0 = timeit(lambda i: lib.anyref_op_gemm(2, None, P(A), P(W), P(bias), P(Cc), None, P(drop), M, N, K, 0, 0))
    t1 = timeit(lambda i: lib.anyref_op_gemm(2, None, P(A), P(W), P(bias), P(Cc), None, P(ident), M, N, K, 0, 0))
    t2 = timeit(lambda i: lib.anyref_op_gemm(2, None, P(A), P(W), P(bias), P(Cf), None, P(ident), M, N, K, 0, 1))
    print(f"K={K}: rows dropped {t0:.1f} us, f16 stored {t1:.1f} us, f32 stored {t2:.1f} us", flush=True)
x = torch.empty(M * N, device='cuda', dtype=torch.float16)
t = timeit(lambda i: x.fill_(1.0))
print(f"torch fill of the same {M*N*2/1e6:.1f} MB: {t:.1f} us")
x = torch.empty(16, device='cuda', dtype=torch.float16)
t = timeit(lambda i: x.fill_(1.0))
print(f"empty-ish launch: {t:.1f} us")

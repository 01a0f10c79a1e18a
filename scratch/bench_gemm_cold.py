"""Lab: the SAM-H GEMM shapes (f16) with warm vs cold weights / activations, with and without bias + window row map."""
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
shapes = [("qkv", 4096, 3840, 1280, 0), ("proj", 4096, 1280, 1280, 0), ("fc1", 4096, 5120, 1280, 2), ("fc2", 4096, 1280, 5120, 0)]
NW = 48
for name, M, N, K, act in shapes:
    A = [torch.randn(M, K, device='cuda').half() for _ in range(NW)]
    W = [(torch.randn(N, K, device='cuda') * 0.05).half() for _ in range(NW)]
    bias = torch.randn(N, device='cuda')
    Cc = torch.empty(M + 1024, N, device='cuda', dtype=torch.float16)
    rm = (torch.arange(M, device='cuda', dtype=torch.int32) * 7919 % M).int().contiguous()  # a permutation (7919 is prime, M = 2^12)
    res = {}
    res["warm"] = timeit(lambda i: lib.anyref_op_gemm(2, None, P(A[0]), P(W[0]), None, P(Cc), None, None, M, N, K, act, 0))
    res["coldW"] = timeit(lambda i: lib.anyref_op_gemm(2, None, P(A[0]), P(W[i % NW]), None, P(Cc), None, None, M, N, K, act, 0))
    res["coldAW"] = timeit(lambda i: lib.anyref_op_gemm(2, None, P(A[i % NW]), P(W[i % NW]), None, P(Cc), None, None, M, N, K, act, 0))
    res["coldW+bias"] = timeit(lambda i: lib.anyref_op_gemm(2, None, P(A[0]), P(W[i % NW]), P(bias), P(Cc), None, None, M, N, K, act, 0))
    res["coldW+bias+rowmap"] = timeit(lambda i: lib.anyref_op_gemm(2, None, P(A[0]), P(W[i % NW]), P(bias), P(Cc), None, P(rm), M, N, K, act, 0))
    fl = 2.0 * M * N * K
    print(name, M, N, K, " ".join(f"{k} {v:.1f}us ({fl / v / 1e6:.0f} TF)" for k, v in res.items()), flush=True)

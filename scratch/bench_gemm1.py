import ctypes as C, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
for M, N, K in [(4096, 3840, 1280), (4096, 3840, 1280), (8192, 8192, 8192), (4096, 3840, 1280)]:
    for dist in ('randn', 'uniform'):
        A = (torch.randn(M, K, device='cuda') if dist == 'randn' else torch.rand(M, K, device='cuda') * 2 - 1).bfloat16()
        W = ((torch.randn(N, K, device='cuda') if dist == 'randn' else torch.rand(N, K, device='cuda') * 2 - 1) * 0.05).bfloat16()
        Cc = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        for n in (20, 100):
            for _ in range(3):
                lib.anyref_op_gemm(1, None, P(A), P(W), None, P(Cc), None, None, M, N, K, 0, 0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                lib.anyref_op_gemm(1, None, P(A), P(W), None, P(Cc), None, None, M, N, K, 0, 0)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            print(f"{M:6d} {N:6d} {K:6d} {dist:8s} n={n:3d} {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF", flush=True)

import ctypes as C, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
for M, N, K in [(4096, 1280, 5120), (4096, 1280, 1280), (4900, 1280, 1280), (4096, 5120, 1280), (4096, 3840, 1280)]:
    A = torch.randn(M, K, device='cuda').bfloat16()
    W = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
    bias = torch.randn(N, device='cuda'); resid = torch.randn(M, N, device='cuda')
    Cc = torch.empty(M, N, device='cuda', dtype=torch.float32)
    def run(): return lib.anyref_op_gemm(1, None, P(A), P(W), P(bias), P(Cc), P(resid), None, M, N, K, 0, 1)
    assert run() == 0, lib.anyref_op_last_error()
    torch.cuda.synchronize()
    ref = A.float() @ W.float().t() + bias + resid
    err = (Cc - ref).abs().max().item()
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"{M:6d} {N:6d} {K:6d} {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF  maxerr {err:.3e}", flush=True)

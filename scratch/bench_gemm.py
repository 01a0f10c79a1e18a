import ctypes as C, torch, sys, os
sys.path.insert(0, '.')
from anyref_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
shapes = [(4096, 3840, 1280), (4096, 1280, 1280), (4096, 5120, 1280), (4096, 1280, 5120), (4900, 3840, 1280), (4900, 1280, 1280),
          (320, 12288, 4096), (320, 4096, 4096), (320, 22016, 4096), (320, 4096, 11008),
          (257, 3072, 1024), (257, 1024, 1024), (257, 4096, 1024), (257, 1024, 4096), (8192, 8192, 8192)]
print("mode", "register-staged" if os.environ.get("ANYREF_GEMM_NO_GLDS") else "glds", "splitk off" if os.environ.get("ANYREF_GEMM_NO_SPLITK") else "")
for M, N, K in shapes:
    A = torch.randn(M, K, device='cuda').bfloat16(); W = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
    Cc = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    for _ in range(3):
        lib.anyref_op_gemm(1, None, P(A), P(W), None, P(Cc), None, None, M, N, K, 0, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        lib.anyref_op_gemm(1, None, P(A), P(W), None, P(Cc), None, None, M, N, K, 0, 0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{M:6d} {N:6d} {K:6d}  {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF")

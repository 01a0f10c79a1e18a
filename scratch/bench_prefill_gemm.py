import ctypes as C, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
pad = int(os.environ.get("ANYREF_OPTEST_LDW_PAD", "0"))
for M, N, K in [(320, 22016, 4096), (320, 12288, 4096), (320, 4096, 4096), (320, 4096, 11008)]:
    nb = max(2, int(600e6 // (N * K * 2)) + 1)     # rotate weights: cold in L2 / MALL like a real layer sweep
    if os.environ.get("WARM"): nb = 1            # same weights every call: MALL-resident (256 MB)
    A = torch.randn(M, K, device='cuda').bfloat16()
    Ws = [(torch.randn(N, K + pad, device='cuda') * 0.05).bfloat16() for _ in range(nb)]
    Cc = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    def run(i): return lib.anyref_op_gemm(1, None, P(A), P(Ws[i % nb]), None, P(Cc), None, None, M, N, K, 0, 0)
    for i in range(nb): assert run(i) == 0
    torch.cuda.synchronize()
    ref = A.float() @ Ws[(nb - 1) % nb][:, :K].float().t()
    err = (Cc.float() - ref).abs().max().item() / ref.abs().max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(40): run(i)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 40
    print(f"pad {pad:4d} {M:5d} {N:6d} {K:6d} {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF  rel err {err:.1e}", flush=True)

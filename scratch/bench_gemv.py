import ctypes as C, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
shapes = [(12288, 4096, 0, 1), (4096, 4096, 0, 0), (11008, 4096, 1, 1), (4096, 11008, 0, 0), (4096, 8192, 0, 0), (32000, 4096, 0, 1)]
print("grid", os.environ.get("ANYREF_GEMV_GRID"))
# rotate over several weight copies so the 256 MiB infinity cache cannot serve the stream
for N, K, dual, norm in shapes:
    nb = max(2, int(600e6 // (N * K * 2 * (2 if dual else 1))) + 1)
    pad = int(os.environ.get("ANYREF_OPTEST_LDW_PAD", "0"))
    Ws = [(torch.randn(N, K + pad, device='cuda') * 0.05).bfloat16() for _ in range(nb)]
    W2s = [(torch.randn(N, K, device='cuda') * 0.05).bfloat16() for _ in range(nb)] if dual else None
    x = torch.randn(1, K, device='cuda'); gain = torch.ones(K, device='cuda'); y = torch.empty(1, N, device='cuda')
    def run(i):
        lib.anyref_op_gemv(1, None, P(x), P(gain) if norm else None, 1e-6, P(Ws[i % nb]), P(W2s[i % nb]) if dual else None, None, P(y), None, 1, N, K, 0)
    for i in range(nb): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 40
    e0.record()
    for i in range(n): run(i)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    by = N * K * 2 * (2 if dual else 1)
    print(f"N={N:6d} K={K:6d} dual={dual} {ms*1e3:8.1f} us  {by/ms/1e6:8.1f} GB/s  ({by/1e6:.0f} MB)")

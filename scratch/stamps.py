import ctypes as C, torch, sys, numpy as np
sys.path.insert(0, '.')
from anyref_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
lib.anyref_dbg_read_stamps.argtypes = [C.c_void_p]
for (N, K, norm) in [(4096, 4096, 0), (12288, 4096, 1), (4096, 11008, 0)]:
    Ws = [(torch.randn(N, K, device='cuda') * 0.05).bfloat16() for _ in range(4)]
    x = torch.randn(1, K, device='cuda'); gain = torch.ones(K, device='cuda'); y = torch.empty(1, N, device='cuda')
    for i in range(8):
        lib.anyref_op_gemv(1, None, P(x), P(gain) if norm else None, 1e-6, P(Ws[i % 4]), None, None, P(y), None, 1, N, K, 0)
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    lib.anyref_dbg_read_stamps(buf.ctypes.data_as(C.c_void_p))
    st = buf.reshape(4096, 8)[:, :5].astype(np.int64)
    t0 = st[:, 0].min()
    rel = (st - t0) * 10 / 1000.0   # 100 MHz ticks -> us
    print(f"N={N} K={K} norm={norm}: kernel span {rel[:,4].max():.2f} us")
    for i, nm in enumerate(['start', 'first loads issued', 'x staged', 'first item done', 'end']):
        c = rel[:, i]
        print(f"   {nm:20s} min {c.min():6.2f} med {np.median(c):6.2f} max {c.max():6.2f}")

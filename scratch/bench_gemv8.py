"""Decode GEMV at 5 .. 8 rows: the matrix-core kernel vs the packed-dot / fp8 two-pass form (ANYREF_GEMV_MFMA=0), per shape.
python scratch/bench_gemv8.py [B]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd import _lib
from anyref_amd.quant import quantize_rows_fp8
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for fp8 in (0, 1):
    shapes = [(64, 5120, 0, 1), (64, 5120, 0, 0), (64, 13824, 0, 0)] if os.environ.get('GEMV8_FIXED') else None
    for N, K, dual, norm in shapes or [(15360, 5120, 0, 1), (5120, 5120, 0, 0), (13824, 5120, 1, 1), (5120, 13824, 0, 0), (12288, 4096, 0, 1), (4096, 11008, 0, 0)]:
        nb = max(2, int(600e6 // (N * K * (1 if fp8 else 2) * (2 if dual else 1))) + 1)
        x = torch.randn(B, K, device="cuda"); gain = torch.ones(K, device="cuda"); y = torch.empty(B, N, device="cuda")
        Ws = []
        for _ in range(nb):
            w = torch.randn(N, K, device="cuda") * 0.03
            w2 = torch.randn(N, K, device="cuda") * 0.03 if dual else None
            if fp8:
                q, s = quantize_rows_fp8(w.cpu()); q2, s2 = quantize_rows_fp8(w2.cpu()) if dual else (None, None)
                Ws.append((q.cuda(), s.cuda(), q2.cuda() if dual else None, s2.cuda() if dual else None))
            else:
                Ws.append((w.bfloat16(), None, w2.bfloat16() if dual else None, None))
        def run(i):
            w, s, w2, s2 = Ws[i % nb]
            if fp8:
                return lib.anyref_op_gemv_fp8(None, P(x), P(gain) if norm else None, 1e-6, P(w), P(w2), P(s), P(s2), P(y), None, B, N, K)
            return lib.anyref_op_gemv(1, None, P(x), P(gain) if norm else None, 1e-6, P(w), P(w2), None, P(y), None, B, N, K, 0)
        for i in range(nb): assert run(i) == 0, lib.anyref_op_last_error()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40): run(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 40 * 1e3
        byt = N * K * (1 if fp8 else 2) * (2 if dual else 1)
        print(f"B={B} {'fp8 ' if fp8 else 'bf16'} N={N:6d} K={K:6d} dual={dual} {us:8.1f} us {byt / us / 1e6:7.2f} TB/s", flush=True)

import ctypes as C, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd import _lib
lib = _lib.load()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
print('RB', os.environ.get('ANYREF_ATTN_RB'))
for (B, H, S, hd, size, causal, name) in [(25, 16, 196, 80, 14, 0, 'sam window'), (25, 16, 196, 80, 0, 0, 'window norel'), (25, 16, 192, 80, 0, 0, 'win192 norel'), (1, 16, 4096, 80, 64, 0, 'sam global'), (1, 16, 4096, 80, 0, 0, 'global norel'),
                                           (1, 16, 257, 64, 0, 0, 'clip'), (1, 32, 320, 128, 0, 1, 'llm prefill')]:
    q, k, v = (torch.randn(B, S, H, hd, device='cuda').bfloat16() for _ in range(3))
    o = torch.empty_like(q)
    rh = rw = None
    if size:
        rh = torch.randn(B, H, S, size, device='cuda'); rw = torch.randn(B, H, S, size, device='cuda')
    def run():
        rc = lib.anyref_op_attention(1, None, P(q), P(k), P(v), P(o), B, H, S, S, hd, hd ** -0.5, causal, None, P(rh), P(rw), size, size)
        assert rc == 0, lib.anyref_op_last_error()
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 4.0 * B * H * S * S * hd * (0.5 if causal else 1)
    print(f"{name:12s} {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF")

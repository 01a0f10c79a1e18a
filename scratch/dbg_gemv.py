import ctypes as C, torch, sys
sys.path.insert(0, '.')
from anyref_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
for ty in (0, 1):
    B, N, K = 1, 8, 256
    x = torch.arange(K).float()[None] * 0 + 1.0
    W = torch.zeros(N, K)
    for n in range(N):
        W[n, :] = n + 1
    xd = x.cuda(); Wd = W.cuda().to(torch.bfloat16) if ty else W.cuda()
    y = torch.zeros(B, N, device='cuda')
    rc = lib.anyref_op_gemv(ty, None, P(xd), None, 1e-6, P(Wd), None, None, P(y), None, B, N, K, 0)
    torch.cuda.synchronize()
    print(ty, rc, y.cpu().tolist(), 'expect', (x @ W.t()).tolist())
    # one-hot x to see k mapping
    x = torch.zeros(1, K); x[0, 5] = 1
    W = torch.arange(N * K).float().view(N, K) / 100
    xd = x.cuda(); Wd = W.cuda().to(torch.bfloat16) if ty else W.cuda()
    rc = lib.anyref_op_gemv(ty, None, P(xd), None, 1e-6, P(Wd), None, None, P(y), None, B, N, K, 0)
    torch.cuda.synchronize()
    print(ty, rc, y.cpu().tolist(), 'expect', (x @ W.t()).tolist())

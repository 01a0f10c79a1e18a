import ctypes as C, torch, sys
sys.path.insert(0, '.')
from anyref_amd import _lib
lib = _lib.load()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
torch.manual_seed(0)
for (Sq, Sk, hd) in [(16, 32, 16), (16, 64, 16), (16, 33, 16)]:
    q, k, v = torch.randn(1, Sq, 1, hd), torch.randn(1, Sk, 1, hd), torch.randn(1, Sk, 1, hd)
    ref = torch.softmax(torch.einsum('bqhd,bkhd->bhqk', q, k) * hd ** -0.5, -1)
    ref = torch.einsum('bhqk,bkhd->bqhd', ref, v)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda(); o = torch.zeros(1, Sq, 1, hd, device='cuda')
    rc = lib.anyref_op_attention(0, None, P(qd), P(kd), P(vd), P(o), 1, 1, Sq, Sk, hd, hd ** -0.5, 0, None, None, None, 0, 0)
    torch.cuda.synchronize()
    err = (o.cpu() - ref).abs()[0, :, 0]
    print(Sq, Sk, 'rc', rc, 'max err', err.max().item())
    print('row max err', [round(x, 3) for x in err.max(1).values.tolist()])

import ctypes as C, torch, sys
sys.path.insert(0, '.')
from anyref_amd import _lib
lib = _lib.load()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
torch.manual_seed(0)
Sq, Sk, hd = 16, 32, 16
def run(q, k, v):
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda(); o = torch.zeros(1, Sq, 1, hd, device='cuda')
    lib.anyref_op_attention(0, None, P(qd), P(kd), P(vd), P(o), 1, 1, Sq, Sk, hd, hd ** -0.5, 0, None, None, None, 0, 0)
    torch.cuda.synchronize(); return o.cpu()[0, :, 0]
q, k, v = torch.randn(1, Sq, 1, hd), torch.randn(1, Sk, 1, hd), torch.randn(1, Sk, 1, hd)
o = run(q, k, torch.ones_like(v)); print('V=1: rows', [round(x, 4) for x in o[:, 0].tolist()])
o = run(q, torch.zeros_like(k), v); ref = v[0, :, 0].mean(0)
print('K=0: row err', [round(x, 4) for x in (o - ref).abs().max(1).values.tolist()])
# V = one-hot key index in column 0 -> output col0 = sum_j p_j * j : reveals which keys are mis-weighted
vv = torch.zeros_like(v); vv[0, :, 0, 0] = torch.arange(Sk).float()
o = run(q, k, vv)
p = torch.softmax(torch.einsum('bqhd,bkhd->bhqk', q, k) * hd ** -0.5, -1)[0, 0]
print('E[j] got', [round(x, 2) for x in o[:, 0].tolist()])
print('E[j] ref', [round(x, 2) for x in (p @ torch.arange(Sk).float()).tolist()])

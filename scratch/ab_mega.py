import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.config import config_7b, config_tiny
from anyref_amd.synth import synth_state_dict
from anyref_amd.model import AnyRefForCausalLM
from bench import make_inputs
dev = torch.device('cuda', 0)
def t(f, n=8):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
which = sys.argv[1] if len(sys.argv) > 1 else 'tiny'
if which == 'tiny':
    for mode in ('parity', 'perf'):
        cfg = config_tiny()
        sd = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.float32)
        for B in (1, 2):
            clip, sam, ids = make_inputs(cfg, B, seed=1); clip, sam = clip.to(dev), sam.to(dev)
            m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode=mode, max_batch=B, max_seg=2); m.config.eos_token_id = None
            sizes, H, W = [(224, 224)] * B, [224] * B, [224] * B
            outs = []
            for pd in (0, 1):
                m.set_persistent_decode(bool(pd))
                o = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=12)
                outs.append(o)
            same = torch.equal(outs[0][0], outs[1][0])
            print(mode, 'B', B, 'ids equal', same, outs[0][0][0, -12:].tolist(), outs[1][0][0, -12:].tolist(), flush=True)
else:
    cfg = config_7b(); cfg.llm.max_seq = 512
    sd = synth_state_dict(cfg, seed=0, device=dev, dtype=torch.bfloat16)
    clip, sam, ids = make_inputs(cfg, 1, seed=1); clip, sam = clip.to(dev), sam.to(dev)
    m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode='perf', max_batch=1, max_seg=2); m.config.eos_token_id = None
    sizes, H, W = [(1024, 1024)], [1024], [1024]
    out_ids, _, _ = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=10)
    m.set_seg_token_idx(int(out_ids[0, ids.shape[1] + 2]))
    ref = None
    for ov in (1, 0):
        m.set_overlap(bool(ov))
        for pd in (0, 1, 0, 1):
            m.set_persistent_decode(bool(pd))
            for T in (10, 40):
                o = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T)
                if T == 10:
                    if ref is None: ref = o
                    print('  equal ids', torch.equal(o[0], ref[0]), 'masks', torch.equal(o[1][0], ref[1][0]))
                print('overlap=%d persistent=%d T=%2d  %.2f ms' % (ov, pd, T, t(lambda: m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=T))), flush=True)

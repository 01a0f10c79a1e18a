"""Multi-process data parallel on the ONE GPU of the test box (SURVEY.md §8e rehearsal): two ranks share cuda:0 and
talk over gloo, through the same `DataParallelAnyRef.generate` / `bench.py --gpus N` code the 8-GPU RCCL run uses
(only the backend differs; `backend="nccl"` = RCCL stays the default for real N > 1)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, sys.argv[1])
from anyref_amd.config import config_tiny, IMAGE_TOKEN_INDEX
from anyref_amd.model import AnyRefForCausalLM
from anyref_amd.parallel import DataParallelAnyRef
from anyref_amd.synth import synth_state_dict
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=rank, world_size=world)
torch.cuda.set_device(0)
inp = torch.load(sys.argv[3])
cfg = config_tiny()
cfg.seg_token_idx = inp["seg"]
sd = synth_state_dict(cfg, seed=5, scale=0.05)
m = AnyRefForCausalLM.from_state_dict(cfg, {k: v.cuda() for k, v in sd.items()}, mode="parity", max_batch=3, max_seg=4)
m.config.eos_token_id = None
dp = DataParallelAnyRef(m)
ids, masks, rest = dp.generate(inp["clip"], inp["ids"], inp["sam"], inp["sizes"], inp["H"], inp["W"], max_new_tokens=5,
                               attention_masks=inp["mask"])
torch.cuda.synchronize()
torch.save(dict(ids=ids.cpu(), masks=None if masks is None else [t.cpu() for t in masks]), sys.argv[4] + f".{rank}")
dist.barrier()
dist.destroy_process_group()
print("OK", rank)
'''


def test_two_ranks_one_gpu_equal_single_process(tmp_path):
    """ragged global batch of 5 over 2 ranks (3 + 2): ids and full-resolution masks bit-identical to one process"""
    from anyref_amd.config import config_tiny, IMAGE_TOKEN_INDEX
    from anyref_amd.model import AnyRefForCausalLM
    from anyref_amd.synth import synth_state_dict
    cfg = config_tiny()
    sd = synth_state_dict(cfg, seed=5, scale=0.05)
    g = torch.Generator().manual_seed(6)
    n = 5
    clip = torch.randn(n, 3, 224, 224, generator=g)
    sam = torch.randn(n, 3, 224, 224, generator=g)
    rows = [torch.cat([torch.tensor([1, IMAGE_TOKEN_INDEX]), torch.randint(3, 990, (12 - b,), generator=g)]) for b in range(n)]
    L = max(len(r) for r in rows)
    ids = torch.zeros(n, L, dtype=torch.long)
    mask = torch.zeros(n, L, dtype=torch.bool)
    for b, r in enumerate(rows):
        ids[b, : len(r)] = r
        mask[b, : len(r)] = True
    sizes, H, W = [(224, 224 - 8 * b) for b in range(n)], [200 + 7 * b for b in range(n)], [180 + 11 * b for b in range(n)]
    m = AnyRefForCausalLM.from_state_dict(cfg, {k: v.cuda() for k, v in sd.items()}, mode="parity", max_batch=n, max_seg=4)
    m.config.eos_token_id = None
    o0, _, _ = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=5, attention_masks=mask)
    seg = int(o0[0, len(rows[0]) + 2])
    m.set_seg_token_idx(seg)
    want_ids, want_masks, _ = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=5, attention_masks=mask)
    assert want_masks is not None
    want_ids, want_masks = want_ids.cpu(), [t.cpu() for t in want_masks]
    inp = str(tmp_path / "in.pt")
    torch.save(dict(clip=clip, sam=sam, ids=ids, mask=mask, sizes=sizes, H=H, W=W, seg=seg), inp)
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    port = str(33500 + os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, port, inp, str(tmp_path / "out")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0 and "OK" in out, out
    for r in range(2):
        got = torch.load(str(tmp_path / "out") + f".{r}")
        assert torch.equal(got["ids"], want_ids), f"rank {r}: ids differ"
        assert len(got["masks"]) == n
        for b in range(n):
            assert torch.equal(got["masks"][b], want_masks[b]), f"rank {r}: masks of image {b} differ"


def test_bench_two_ranks_same_device_smoke():
    """`bench.py --gpus 2` end to end (tiny config, gloo, both ranks on cuda:0): one JSON line, aggregate over 2 ranks"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(34500 + os.getpid() % 2000), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--config", "tiny", "--dist-backend", "gloo", "--same-device"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["scaling"] == "weak" and res["value"] > 0
    assert res["config"]["global_batch"] == 2 * res["config"]["batch_per_gpu"]
    # who took part and what the per-step exchange cost (the first real SCALE record must show N ranks on N devices)
    assert [r["rank"] for r in res["ranks"]] == [0, 1] and all(r["device_uuid"] or r["pci_bus"] for r in res["ranks"])
    assert res["distinct_devices"] == 1               # (this rehearsal puts both ranks on cuda:0)
    assert res["collective"]["ms"] > 0 and res["collective"]["bytes_per_rank"] > 0 and res["collective"]["backend"] == "gloo"


_RCCL_WORKER = r'''
import os, sys, torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, sys.argv[1])
from anyref_amd.config import config_tiny, IMAGE_TOKEN_INDEX
from anyref_amd.model import AnyRefForCausalLM
from anyref_amd.parallel import gather_results
from anyref_amd.synth import synth_state_dict
assert not torch.cuda.is_initialized()                     # the rank picks its device before anything touches HIP
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=0, world_size=1, device_id=dev)
cfg = config_tiny()
sd = synth_state_dict(cfg, seed=5, scale=0.05)
g = torch.Generator().manual_seed(6)
n = 3
clip = torch.randn(n, 3, 224, 224, generator=g)
sam = torch.randn(n, 3, 224, 224, generator=g)
ids = torch.stack([torch.cat([torch.tensor([1, IMAGE_TOKEN_INDEX]), torch.randint(3, 990, (10,), generator=g)]) for _ in range(n)])
sizes, H, W = [(224, 224 - 8 * b) for b in range(n)], [200 + 7 * b for b in range(n)], [180 + 11 * b for b in range(n)]
m = AnyRefForCausalLM.from_state_dict(cfg, {k: v.cuda() for k, v in sd.items()}, mode="parity", max_batch=n, max_seg=4)
m.config.eos_token_id = None
o0, _, _ = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=5)
m.set_seg_token_idx(int(o0[0, ids.shape[1] + 2]))
(oids, masks, _), ex = m.generate(clip, ids, sam, sizes, H, W, max_new_tokens=5, _return_extras="low")
assert masks is not None
Lout = ids.shape[1] + 5
idp = torch.zeros(n, Lout, dtype=torch.long, device=dev)
idp[:, : oids.shape[1]] = oids
low, nseg, gids, glen = gather_results(ex["low_res"], ex["nseg"].to(dev), idp, ex["out_lens"].to(dev), n)   # RCCL, device tensors
torch.cuda.synchronize()
assert low.is_cuda and torch.equal(low, ex["low_res"]) and torch.equal(gids, idp)
assert torch.equal(nseg.cpu(), ex["nseg"]) and torch.equal(glen.cpu(), ex["out_lens"])
for b in range(n):                                           # full-resolution masks re-created from the gathered logits
    k = int(nseg[b])
    assert torch.equal(m.postprocess(low[b, :k], sizes[b], (H[b], W[b])), masks[b]), b
t = torch.ones(4, device=dev)
dist.all_reduce(t)                                           # bench.py's max-over-ranks reduction path
dist.barrier()
dist.destroy_process_group()
print("OK rccl", dist.is_nccl_available())
'''


@pytest.mark.skipif(os.environ.get("ANYREF_SKIP_RCCL_TEST") == "1", reason="ANYREF_SKIP_RCCL_TEST=1")
def test_gather_results_over_rccl_world_size_1(tmp_path):
    """`gather_results` over backend="nccl" (= RCCL) with a world of one on the test box's GPU: RCCL initialisation with
    `device_id`, the in-place `all_gather_into_tensor` on DEVICE tensors (the path gloo never takes: it stages through
    the host) and the all-reduce / barrier `bench.py --gpus N` ends with.  No scaling is measured here."""
    script = tmp_path / "w.py"
    script.write_text(_RCCL_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, str(script), ROOT, str(35500 + os.getpid() % 2000)], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "OK rccl" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])

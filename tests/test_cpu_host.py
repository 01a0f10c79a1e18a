"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol the headers
declare, the headers and the ctypes table agree, config/weight-name plumbing, and the DP
shard/gather logic over gloo with world_size 2."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()


def test_library_builds_and_exports_every_declared_symbol():
    _build()
    from anyref_amd import _lib
    lib = _lib.load()
    declared = set()
    for hdr in ("anyref_hip.h", "anyref_hip_ops.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        declared |= set(re.findall(r"\b(anyref_[a-z0-9_]+)\s*\(", txt))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))


def test_config_struct_matches_header_field_order():
    from anyref_amd import _lib
    txt = open(os.path.join(ROOT, "include", "anyref_hip.h")).read()
    body = txt[txt.index("typedef struct anyref_config {"): txt.index("} anyref_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(None, 1)[1].split(","):
            names.append(part.strip().split("[")[0])
    assert names == [f[0] for f in _lib.AnyrefConfig._fields_]


def test_no_gpu_means_loud_failure_not_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    _build()
    from anyref_amd.config import config_tiny
    from anyref_amd.model import AnyRefForCausalLM
    with pytest.raises(RuntimeError):
        AnyRefForCausalLM(config_tiny())


def test_weight_names_cover_reference_keys():
    from anyref_amd.config import config_7b
    from anyref_amd.synth import weight_shapes
    names = {n for n, _, _ in weight_shapes(config_7b())}
    for must in ("model.visual_model.image_encoder.blocks.31.attn.rel_pos_h",
                 "model.visual_model.mask_decoder.output_hypernetworks_mlps.3.layers.2.weight",
                 "model.visual_model.prompt_encoder.pe_layer.positional_encoding_gaussian_matrix",
                 "model.text_hidden_fcs.0.0.weight", "model.text_hidden_fcs.0.2.bias", "model.audio_projector.weight",
                 "model.layers.31.mlp.down_proj.weight", "lm_head.weight", "model.mm_projector.weight",
                 "model.vision_tower.vision_tower.vision_model.encoder.layers.22.self_attn.q_proj.weight"):
        assert must in names, must
    # hidden_states[-2] of a 24-layer tower: layer 23 is never run, so never requested
    assert not any(".encoder.layers.23." in n for n in names)


def test_shard_range_partitions_exactly():
    from anyref_amd.parallel import shard_range
    for n in (1, 7, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from anyref_amd.parallel import gather_results, shard_range
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=rank, world_size=world)
n, L = int(sys.argv[3]), 8
g = torch.Generator().manual_seed(0)
low_all = torch.randn(n, 2, L, L, generator=g)
nseg_all = torch.randint(0, 3, (n,), generator=g).int()
ids_all = torch.randint(0, 100, (n, 11), generator=g)
len_all = torch.randint(5, 11, (n,), generator=g).int()
lo, hi = shard_range(n, rank, world)
low, nseg, ids, ln = gather_results(low_all[lo:hi], nseg_all[lo:hi], ids_all[lo:hi], len_all[lo:hi], n)
assert torch.equal(low, low_all) and torch.equal(nseg, nseg_all) and torch.equal(ids, ids_all) and torch.equal(ln, len_all)
dist.barrier()
dist.destroy_process_group()
print("OK", rank)
'''


@pytest.mark.parametrize("n", [4, 5])
def test_dp_gather_gloo_world2(tmp_path, n):
    """every rank ends up with the global batch in order, ragged tail included"""
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000 + n)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, port, str(n)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0 and "OK" in out, out


_DP_WORKER = r'''
import os, sys, types, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from anyref_amd.parallel import DataParallelAnyRef, shard_range
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=rank, world_size=world)
n, T, L, G = int(sys.argv[3]), 3, 6, 4

class FakeModel:
    """per-GPU model stand-in with the mirror's generate(..., _return_extras=True) contract, CPU tensors"""
    device = torch.device("cpu")
    max_seg = 2
    cfg = types.SimpleNamespace(sam=types.SimpleNamespace(grid=G))
    calls = 0
    def generate(self, clip, ids, sam, sizes, H, W, audios=None, ref_images=None, max_new_tokens=128,
                 attention_masks=None, _return_extras=False):
        FakeModel.calls += 1
        b = clip.shape[0]
        assert b > 0, "an empty shard must not reach the model"
        key = clip[:, 0, 0, 0].long()                      # the global image index planted by the test
        out = torch.cat([ids, key[:, None] * 10 + torch.arange(max_new_tokens)[None]], 1)
        nseg = (key % 3).int()                              # 0, 1 or 2 masks per image
        low = key.float()[:, None, None, None] + torch.arange(self.max_seg).float()[None, :, None, None] * 0.5 \
            + torch.zeros(b, self.max_seg, 4 * G, 4 * G)
        masks = [low[i, : int(nseg[i])] for i in range(b)]
        return (out, masks, (None, None, None)), dict(low_res=low, nseg=nseg, out_lens=torch.full((b,), L + max_new_tokens).int())
    def postprocess(self, low, resized, orig):
        return low + 100.0                                  # marks "re-created from the gathered low-res logits"

clip = torch.zeros(n, 3, 2, 2); clip[:, 0, 0, 0] = torch.arange(n).float()
ids = torch.arange(n * L).reshape(n, L)
dp = DataParallelAnyRef(FakeModel())
out_ids, masks, rest = dp.generate(clip, ids, torch.zeros(n, 3, 2, 2), [(8, 8)] * n, [8] * n, [8] * n, max_new_tokens=T)
lo, hi = shard_range(n, rank, world)
assert FakeModel.calls == (1 if hi > lo else 0)
assert rest == (None, None, None) and out_ids.shape == (n, L + T)
for b in range(n):
    assert out_ids[b, :L].tolist() == ids[b].tolist() and out_ids[b, L:].tolist() == [b * 10 + t for t in range(T)]
if all(b % 3 == 0 for b in range(n)):
    assert masks is None
else:
    for b in range(n):
        k = b % 3
        assert masks[b].shape[0] == k
        for j in range(k):
            assert float(masks[b][j, 0, 0]) == b + 0.5 * j + 100.0
dist.barrier()
dist.destroy_process_group()
print("OK", rank)
'''


@pytest.mark.parametrize("n", [1, 2, 5])
def test_data_parallel_wrapper_gloo_world2(tmp_path, n):
    """`DataParallelAnyRef.generate` itself over gloo with 2 ranks: a global batch smaller than the world (a rank
    with NO images still joins both collectives), an even split, and a ragged tail; every rank returns the global
    result in order, masks re-created from the gathered low-res logits."""
    script = tmp_path / "dp.py"
    script.write_text(_DP_WORKER)
    port = str(31500 + os.getpid() % 2000 + n)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, port, str(n)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0 and "OK" in out, out


def test_oracle_eval_steps_known_answers():
    """SURVEY.md §8 f-1 / f-2 restatements against hand-computed values (utils/utils.py:79-91,
    utils/refer_seg.py:560-570)."""
    import torch
    from oracle import anyref_oracle as O
    out = torch.tensor([[1, 1, 0, 0, 1, 0]])
    tgt = torch.tensor([[1, 0, 0, 255, 255, 1]])
    i, u, t = O.intersection_and_union(out, tgt, 2, ignore_index=255)
    # ignored pixels leave every histogram; kept pixels: (1,1) (1,0) (0,0) (0,1)
    assert i.tolist() == [1.0, 1.0] and t.tolist() == [2.0, 2.0] and u.tolist() == [3.0, 3.0]
    logits = torch.tensor([[2.0, 0.5, -1.0, -3.0, 4.0, 0.0]])     # sigmoid(0) = 0.5 is not > 0.5
    i2, u2, t2 = O.eval_mask_counts(logits, tgt)
    assert (i2.tolist(), u2.tolist(), t2.tolist()) == (i.tolist(), u.tolist(), t.tolist())
    img = torch.tensor([[[0, 128, 255]], [[255, 0, 128]]], dtype=torch.uint8)   # [2, 1, 3]
    x = O.sam_preprocess(img, 4)
    assert x.shape == (3, 4, 4) and float(x[:, 2:, :].abs().sum()) == 0 and float(x[:, :, 1:].abs().sum()) == 0
    assert abs(float(x[0, 0, 0]) - (0 - 123.675) / 58.395) < 1e-6 and abs(float(x[2, 1, 0]) - (128 - 103.53) / 57.375) < 1e-6


def test_oracle_avs_metrics_known_answers():
    """utils/pyutils.py:163-236 restatements against hand-computed values."""
    import torch
    from oracle import anyref_oracle as O
    # mask 0: P = {0, 1}, G = {1, 2}: inter 1, union 3.   mask 1: empty G, P = {3}: 3 of 4 pixels agree -> 3 / 4
    logits = torch.tensor([[[4.0, 2.0], [-2.0, -4.0]], [[-1.0, -1.0], [-1.0, 1.0]]])
    gt = torch.tensor([[[0, 1], [1, 0]], [[0, 0], [0, 0]]])
    want = (1 / 3 + 3 / 4) / 2
    assert abs(float(O.avs_mask_iou(logits, gt)) - want) < 1e-6
    # F-measure: only mask 0 counts.  Thresholds between sigmoid(-2) and sigmoid(2) pass pixels {0, 1}: tp 1,
    # prec 1/2, recall 1/2 -> F = 1.3 * .25 / (.15 + .5); thresholds up to sigmoid(-2) pass {0, 1, 2}: tp 2, prec 2/3,
    # recall 1 -> F = 1.3 * (2/3) / (0.2 + 1) = 0.7222 (the maximum)
    f = O.avs_fmeasure(logits, gt.float())
    assert abs(f - 1.3 * (2 / 3) / (0.3 * 2 / 3 + 1)) < 1e-6
    assert O.avs_fmeasure(logits[1:], gt[1:].float()) == 0.0       # nothing but empty ground truths


def test_avs_cut_logits_are_exact_crossovers():
    """anyref_amd/evalops.py turns `sigmoid(x) >= th` into `x >= cut`: every cut must be the first passing f32."""
    import torch
    from anyref_amd import evalops as E
    cuts, cut_pred = E._avs_cuts(255)
    th = torch.linspace(0, 1 - 1e-10, 255)
    assert cuts[0] == -float("inf") and bool((cuts[1:] > cuts[:-1]).all())
    below = torch.nextafter(cuts[1:], torch.full_like(cuts[1:], -float("inf")))
    pad = torch.zeros(64)                       # keeps the probes out of ATen's scalar end-of-tensor remainder
    sig = lambda x: torch.sigmoid(torch.cat([x, pad]))[:x.numel()]
    assert bool((sig(cuts[1:]) >= th[1:]).all()) and not bool((sig(below) >= th[1:]).any())
    c = torch.tensor([cut_pred])
    assert bool(sig(c) > 0.5) and not bool(sig(torch.nextafter(c, torch.tensor([-1.0]))) > 0.5)


def test_label_rows_are_cut_where_their_id_rows_were():
    """Batched callers pass left-padded ids and no mask (eval_referseg.py:124-137): the pad rule `ids != pad` picks the
    kept positions, and `labels` (-100 / ids, never equal to pad) must be cut at the SAME positions, or the LM loss and
    `where(labels > 0)[0][0]` (anyref.py:378) shift against the ids."""
    import torch
    from anyref_amd.config import config_tiny
    from anyref_amd.model import AnyRefForCausalLM
    m = AnyRefForCausalLM(config_tiny(), defer=True)
    pad = m.config.pad_token_id
    ids = torch.tensor([[pad, pad, pad, 1, 5, 6, 7], [1, 9, 8, 7, 6, 5, 4]])
    labels = torch.tensor([[-100, -100, -100, -100, -100, 6, 7], [-100, -100, -100, 7, 6, 5, 4]])
    keeps = []
    rows, lens = m._rows(ids, None, keep_out=keeps)
    lab, lab_lens = m._rows(labels, None, keep_in=keeps)
    assert lens.tolist() == lab_lens.tolist() == [4, 7]
    assert rows[0, :4].tolist() == [1, 5, 6, 7] and lab[0, :4].tolist() == [-100, -100, 6, 7]
    assert int((lab[0, :4] > 0).nonzero()[0]) == 2          # first supervised position, in the un-padded frame
    mask = torch.tensor([[0, 0, 0, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1, 1]]).bool()
    keeps2 = []
    rows2, _ = m._rows(ids, mask, keep_out=keeps2)
    assert torch.equal(rows2, rows) and [k.tolist() for k in keeps2] == [k.tolist() for k in keeps]

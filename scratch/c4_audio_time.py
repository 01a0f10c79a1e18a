"""BASELINE configs[3] (C4) timing: the ImageBind audio trunk as a PyTorch-ROCm module (north_star keeps it there) on the
GPU, against the 46.7 ms refer-seg forward it feeds -- the measurement behind the f-4 decision."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyref_amd.audio import ImageBindAudio
torch.manual_seed(0)
for dt in (torch.float32, torch.bfloat16):
    m = ImageBindAudio().eval().cuda().to(dt)
    mel = torch.randn(1, 3, 1, 128, 204, device="cuda", dtype=dt)
    for _ in range(5):
        m.get_audio_feature(mel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m.get_audio_feature(mel)
    torch.cuda.synchronize()
    print(f"ImageBind audio trunk (3 clips x 229 tokens, 12 x 768) {dt}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per call")

# the same trunk inside the HIP handle (f-4)
from anyref_amd.config import config_tiny, AudioTrunkConfig
from anyref_amd.model import AnyRefForCausalLM
from anyref_amd.synth import synth_state_dict
cfg = config_tiny(); cfg.audio_trunk = AudioTrunkConfig()
sd = synth_state_dict(cfg, seed=0, device="cuda")
for mode in ("perf", "parity"):
    m = AnyRefForCausalLM.from_state_dict(cfg, sd, mode=mode, max_batch=1)
    mel = torch.randn(1, 3, 1, 128, 204, device="cuda")
    for _ in range(5):
        m.audio_encode(mel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m.audio_encode(mel)
    torch.cuda.synchronize()
    print(f"ImageBind audio trunk in HIP ({mode}): {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per call")

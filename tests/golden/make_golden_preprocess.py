"""Golden vectors for the preprocessing in front of the hot path, made by the third-party code
the reference calls, run in the build container:

  * Pillow `Image.resize(..., BILINEAR)` at the size `ResizeLongestSide.get_preprocess_shape` gives -- the
    reference's `ResizeLongestSide` class itself is loaded from /root/reference (transforms.py needs torchvision's
    `resize` / `to_pil_image`, absent here: bound to their definition, PIL `Image.resize` / `Image.fromarray`);
  * transformers' `CLIPImageProcessor` (PIL backend, 5.15 here; 4.31 pinned by requirements.txt:29 does the same
    arithmetic) with `do_center_crop` as the datasets set it, then `F.interpolate` as utils/refer_seg.py:581-587.

    python tests/golden/make_golden_preprocess.py      # build container only

Inputs are regenerated from seeds by `preprocess_inputs`; stored outputs are strided to stay small.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {"landscape": (480, 640), "portrait": (1333, 750), "tiny_wide": (37, 91), "big": (1500, 2000), "square": (224, 224),
         "exact": (768, 1024)}


def preprocess_inputs(name):
    h, w = CASES[name]
    rng = np.random.default_rng(sum(map(ord, name)))
    base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if name == "big":                                       # smooth content as well as noise
        yy, xx = np.mgrid[0:h, 0:w]
        base[..., 0] = ((np.sin(xx / 37.0) * 0.5 + 0.5) * 255).astype(np.uint8)
        base[..., 1] = ((yy * 255) // h).astype(np.uint8)
    return base


def main():
    from PIL import Image
    import torch.nn.functional as F
    # the reference's ResizeLongestSide, with torchvision's two functions bound to what they are defined as
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvf.resize = lambda img, size: img.resize(tuple(size[::-1]), Image.BILINEAR)
    tvf.to_pil_image = lambda arr: Image.fromarray(arr)
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_transforms", "/root/reference/model/segment_anything/utils/transforms.py")
    T = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(T)
    for k in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
        del sys.modules[k]                                   # transformers probes for a real torchvision below
    rls = T.ResizeLongestSide(1024)
    from transformers import CLIPImageProcessor
    out = {}
    for name in CASES:
        img = preprocess_inputs(name)
        r = rls.apply_image(img)                                         # transforms.py:27-34
        out[name + ".sam_shape"] = np.array(r.shape)
        out[name + ".sam_u8"] = r[::8, ::8]
        out[name + ".sam_sum"] = np.int64(r.astype(np.int64).sum())
        for wo_crop in (True, False):
            proc = CLIPImageProcessor()
            if wo_crop:
                proc.do_center_crop = False                               # utils/refer_seg.py:301-302
            x = proc.preprocess(img, return_tensors="pt")["pixel_values"][0]
            if wo_crop:
                x = F.interpolate(x.unsqueeze(0), size=(224, 224), mode="bilinear", align_corners=False)[0]
            out[f"{name}.clip_{'wo' if wo_crop else 'crop'}"] = x.numpy()[:, ::5, ::5]
        print(name, img.shape, "->", r.shape)
    np.savez_compressed(os.path.join(HERE, "preprocess_pil.npz"), **out)


if __name__ == "__main__":
    main()

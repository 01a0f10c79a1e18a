#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_traffic.json

Units/corrections per MI355X_MICROARCH.md "HBM": the counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide coalesced streaming read, so fetch bytes = 2 * FETCH_SIZE * 1024;
WRITE_SIZE is exact for 16-byte streaming stores.
"""
import collections
import csv
import glob
import json
import sys

import re

def _glds(bm, bn, wm, wn, ns, ty):
    # gemm_glds_kernel<BM, BN, WM, WN, NS, W8 = false, PERSIST, T>: the walking (capped) variant counts with its tile
    return [rf"gemm_glds_kernel<{bm}, {bn}, {wm}, {wn}, {ns}, false, (false|true), anyref::{ty}>"]


TAGS = {  # bench.py roofline tag -> regexes of the kernel names (rocprofv3 Kernel_Name without "void anyref::")
    "gemv_bf16_x8": [r"gemv_kernel<anyref::bf16, 1, false, 8, false, (false|true)>"],      # plain and wave-pair split
    "gemv_bf16_x24": [r"gemv_kernel<anyref::bf16, 1, false, 24, false, (false|true)>"],
    "gemv_bf16_swiglu_x8": [r"gemv_kernel<anyref::bf16, 1, true, 8, false, (false|true)>"],
    "gemm_bf16_256x256": _glds(256, 256, 2, 4, 2, "bf16"), "gemm_f16_256x256": _glds(256, 256, 2, 4, 2, "f16"),
    "gemm_bf16_256x320": _glds(256, 320, 2, 4, 2, "bf16"), "gemm_f16_256x320": _glds(256, 320, 2, 4, 2, "f16"),
    "gemm_bf16_128x128g": _glds(128, 128, 2, 4, 2, "bf16"), "gemm_f16_128x128g": _glds(128, 128, 2, 4, 2, "f16"),
    "gemm_bf16_128x128s3": _glds(128, 128, 2, 4, 3, "bf16"), "gemm_f16_128x128s3": _glds(128, 128, 2, 4, 3, "f16"),
    "gemm_bf16_128x160s3": _glds(128, 160, 4, 2, 3, "bf16"), "gemm_f16_128x160s3": _glds(128, 160, 4, 2, 3, "f16"),
    "gemm_bf16_64x256": _glds(64, 256, 1, 4, 2, "bf16"), "gemm_bf16_64x256s3": _glds(64, 256, 1, 4, 3, "bf16"),
    "gemm_bf16_320x96": _glds(320, 96, 4, 2, 2, "bf16"), "gemm_bf16_320x64": _glds(320, 64, 4, 2, 3, "bf16"),
    "gemm_bf16_128x128": [r"gemm_kernel<anyref::bf16, 128, 128, 64>"],
    "gemm_bf16_64x128": [r"gemm_kernel<anyref::bf16, 64, 128, 64>"],
    "gemm_bf16_64x64": [r"gemm_kernel<anyref::bf16, 64, 64, 64>"],
    "attn_f16_hd80_w8": [r"attn_(walk_)?kernel<anyref::f16, 80, 8, 64, 0>"],
    "attn_f16_hd80_w13_res": [r"attn_kernel<anyref::f16, 80, 13, 48, 5>"],
    "attn_bf16_hd80_w8": [r"attn_(walk_)?kernel<anyref::bf16, 80, 8, 64, 0>"],
    "attn_bf16_hd80_w13_res": [r"attn_kernel<anyref::bf16, 80, 13, 48, 5>"],
    "attn_bf16_hd128": [r"attn_kernel<anyref::bf16, 128, 4, 0, 0>"],
    "attn_bf16_hd64_res": [r"attn_kernel<anyref::bf16, 64, 4, 64, 5>"],
    "decode_attn_bf16": [r"decode_attn_kernel<anyref::bf16, 128>"],
}


def agg(d, cname):
    out = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != cname:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void anyref::", "")
            out[k][0] += float(r["Counter_Value"])
            out[k][1] += 1
    return out


def match(table, pats):
    """sum / launches over every kernel name one of the regexes matches in full"""
    tot, n, names = 0.0, 0, []
    for k, (v, c) in table.items():
        if any(re.fullmatch(p, k) for p in pats):
            tot, n = tot + v, n + c
            names.append(k)
    return tot, n, names


fe, wr = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
res = {}
for tag, pats in TAGS.items():
    f, n, names = match(fe, pats)
    if n:
        w, nw, _ = match(wr, pats)
        res[tag] = {"kernel": " | ".join(sorted(names)), "launches": n, "fetch_bytes_per_launch": 2 * f / n * 1024,
                    "write_bytes_per_launch": w / max(nw, 1) * 1024,
                    "hbm_bytes_per_launch": 2 * f / n * 1024 + w / max(nw, 1) * 1024,
                    "note": "FETCH_SIZE doubled (gfx950 counts wide coalesced reads at half), KiB -> bytes"}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res, indent=1))

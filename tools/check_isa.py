#!/usr/bin/env python3
"""Static checks on the gfx950 code objects that correctness/performance depend on (no GPU needed).

  * no kernel may use scratch (private segment): a kernel with register spills returned wrong,
    run-to-run different results from a hipGraph replay / beside a second stream on this stack
    ("Compiler / runtime hazards")
  * the LDS-DMA GEMM must not wait vmcnt(0) in front of a tile's first ds_read (Makefile note on gemm.o)

usage: python tools/check_isa.py    (compiles anyref_amd/csrc/*.hip to assembly under /tmp)
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "anyref_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-I" + os.path.join(ROOT, "include"),
         "--cuda-device-only", "-S"]


def asm_of(src, strict):
    out = os.path.join(tempfile.gettempdir(), "anyref_isa_" + os.path.basename(src) + ".s")
    flags = FLAGS + ([] if strict else ["-fno-strict-aliasing"])
    subprocess.run(["hipcc"] + flags + [os.path.join(CSRC, src), "-o", out], check=True)
    return open(out).read()


def main():
    bad = 0
    allowed_scratch = ()
    for src in ("gemm.hip", "gemm_f16.hip", "gemm_sp16.hip", "gemv.hip", "attention.hip", "ops.hip"):
        s = asm_of(src, strict=src in ("gemm.hip", "gemm_f16.hip", "gemm_sp16.hip", "gemv.hip"))
        for name, seg in re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", s):
            if int(seg) > 0 and not any(a in name for a in allowed_scratch):
                print(f"FAIL {src}: {name} uses {seg} bytes of scratch")
                bad += 1
        if src in ("gemm.hip", "gemm_f16.hip", "gemm_sp16.hip"):
            for m in re.finditer(r"^(_ZN6anyref16gemm_glds_kernel\S*):\n(.*?)\.Lfunc_end", s, re.S | re.M):
                lines = m.group(2).split("\n")
                n = sum(1 for k, l in enumerate(lines)
                        if "s_waitcnt vmcnt(0)" in l and any("ds_read" in x for x in lines[k + 1:k + 4]))
                if n:
                    print(f"FAIL {src}: {m.group(1)} waits vmcnt(0) before {n} ds_read group(s)")
                    bad += 1
    print("isa check:", "FAILED" if bad else "ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

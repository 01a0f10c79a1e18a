#!/bin/bash
# Collect the round's profile evidence on the GPU box (run from the repo root through gpurun):
#   tools/collect_profiles.sh gpurun_out/v7
# Raw rocprofv3 output goes to /tmp (hundreds of MB); only the summaries land in the output directory, which is
# then copied into profiles/ by hand.  Kernel trace / stats and the PMC passes are SEPARATE runs (a combined run is
# refused on this pool), and the program comes directly after `--`.
set -o pipefail
OUT=${1:-gpurun_out/prof}
R=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
W=/tmp/anyref_prof; rm -rf $W; mkdir -p $W
python3 bench.py --stamps-out "$OUT/gemv_stamps_c2.csv" > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python3 bench.py --no-cpu-baseline --no-parity --no-secondary > $W/kt.json 2> $W/kt.err || exit 2
cp $W/kt/kt_kernel_stats.csv "$OUT/kt_kernel_stats.csv" 2>/dev/null || cp $(ls $W/kt/*kernel_stats.csv $W/kt/*/*kernel_stats.csv 2>/dev/null | head -1) "$OUT/kt_kernel_stats.csv"
rocprofv3 --kernel-trace --stats --output-format csv -d $W/pd -o pd -- python3 tools/prof_decode.py > /dev/null 2>&1 || exit 3
cp $W/pd/pd_kernel_stats.csv "$OUT/pd_kernel_stats.csv" 2>/dev/null || cp $(ls $W/pd/*kernel_stats.csv $W/pd/*/*kernel_stats.csv 2>/dev/null | head -1) "$OUT/pd_kernel_stats.csv"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-secondary --roofline-steps 0 > $W/pf.json 2> $W/pf.err || exit 4
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-secondary --roofline-steps 0 > $W/pw.json 2> $W/pw.err || exit 5
python3 tools/pmc_summary.py $W/pmc_fetch $W/pmc_write "$OUT/pmc_traffic.json" > /dev/null || exit 6
# per-(kernel, grid) table with the overlap off, and what the profiler sees of the two-stream overlap
rocprofv3 --kernel-trace --output-format csv -d $W/kt0 -o kt0 -- python3 tools/prof_c2.py --iters 3 > /dev/null 2>&1 || exit 7
python3 tools/trace_summary.py $W/kt0 5 45 > "$OUT/kernels_by_grid_overlap_off.txt" || exit 8
rocprofv3 --kernel-trace --output-format csv -d $W/kt1 -o kt1 -- python3 tools/prof_c2.py --iters 3 --overlap > /dev/null 2>&1 || exit 9
python3 tools/trace_summary.py $W/kt1 5 --overlap gemv_kernel > "$OUT/overlap_under_rocprof.txt" || exit 10
# the tolerance-meeting mode (parity16), overlap off: per-kernel durations of one C2 generate
rocprofv3 --kernel-trace --stats --output-format csv -d $W/p16 -o p16 -- python3 tools/prof_mode.py parity16 --iters 3 > $W/p16.log 2>&1 || exit 11
cp $W/p16/p16_kernel_stats.csv "$OUT/parity16_kernel_stats.csv" 2>/dev/null || cp $(ls $W/p16/*kernel_stats.csv $W/p16/*/*kernel_stats.csv 2>/dev/null | head -1) "$OUT/parity16_kernel_stats.csv"
# SQ counters per kernel on the final build (8 SQ slots per pass; program directly after `--`): where the wave cycles go, MFMA-busy
# cycles, LDS bank conflicts / LDS issue stalls -- overlap off, perf mode
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --output-format csv -d $W/sq1 -- python3 tools/prof_c2.py --iters 1 > $W/sq1.log 2>&1 || exit 12
python3 tools/pmc_sq_summary.py $W/sq1 40 > "$OUT/pmc_sq_per_kernel.txt" || exit 13
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $W/sq2 -- python3 tools/prof_c2.py --iters 1 > $W/sq2.log 2>&1 \
  && python3 tools/pmc_sq_summary.py $W/sq2 40 > "$OUT/pmc_sq_mfma_ops_per_kernel.txt"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --output-format csv -d $W/sq3 -- python3 tools/prof_mode.py parity16 --iters 1 > $W/sq3.log 2>&1 \
  && python3 tools/pmc_sq_summary.py $W/sq3 30 > "$OUT/pmc_sq_per_kernel_parity16.txt"
python3 - "$W" "$OUT" <<'PY'
import collections, csv, glob, sys
w, out = sys.argv[1], sys.argv[2]
for d, c in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{w}/{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                k = r["Kernel_Name"].split("(")[0]
                agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    with open(f"{out}/pmc_{c.lower()}_per_kernel.csv", "w") as fo:
        fo.write(f"kernel,launches,{c}_KiB_sum,{c}_KiB_per_launch\n")
        for k, (s, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            fo.write(f"\"{k}\",{n},{round(s, 1)},{round(s / n, 2)}\n")
PY
ls -la "$OUT"

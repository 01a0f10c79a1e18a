#!/usr/bin/env python3
"""Per-kernel SQ counter summary of a `rocprofv3 --pmc SQ_...` run: where the wave cycles go.
python tools/pmc_sq_summary.py <dir> [rows]"""
import collections, csv, glob, sys
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"]
        if "anyref::" not in nm:
            continue
        k = nm.replace("void anyref::", "").split("(")[0][:58] + f" g{int(r['Grid_Size']) // max(1, int(r['Workgroup_Size']))}" if "Grid_Size" in r else nm[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[k] += 1
names = sorted({c for v in agg.values() for c in v})
print("counters:", names)
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:top]
for k, v in rows:
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    parts = " ".join(f"{c.replace('SQ_', '')}={v[c] / wc:6.3f}" for c in names if c != "SQ_WAVE_CYCLES")
    print(f"{k:70s} n={cnt[k]:5d} wave_cyc={wc:12.3e} | {parts}")

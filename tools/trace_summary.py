#!/usr/bin/env python3
"""Summarise a `rocprofv3 --kernel-trace --output-format csv` run of scratch/prof_c2.py / bench.py.

    python tools/trace_summary.py <trace dir> <iterations> [rows]           per (kernel, grid) table
    python tools/trace_summary.py <trace dir> <iterations> --overlap KEY    co-running analysis for kernels whose
                                                                            name contains KEY (e.g. gemv_kernel)
    python tools/trace_summary.py <trace dir> <iterations> --gaps           device idle time inside the last generate call

The overlap analysis answers the measurement question of DESIGN.md §6: does the profiler see the SAM encoder (second
stream) running BESIDE the decode GEMVs, and what does a co-running launch cost?  For every launch of KEY it
computes the part of its [start, end) interval during which a kernel of ANOTHER queue was executing.
"""
import collections
import csv
import glob
import sys


def load(d):
    fs = glob.glob(d + "/*kernel_trace.csv") + glob.glob(d + "/*/*kernel_trace.csv")
    if not fs:
        sys.exit(f"no *kernel_trace.csv under {d}")
    rows = []
    for r in csv.DictReader(open(fs[0])):
        nm = r["Kernel_Name"]
        if "anyref::" not in nm:
            continue
        q = r.get("Queue_Id") or r.get("Stream_Id") or "0"
        rows.append(dict(name=nm.replace("void anyref::", ""), q=q, s=int(r["Start_Timestamp"]), e=int(r["End_Timestamp"]),
                         grid=(int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))))
    return rows


def main():
    d, n = sys.argv[1], int(sys.argv[2])
    rows = load(d)
    if "--overlap" in sys.argv:
        key = sys.argv[sys.argv.index("--overlap") + 1]
        queues = collections.Counter(r["q"] for r in rows)
        print("queues (kernel launches):", dict(queues))
        byq = collections.defaultdict(list)
        for r in rows:
            byq[r["q"]].append((r["s"], r["e"]))
        for q in byq:
            byq[q].sort()
        import bisect
        alone, co = collections.defaultdict(list), collections.defaultdict(list)
        for r in rows:
            if key not in r["name"]:
                continue
            ov = 0
            for q, iv in byq.items():
                if q == r["q"]:
                    continue
                i = bisect.bisect_left(iv, (r["s"], 0))
                for s, e in iv[max(0, i - 1):]:
                    if s >= r["e"]:
                        break
                    ov += max(0, min(e, r["e"]) - max(s, r["s"]))
            dur = r["e"] - r["s"]
            inst = r["name"].split("(")[0][:70]
            (co if ov > 0.5 * dur else alone)[inst].append(dur / 1e3)
        print(f"{'kernel':72s} {'alone n':>8s} {'avg us':>8s} {'co-run n':>9s} {'avg us':>8s} {'ratio':>6s}")
        for k in sorted(set(alone) | set(co)):
            a, c = alone.get(k, []), co.get(k, [])
            aa = sum(a) / len(a) if a else float("nan")
            cc = sum(c) / len(c) if c else float("nan")
            print(f"{k:72s} {len(a) / n:8.1f} {aa:8.2f} {len(c) / n:9.1f} {cc:8.2f} {cc / aa if a and c else float('nan'):6.2f}")
        return
    if "--gaps" in sys.argv:
        # idle time between consecutive kernels of the LAST generate call in the trace (all queues merged): where the
        # device waits for the host or for a dependent launch
        rows.sort(key=lambda r: r["s"])
        # the last iteration starts at the last clip_assemble (CLIP stem) launch
        starts = [i for i, r in enumerate(rows) if "clip_assemble" in r["name"]]
        it = rows[starts[-1]:]
        busy_end, gaps, busy = it[0]["s"], [], 0
        for a in it:
            if a["s"] > busy_end:
                gaps.append((a["s"] - busy_end, prev["name"][:50], a["name"][:50], (a["s"] - it[0]["s"]) / 1e6))
            busy_end = max(busy_end, a["e"])
            prev = a
        span = (busy_end - it[0]["s"]) / 1e6
        idle = sum(g[0] for g in gaps) / 1e6
        print(f"last iteration: span {span:.3f} ms, device idle {idle:.3f} ms in {len(gaps)} gaps, {len(it)} launches")
        hist = collections.Counter(min(int(g[0] / 1e3), 20) for g in gaps)
        print("gap histogram (us: count):", dict(sorted(hist.items())))
        by_pair = collections.defaultdict(lambda: [0, 0.0])
        for g in gaps:
            k = (g[1].split("<")[0].split("(")[0], g[2].split("<")[0].split("(")[0])
            by_pair[k][0] += 1
            by_pair[k][1] += g[0] / 1e3
        print("idle by (previous kernel -> next kernel), us:")
        for k, (c, t) in sorted(by_pair.items(), key=lambda kv: -kv[1][1])[:25]:
            print(f"  {t:9.1f} us in {c:5d} gaps  {k[0]:36s} -> {k[1]}")
        print("largest gaps:")
        for g in sorted(gaps, reverse=True)[:12]:
            print(f"  {g[0] / 1e3:8.1f} us at t={g[3]:7.3f} ms  {g[1]} -> {g[2]}")
        return
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    agg = collections.defaultdict(list)
    for r in rows:
        agg[(r["name"][:64],) + r["grid"]].append((r["e"] - r["s"]) / 1e3)
    tot = sum(sum(v) for v in agg.values()) / n / 1e3
    print(f"sum of kernel time: {tot:.3f} ms / iteration over {sum(len(v) for v in agg.values()) / n:.0f} launches")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
        print(f"{k[0]:64s} grid={k[1]:6d}x{k[2]}x{k[3]} calls/iter={len(v) / n:6.1f} avg_us={sum(v) / len(v):8.1f} ms/iter={sum(v) / n / 1e3:7.3f}")


if __name__ == "__main__":
    main()

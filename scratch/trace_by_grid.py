import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
n = int(sys.argv[2])
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    nm = r['Kernel_Name']
    if 'anyref::' not in nm: continue
    key = (nm.replace('void anyref::', '')[:60], int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
for k, v in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{k[0]:60s} grid={k[1]:6d}x{k[2]}x{k[3]} calls/iter={len(v)/n:6.1f} avg_us={sum(v)/len(v):8.1f} ms/iter={sum(v)/n/1e3:7.3f}")

import csv, sys, glob
f = glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
n = int(sys.argv[2])
tot = 0
for r in rows:
    if 'anyref::' not in r['Name']: continue
    t = float(r['TotalDurationNs']) / 1e6 / n
    tot += t
    print(f"{r['Name'].replace('void anyref::','')[:70]:70s} calls/iter={int(r['Calls'])/n:7.1f} ms/iter={t:7.3f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
print('total ms/iter', round(tot, 3))

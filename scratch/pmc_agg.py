import csv, glob, collections, sys
def agg(d):
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(d + '/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void anyref::', '')
            g = (k, r.get('Grid_Size', ''))
            out[g][r['Counter_Name']][0] += float(r['Counter_Value']); out[g][r['Counter_Name']][1] += 1
    return out
pat = sys.argv[2:]
o = agg(sys.argv[1])
for g, c in sorted(o.items()):
    if any(p in g[0] for p in pat):
        n = list(c.values())[0][1]
        print(g, 'launches', n, {k: round(v[0] / v[1]) for k, v in c.items()})

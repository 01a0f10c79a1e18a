"""Print calls / average / min of the kernels matching a pattern from a rocprofv3 kernel_stats.csv found under a directory."""
import csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
f = (glob.glob(d + "/*kernel_stats.csv") + glob.glob(d + "/*/*kernel_stats.csv"))[0]
for r in csv.DictReader(open(f)):
    if pat in r["Name"]:
        print(f"{int(r['Calls']):6d} avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:7.1f}  {r['Name'][:80]}")

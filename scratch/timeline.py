import csv, sys, collections
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if 'anyref::' in r['Kernel_Name']]
# find the last generate: split by im2col_patch (first kernel of sam fork) occurrences
starts = [i for i, r in enumerate(rows) if 'im2col_patch' in r['Kernel_Name']]
# each generate has 2 im2col_patch (sam, clip). take last two
i0 = starts[-2]
seg = rows[i0:]
t0 = min(int(r['Start_Timestamp']) for r in seg)
t1 = max(int(r['End_Timestamp']) for r in seg)
print('last generate wall (kernels) ms:', (t1 - t0) / 1e6, 'n kernels', len(seg))
by = collections.defaultdict(list)
for r in seg:
    by[r['Stream_Id']].append(r)
for sid, rs in by.items():
    a = min(int(r['Start_Timestamp']) for r in rs); b = max(int(r['End_Timestamp']) for r in rs)
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
    print(f'stream {sid}: n={len(rs)} span {(a-t0)/1e6:.2f}..{(b-t0)/1e6:.2f} ms busy {busy/1e6:.2f} ms')
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rs:
        k = r['Kernel_Name'].split('(')[0].replace('void anyref::', '')[:48]
        agg[k][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); agg[k][1] += 1
    for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f'    {k:50s} {t/1e6:8.2f} ms  n={n:5d} avg {t/n/1e3:7.1f} us')
# phases on the main stream: find markers
main = max(by.values(), key=len)
def first(name):
    for r in main:
        if name in r['Kernel_Name']: return (int(r['Start_Timestamp']) - t0) / 1e6
def last(name):
    v = None
    for r in main:
        if name in r['Kernel_Name']: v = (int(r['End_Timestamp']) - t0) / 1e6
    return v
print('embed_splice at', first('embed_splice'), ' first decode (embed_rows) at', first('embed_rows'), ' last argmax', last('argmax'), ' build_tokens', first('build_tokens'), 'postprocess end', last('postprocess'))
# decode step durations on the main stream: between consecutive embed_rows launches
ts = [(int(r['Start_Timestamp']) - t0) / 1e6 for r in main if 'embed_rows' in r['Kernel_Name']]
print('decode step starts (ms):', [round(x, 2) for x in ts])
print('step durations:', [round(b - a, 2) for a, b in zip(ts, ts[1:])])
other = [rs for sid, rs in by.items() if rs is not main]
if other:
    o = other[0]
    print('second stream span: %.2f .. %.2f' % ((int(o[0]['Start_Timestamp']) - t0) / 1e6, (max(int(r['End_Timestamp']) for r in o) - t0) / 1e6))

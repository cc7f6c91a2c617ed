import csv, collections, sys, glob
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]      # (rocprofv3 -d <dir> -o <name>: directly in <dir>, or one level down)
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('void ','')[:32]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'tgs' not in k: continue
    print(k, {c: round(sum(v)/len(v)/1e6, 2) for c, v in d.items()})

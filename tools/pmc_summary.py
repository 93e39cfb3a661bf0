import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")

import csv, sys, collections
f = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:32]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    for c, v in d.items():
        print(f"{k:34s} {c:12s} n={len(v):4d} mean={sum(v)/len(v):.6g} sum={sum(v):.6g}")

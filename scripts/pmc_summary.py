"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch, per kernel."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + '/*/*_counter_collection.csv'):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, c in acc.items():
            if not any(s in k for s in ('screen_kernel', 'mlp3_kernel', 'nearest_rows', 'nearest_direct', 'tokenize')): continue
            print(k)
            for name, v in sorted(c.items()):
                print(f"   {name:28s} {sum(v)/len(v):16.0f}   (n={len(v)})")

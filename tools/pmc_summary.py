"""Summarises rocprofv3 --pmc output (…_counter_collection.csv) per kernel: sum over launches and mean per launch of every counter."""
import csv, glob, sys, collections
root = sys.argv[1]
rows = collections.defaultdict(lambda: [0.0, set()])
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])
        rows[k][0] += float(r["Counter_Value"]); rows[k][1].add(r["Dispatch_Id"])
names = sorted({k[0] for k in rows})
ctrs = sorted({k[1] for k in rows})
only = sys.argv[2] if len(sys.argv) > 2 else ""
for n in names:
    if only and only not in n: continue
    print(n)
    for c in ctrs:
        if (n, c) in rows:
            s, d = rows[(n, c)]
            print("   %-28s sum %16.0f  launches %4d  mean %16.1f" % (c, s, len(d), s / max(len(d), 1)))

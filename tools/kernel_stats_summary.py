import csv,glob,sys,os
d=sys.argv[1]
f=max(glob.glob(d+"/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "ftn::" in n or "rocprim" in n: print("  %-60s calls %4s total %8.2f avg %8.3f ms" % (n.split("(")[0][-60:], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6))

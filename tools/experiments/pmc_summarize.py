import csv, glob, sys, collections
d = sys.argv[1]
agg = collections.OrderedDict()
for f in sorted(glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_trace" not in k: continue
        short = "any" if "<32, true" in k or ", true, false>" in k.split("(")[0] else "closest"
        key = (short, r["Counter_Name"])
        agg.setdefault(key, []).append(float(r["Counter_Value"]))
for (k, c), v in agg.items():
    print("%-8s %-40s n=%d last=%.6g" % (k, c, len(v), v[-1]))

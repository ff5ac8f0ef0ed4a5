import csv, glob, collections, sys, re
for d in sys.argv[1:]:
    fs = glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")
    if not fs: print(d, "missing"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        m = re.search(r"berg_kernel<(\w+), (\w+), (\d+)u, (\w+)>", k)
        if not m: continue
        key = "PH%s_%s" % (m.group(3), "fast" if m.group(4) == "true" else "gen")
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(k, " ".join("%s=%.4g" % (c, sum(x) / len(x)) for c, x in sorted(v.items())))

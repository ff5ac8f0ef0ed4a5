import csv, glob, collections, sys
for d in sys.argv[1:]:
    fs=glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")
    if not fs: print(d,"missing"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"]
        if "berg_kernel" not in k: continue
        key="fast" if ", true>" in k else "general"
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()):
        for c,vals in sorted(v.items()):
            print(d, k, c, "n=%d"%len(vals), "mean=%.4g"%(sum(vals)/len(vals)))

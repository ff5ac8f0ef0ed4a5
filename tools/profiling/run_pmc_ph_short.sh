cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ph1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/ph1 -- python3 $R/tools/profiling/bench_c2_phases.py > $R/gpurun_out/ph1.log 2>&1
tail -2 $R/gpurun_out/ph1.log
python3 - $R/gpurun_out/ph1 <<'PY'
import collections, csv, glob, sys, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "berg_kernel" not in k or int(r["Grid_Size"]) < 500000: continue
        m = re.search(r"berg_kernel<([^>]*)>", k)
        agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    w = sorted(cs["SQ_WAVES"])[len(cs["SQ_WAVES"])//2]
    print(k, " ".join("%s/wave=%.0f" % (c.replace("SQ_",""), sorted(v)[len(v)//2] / w) for c, v in sorted(cs.items()) if c != "SQ_WAVES"))
PY

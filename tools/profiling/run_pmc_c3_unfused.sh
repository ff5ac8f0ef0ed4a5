cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export KID_FL_UNFUSED=1
rm -rf $R/gpurun_out/pmc_c3u
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_c3u -- python3 $R/tools/profiling/bench_c3.py 1e7 4 > $R/gpurun_out/pmc_c3u.log 2>&1
python3 - $R/gpurun_out/pmc_c3u <<'PY'
import collections, csv, glob, sys, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]
        if int(r["Grid_Size"]) < 5000000: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()): print("    %-20s %14.0f (%d)" % (c, sorted(v)[len(v)//2], len(v)))
PY

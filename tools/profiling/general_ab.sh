#!/bin/bash
# mean duration of the general-build launches of the headline workload under rocprofv3, per library build: general_ab.sh build1.so ... ("default" = in-tree)
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  if [ "$so" = "default" ]; then unset KID_HIP_SO; else export KID_HIP_SO=$GRAFT_REPO_ROOT/$so; fi
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/tlg
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tlg -- python3 $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/tlg.log 2>&1
  python3 - $GRAFT_REPO_ROOT/gpurun_out/tlg "$so" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "berg_kernel" in r["Name"]:
            print(sys.argv[2], r["Name"].split("berg_kernel")[1].split(">")[0] + ">", r["Calls"], "avg_us %.1f min %.1f max %.1f" % (float(r["AverageNs"]) / 1e3, int(r["MinNs"]) / 1e3, int(r["MaxNs"]) / 1e3))
PY
done

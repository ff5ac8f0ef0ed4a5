#!/bin/bash
# PMC passes of config 3 (tools/profiling/bench_c3.py): gpurun -- 'bash tools/profiling/run_pmc_c3.sh'; prints per-launch means of the fused hot build
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export KID_C3_NO_ENV_STORE=1   # as bench.py's other_configs.c3 runs it
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAIT_ANY" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_c3_$i
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_c3_$i -- python3 $R/tools/profiling/bench_c3.py 1e7 4 > $R/gpurun_out/pmc_c3_$i.log 2>&1 || echo "pass $i failed: $(tail -2 $R/gpurun_out/pmc_c3_$i.log)"
done
python3 - $R/gpurun_out/pmc_c3_1 $R/gpurun_out/pmc_c3_2 $R/gpurun_out/pmc_c3_3 $R/gpurun_out/pmc_c3_4 <<'PY'
import collections, csv, glob, re, sys
agg = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if re.search(r"berg_kernel<false, false, 31u, true, \d>", r["Kernel_Name"])]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})[1:]      # the first launch allocates scratch: leave it out
        for r in rows:
            if int(r["Dispatch_Id"]) in ids: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print("%-28s %16.1f per launch (%d launches)" % (k, sum(v) / len(v), len(v)))
m = {k: sum(v) / len(v) for k, v in agg.items()}
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    n = 1.0e7
    print("HBM bytes per launch: fetch %.3e (FETCH_SIZE KB x 1024 x 2: gfx950 counts 128-B requests at 64 B), write %.3e, total %.3e = %.0f B per berg-step (algorithmic 320)" % (
        m["FETCH_SIZE"] * 2048.0, m["WRITE_SIZE"] * 1024.0, m["FETCH_SIZE"] * 2048.0 + m["WRITE_SIZE"] * 1024.0, (m["FETCH_SIZE"] * 2048.0 + m["WRITE_SIZE"] * 1024.0) / n))
if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m: print("VALU wave-instructions per berg-step: %.0f" % (m["SQ_INSTS_VALU"] / m["SQ_WAVES"]))
PY

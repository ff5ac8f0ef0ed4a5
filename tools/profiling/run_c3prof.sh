#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
export KID_C3_NO_ENV_STORE=1   # as bench.py's other_configs.c3 runs it
rm -rf $R/gpurun_out/prof_c3
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python3 $R/tools/profiling/bench_c3.py 1e7 20 > $R/gpurun_out/prof_c3.log 2>&1
tail -3 $R/gpurun_out/prof_c3.log
cp $R/gpurun_out/prof_c3/*/*kernel_stats.csv $R/gpurun_out/c3_kernel_stats.csv

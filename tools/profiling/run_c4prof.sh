#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python -m pytest tests -q -m gpu -k "c4 or fortran" > gpurun_out/t_c4.log 2>&1; tail -3 gpurun_out/t_c4.log
timeout -k 10 300 python tools/profiling/bench_c4.py 224 5 > gpurun_out/bench_c4.log 2>&1; tail -4 gpurun_out/bench_c4.log
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_c4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c4 -- python3 $R/tools/profiling/bench_c4.py 224 3 > $R/gpurun_out/prof_c4.log 2>&1
head -25 $R/gpurun_out/prof_c4/*/*_kernel_stats.csv | cut -c1-150

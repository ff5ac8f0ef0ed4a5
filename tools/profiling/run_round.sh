#!/bin/bash
# End-of-iteration GPU run (gpurun -- 'bash tools/profiling/run_round.sh [prefix]'): GPU tests, smoke, the separate --pmc passes
# of the bench command, then the default bench line (which quotes the PMC summary just taken: HBM traffic and VALU instructions
# per berg-step at the population it runs) and the rocprofv3 --stats summary of the same command.  Outputs land in gpurun_out/;
# the ones to keep are copied to profiles/ by tools/profiling/keep_round.sh.
R=$GRAFT_REPO_ROOT
P=${1:-r03}
cd $R
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_round.log 2>&1; echo "pytest exit=$?" | tee -a gpurun_out/pytest_gpu_round.log; tail -3 gpurun_out/pytest_gpu_round.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_round.log 2>&1; tail -1 gpurun_out/smoke_round.log
bash tools/profiling/run_pmc.sh > gpurun_out/round_pmc_summary.json 2> gpurun_out/round_pmc.err
# (on the box: what bench.py reads below) -- only a summary whose five passes all completed replaces the committed one
if python3 -c "import json,sys; sys.exit(0 if json.load(open('gpurun_out/round_pmc_summary.json')).get('complete') else 1)"; then
  cp gpurun_out/round_pmc_summary.json profiles/${P}_pmc_summary.json
else
  echo "PMC passes incomplete: profiles/${P}_pmc_summary.json left as it is" | tee -a gpurun_out/round_pmc.err
fi
cd $R
python bench.py > gpurun_out/bench_round.log 2>&1; tail -1 gpurun_out/bench_round.log | cut -c1-400
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_round
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_round -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof_round.log 2>&1
cp $R/gpurun_out/prof_round/*/*kernel_stats.csv $R/gpurun_out/round_kernel_stats.csv
head -c 600 $R/gpurun_out/round_kernel_stats.csv

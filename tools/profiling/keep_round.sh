#!/bin/bash
# copies the summaries of the last run_round.sh from gpurun_out/ into profiles/ under the round's prefix:  keep_round.sh r02
cd "$(dirname "$0")/../.."
P=${1:-r02}
tail -1 gpurun_out/bench_round.log > profiles/${P}_bench_line.json
cp gpurun_out/round_kernel_stats.csv profiles/${P}_bench_kernel_stats.csv
cp gpurun_out/round_pmc_summary.json profiles/${P}_pmc_summary.json
ls -la profiles/${P}_*

#!/bin/bash
# copies the summaries of the last run_round.sh from gpurun_out/ into profiles/ under the round's prefix:  keep_round.sh r02
cd "$(dirname "$0")/../.."
P=${1:-r03}
tail -1 gpurun_out/bench_round.log > profiles/${P}_bench_line.json
cp gpurun_out/round_kernel_stats.csv profiles/${P}_bench_kernel_stats.csv
python3 -c "import json,sys; sys.exit(0 if json.load(open('gpurun_out/round_pmc_summary.json')).get('complete') else 1)" && cp gpurun_out/round_pmc_summary.json profiles/${P}_pmc_summary.json
# the HBM bytes of the hot build alone, under the name the reviews ask for
python3 - profiles/${P}_pmc_summary.json profiles/${P}_hbm_traffic.json <<'PY'
import json, sys
s = json.load(open(sys.argv[1]))["hot"]
json.dump({"kernel": s["kernel"], "grid_size": s["grid_size_mean"], "per_launch": s["hbm"],
           "algorithmic_bytes_per_launch": 256.0 * s["grid_size_mean"], "algorithmic_note": "256 B per berg-step (SURVEY 8d) x lanes launched (the population rounded up to a workgroup)",
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes with --kernel-trace only (tools/profiling/run_pmc.sh)"}, open(sys.argv[2], "w"), indent=1)
PY
ls -la profiles/${P}_*

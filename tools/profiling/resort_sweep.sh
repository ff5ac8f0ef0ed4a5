#!/bin/bash
# the headline workload over 96 steps at several re-binning intervals: resort_sweep.sh 12 16 24 ...
for rs in "$@"; do
  python bench.py --no-cpu-baseline --no-other-configs --steps 96 --warmup 4 --resort $rs > /tmp/rs_$rs.json 2>/dev/null
  python - /tmp/rs_$rs.json $rs <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("resort %s: ms/step %.4f kernel %.4f" % (sys.argv[2], l["ms_per_step"], l["roofline"]["kernel_ms_avg"]))
PY
done

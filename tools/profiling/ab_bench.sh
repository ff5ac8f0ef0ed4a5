#!/bin/bash
# A/B of library builds on the headline workload: ab_bench.sh out_dir build1.so build2.so ...   ("default" = the in-tree library)
out=$1; shift
mkdir -p $out
for so in "$@"; do
  name=$(basename $so .so)
  if [ "$so" = "default" ]; then unset KID_HIP_SO; else export KID_HIP_SO=$PWD/$so; fi
  python bench.py --no-cpu-baseline --no-other-configs --steps 48 --warmup 12 > $out/$name.json 2> $out/$name.err || echo "FAILED $name" >> $out/summary.txt
  python - $out/$name.json $name >> $out/summary.txt <<'PY'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-12s ms/step %.4f kernel_ms %.4f frac %.4f %s" % (sys.argv[2], l["ms_per_step"], l["roofline"]["kernel_ms_avg"], l["roofline"]["frac"], l["library"]))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
done
cat $out/summary.txt

#!/bin/bash
# A/B of library builds on config 3 (and 4) as bench.py's other_configs runs them: ab_c3.sh out_dir build1.so ...  ("default" = in-tree)
out=$1; shift
mkdir -p $out
for so in "$@"; do
  name=$(basename $so .so)
  if [ "$so" = "default" ]; then unset KID_HIP_SO; else export KID_HIP_SO=$PWD/$so; fi
  python bench.py --no-cpu-baseline --steps 4 --warmup 2 > $out/$name.json 2> $out/$name.err || echo "FAILED $name" >> $out/summary.txt
  python - $out/$name.json $name >> $out/summary.txt <<'PY'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c3, c4 = l["other_configs"]["c3"], l["other_configs"]["c4"]
    print("%-12s c3 ms/step %.4f kernel %.4f frac %.4f | c4 ms/step %.4f | %s" % (sys.argv[2], c3["ms_per_step"], c3["roofline"]["kernel_ms_per_step"], c3["roofline"]["frac"], c4["ms_per_step"], l["library"]))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
done
cat $out/summary.txt

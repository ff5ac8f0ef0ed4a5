#!/bin/bash
# config 3 over two re-binning intervals (tools/profiling/bench_c3.py 1e7 32, environment store off): ab_c3_long.sh build1.so ... ("default" = in-tree)
export KID_C3_NO_ENV_STORE=1
for so in "$@"; do
  if [ "$so" = "default" ]; then unset KID_HIP_SO; else export KID_HIP_SO=$PWD/$so; fi
  echo "$so: $(python tools/profiling/bench_c3.py 1e7 ${STEPS:-32} $INTERVAL 2>&1 | grep 'ms/step')"
done

#!/bin/bash
# build and run the grid-barrier probe (on the GPU box: gpurun -- 'bash tools/profiling/barrier_probe.sh')
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -o gpurun_out/barrier_probe tools/profiling/barrier_probe.hip || exit 1
timeout -k 10 120 ./gpurun_out/barrier_probe | tee gpurun_out/barrier_probe.log

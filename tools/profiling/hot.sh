#!/bin/bash
# compile the hot build alone (tools/profiling/hot_only.hip) and print registers + static instruction mix; extra flags pass through
cd "$(dirname "$0")/../.."
mkdir -p /tmp/isa
# the plain hot builds (K = 1, 3: the default instantiation here) are compiled with the max-ILP scheduler in the product (csrc/Makefile)
SCHED="-mllvm -amdgpu-sched-strategy=max-ilp"
case "$*" in *KID_HOT_ARGS*) case "$*" in *",true,1"*|*",true,3"*) ;; *) SCHED="";; esac;; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics --cuda-device-only -S -DKID_EXPERIMENTS -DKID_EXP_MARKERS \
  -Rpass-analysis=kernel-resource-usage $SCHED "$@" -o /tmp/isa/hot_only.s tools/profiling/hot_only.hip 2>&1 | grep -E "error|TotalSGPRs|VGPRs:|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill|LDS Size" | sed 's/.*remark: //' | tr '\n' ' '
echo
python3 tools/profiling/isa_stats.py /tmp/isa/hot_only.s

#!/bin/bash
# PMC passes of bench.py (hot build of config 2 at the bench's population): gpurun -- 'bash tools/profiling/run_pmc.sh [bench flags]'
# Each --pmc pass is its own run with --kernel-trace only.  Summaries: tools/profiling/pmc_summary.py
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-other-configs $@"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_IFETCH GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc$i
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc$i -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc$i.log 2>&1 || echo "pass $i failed: $(tail -2 $R/gpurun_out/pmc$i.log)"
done
python3 $R/tools/profiling/pmc_summary.py $R/gpurun_out/pmc1 $R/gpurun_out/pmc2 $R/gpurun_out/pmc3 $R/gpurun_out/pmc4 $R/gpurun_out/pmc5

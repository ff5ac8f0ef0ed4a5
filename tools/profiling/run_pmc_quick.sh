#!/bin/bash
# the three SQ passes of run_pmc.sh only (instruction counts and wave cycles of the hot build): gpurun -- 'bash tools/profiling/run_pmc_quick.sh'
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-other-configs $@"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_IFETCH GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcq$i
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcq$i -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmcq$i.log 2>&1 || echo "pass $i failed: $(tail -2 $R/gpurun_out/pmcq$i.log)"
done
python3 $R/tools/profiling/pmc_summary.py $R/gpurun_out/pmcq1 $R/gpurun_out/pmcq2 $R/gpurun_out/pmcq3

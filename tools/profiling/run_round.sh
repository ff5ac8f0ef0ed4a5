#!/bin/bash
# End-of-iteration GPU run: parity tests, smoke, default bench, rocprof summary + PMC traffic of the same command.
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_round.log 2>&1; echo "pytest exit=$?" | tee -a gpurun_out/pytest_gpu_round.log; tail -3 gpurun_out/pytest_gpu_round.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_round.log 2>&1; tail -1 gpurun_out/smoke_round.log
python bench.py > gpurun_out/bench_round.log 2>&1; tail -1 gpurun_out/bench_round.log | cut -c1-300
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_round $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_round -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_round.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 --output-format csv -d $R/gpurun_out/pmc_valu -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_valu.log 2>&1
ls $R/gpurun_out/prof_round/*/

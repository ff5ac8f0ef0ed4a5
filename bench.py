#!/usr/bin/env python3
"""bench.py -- berg-steps/s of the MI355X evolve loop on BASELINE config 2 (configs[1]).

Workload: 1e6 synthetic bergs (random mass classes) per GPU on the 360x200 lat-lon ocean grid, default namelist
physics (RK4, drag + Coriolis + wave radiation + SSH slope, melt, rolling, rectangular mass spreading), fp64.
One *step* is one icebergs_run() worth of the hot path over the whole population: device-side forcing prepass,
accumulator zeroing, the fused per-berg kernel (evolve + thermodynamics + mass spreading), the cross-GPU sum of the
per-cell accumulators (RCCL all-reduce, N>1 only) and the 9-point gather.  Inputs are resident in HBM before the
timed region starts.  Weak scaling: every rank owns its own 1e6 bergs (different seeds), the grid is replicated.

Launch: `python bench.py --gpus 1 --steps K --warmup W`, or for N>1
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The schedule uses up to three streams next to RCCL's own; the HIP runtime maps streams onto 4 hardware queues by
# default and streams that share a queue serialise (measured: +68 us/step when the prepass shared one with the general
# build).  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ALGO_BYTES_PER_BERG_STEP = 256.0  # SURVEY.md 8d (config 2): 129 B read + 128 B written of per-berg SoA state
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6           # MI355X vector fp64 peak
# hot build, per berg-step: 637 FMA (x2) + 1066 ADD + ~1000 MUL fp64 lane-ops (SQ_INSTS_VALU_FMA_F64 / ADD_F64 x 64 lanes / 1e6 bergs)
FP64_FLOP_PER_BERG_STEP = 3300.0


def cpu_baseline(nbergs, nsteps):
    """The CPU oracle (scalar C restatement of the reference loop), 1 core, on a bounded cut of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib
    from icebergs_amd import synthetic as S
    grid, p, b = S.config_c2(n=nbergs, seed=2)
    o = oracle_lib.Oracle(grid, p)
    o.run_step(b, 1)  # touch everything once
    t0 = time.perf_counter()
    o.run_step(b, nsteps)
    dt = time.perf_counter() - t0
    return {"value": nbergs * nsteps / dt, "unit": "berg-steps/s", "cores": 1, "kind": "port",
            "sample": "%d bergs x %d steps of config 2 (same generator, seed 2), oracle/kid_oracle.c -O2, 1 thread, %.1f s"
                      % (nbergs, nsteps, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bergs", type=int, default=1_000_000, help="bergs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="N>1: keep the all-reduce and the gather on the critical path")
    ap.add_argument("--split-general", action="store_true", help="experiment: hot build in two halves, general build on the side stream")
    ap.add_argument("--force-collective", action="store_true", help="rehearsal: run the N>1 code path (RCCL all-reduce) with one rank")
    ap.add_argument("--resort", type=int, default=12, help="re-bin the bergs by cell every this many steps (move_berg_between_cells)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="N>1 rehearsal without N GPUs: every rank uses GPU 0 and the exchange goes through gloo (what is exercised is the N>1 code path, not its speed)")
    ap.add_argument("--advance-clock", action="store_true", help="kid_set_params with an advancing current_yearday before every step, as a model run does")
    ap.add_argument("--no-slow-lane", action="store_true", help="keep the general build between two hot builds (the plain schedule)")
    ap.add_argument("--cpu-bergs", type=int, default=500_000)
    ap.add_argument("--cpu-steps", type=int, default=16)
    args = ap.parse_args()

    import numpy as np
    import torch
    from icebergs_amd import synthetic as S, types as T
    from icebergs_amd.framework import Icebergs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        if args.force_collective and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)  # nccl == RCCL on ROCm

    # ---- workload: config 2, one shard of `bergs` per rank, replicated grid ----
    # weak scaling: every rank generates its own shard of the 8e7-class population (seed differs per rank)
    grid, params, bergs = S.config_c2(n=args.bergs, seed=2 + 1000 * rank)
    ib = Icebergs(grid, params, capacity=args.bergs, device=local_rank)
    # one explicit stream for the kernels, the torch copies and (N>1) the RCCL collectives that order themselves
    # against torch's current stream
    main_stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(main_stream)
    ib.set_stream(main_stream.cuda_stream)
    ib.upload_bergs(bergs)
    ib.set_store_environment(False)  # config 2 runs with ignore_traj=T: nobody reads berg%uo..hi (see include/kid.h)
    # forcing planes resident on the device (as an ocean/ice model on the same GPU would hand them over)
    forcing_dev = [torch.from_numpy(np.ascontiguousarray(grid["forcing"][name])).to(dev) for name in T.FORCING_NAMES]
    forcing_ptrs = [t.data_ptr() for t in forcing_dev]
    # accumulator block(s) in torch tensors so that RCCL can reduce them in place
    from icebergs_amd.distributed import ShardedStepper, PipelinedStepper, accumulator_views
    _, count = ib.accum_device_ptr()
    multi = world > 1 or args.force_collective
    slow_lane = not args.no_slow_lane and not args.split_general and not args.no_pipeline
    pipelined = (multi and not args.no_pipeline) or args.split_general or slow_lane
    if pipelined:
        # the all-reduce (N>1) and the gather of step k run on a second stream under the per-berg kernels of step k+1;
        # slow lane: so do the general-build launches (bergs that crossed a cell edge or bounced)
        stepper = PipelinedStepper(ib, params, dist, force_collective=args.force_collective, split_general=args.split_general, slow_lane=slow_lane, resort_interval=args.resort)
        nreduced = stepper.views[0][0].numel()
    else:
        acc_t = torch.zeros(count, dtype=torch.float64, device=dev)
        ib.bind_accum_buffer(acc_t.data_ptr(), count)
        stepper = ShardedStepper(ib, acc_t, ib.ncell, params.diag_mask, dist, params=params, force_collective=args.force_collective, resort_interval=args.resort)
        nreduced = stepper.planes.numel()

    def step():
        if args.advance_clock:
            params.current_yearday += params.dt / 86400.0
            ib.set_params(params)
        stepper.set_forcing_device(forcing_ptrs)     # applied by the step's prepass: per-cell forcing records + accumulator zeroing
        stepper.step()                               # per-berg kernels; RCCL all-reduce (N>1); 9-point gather

    def fence():
        stepper.flush()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ib.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_submit = time.perf_counter() - t0   # host time to enqueue the steps (diagnostic: is the host the bottleneck?)
    fence()
    elapsed = time.perf_counter() - t0
    berg_ms, launches, _ = ib.profile_get()
    ib.profile(False)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    n_slots, n_alive = ib.num_bergs()

    if rank == 0:
        total_bergs = args.bergs * world
        value = total_bergs * args.steps / elapsed
        # the dominant kernel is the hot build of berg_kernel: one launch per step, or two (one per half-population)
        # when the pipelined schedule is on; every berg goes through exactly one of them per step
        kern_ms = berg_ms / max(launches, 1)
        bergs_per_launch = args.bergs * args.steps / max(launches, 1)
        achieved = ALGO_BYTES_PER_BERG_STEP * bergs_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside the process, so this is
        # the committed rocprofv3 --pmc measurement of this same command (profiles/r01_hbm_traffic.json), valid only
        # for the population it was taken at
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as f:
                tj = json.load(f)
            if tj["bergs_per_launch"] == int(round(bergs_per_launch)):
                traffic, traffic_src = tj["hbm_traffic_bytes_per_launch"], "profiles/r01_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; gfx950 fetch correction x2)"
        except (OSError, KeyError, ValueError):
            pass
        line = {
            "metric": "berg_steps_per_sec", "value": value, "unit": "berg-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d synthetic bergs/GPU (random mass classes), 360x200 lat-lon ocean grid, "
                                   "RK4 drag+Coriolis+melt+mass spreading, dt=1800 s, ignore_traj=T" % args.bergs,
                       "bergs_per_gpu": args.bergs, "grid": "360x200", "sharding": "particle index, replicated grid",
                       "exchange": ("RCCL all-reduce of %d per-cell planes (%.1f MB) per step%s" % (nreduced // ib.ncell, nreduced * 8 / 1e6, ", overlapped with the next step's kernels" if pipelined else "")) if multi else "none (1 GPU)",
                       "bergs_alive_at_end": n_alive},
            "per_gpu_value": value / world, "host_submit_ms_per_step": 1e3 * t_submit / args.steps,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "berg_kernel<true, true, 14u, true> (RK4, old interp order, evolve|thermo|spread, hot build)",
                         "kernel_ms_avg": kern_ms, "kernel_launches": launches, "bergs_per_launch": bergs_per_launch,
                         "algorithmic_bytes_per_berg_step": ALGO_BYTES_PER_BERG_STEP,
                         # SURVEY 8d asks for GFLOP/s beside GB/s: the kernel sits at the ridge (AI ~ 12 flop/B).  Flops per berg-step
                         # from the committed PMC pass (profiles/r01_pmc_valu.txt: 2 x FMA_F64 + ADD_F64 + as many MUL_F64 as ADD, per lane)
                         "fp64_flop_per_berg_step": FP64_FLOP_PER_BERG_STEP,
                         "fp64_achieved_tflops": FP64_FLOP_PER_BERG_STEP * bergs_per_launch / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0,
                         "fp64_peak_tflops": FP64_PEAK_TFLOPS},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_bergs, args.cpu_steps)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ib.close()


if __name__ == "__main__":
    main()

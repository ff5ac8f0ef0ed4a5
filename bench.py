#!/usr/bin/env python3
"""bench.py -- berg-steps/s of the MI355X evolve loop at the size BASELINE.json's target names.

Workload: 1e7 synthetic bergs (random mass classes) per GPU -- the per-GPU share of BASELINE configs[4], with the physics
of configs[1] -- on the 360x200 lat-lon ocean grid, default namelist (RK4, drag + Coriolis + wave radiation + SSH slope,
melt, rolling, rectangular mass spreading), fp64.  One *step* is one icebergs_run() worth of the hot path over the whole
population: device-side forcing prepass, accumulator zeroing, the fused per-berg kernel (evolve + thermodynamics + mass
spreading), the cross-GPU sum of the per-cell accumulators (RCCL all-reduce, N>1 only) and the 9-point gather.  Inputs are
resident in HBM before the timed region starts.  Weak scaling: every rank owns its own 1e7 bergs (different seeds), the
grid is replicated.

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
N_SIMD, SIMD_CLOCK_HZ = 1024, 2.4e9   # 256 CUs x 4 SIMDs; one wave-instruction (64 lanes, fp64 included) issues per 4 cycles per SIMD
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r03_pmc_summary.json")   # rocprofv3 --pmc passes of this same command (tools/profiling/run_pmc.sh)


def _cpu_worker(nbergs, nsteps, seed):
    """one shard of the CPU baseline: the oracle (scalar C restatement of the reference loop) on `nbergs` bergs of config 2;
    says READY when set up, starts on a line from stdin, prints its own wall time"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_lib
    from icebergs_amd import synthetic as S
    grid, p, b = S.config_c2(n=nbergs, seed=seed)
    o = oracle_lib.Oracle(grid, p)
    o.run_step(b, 1)  # touch everything once
    print("READY", flush=True)
    sys.stdin.readline()
    t0 = time.perf_counter()
    o.run_step(b, nsteps)
    print(json.dumps({"seconds": time.perf_counter() - t0, "berg_steps": nbergs * nsteps}), flush=True)


def _run_cpu_shards(nshards, nbergs, nsteps):
    """`nshards` independent oracle processes (non-interacting bergs shard trivially) started together; wall time from the
    common start to the last one finishing"""
    import subprocess
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(nbergs), str(nsteps), str(2 + k)],
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for k in range(nshards)]
    for q in procs:
        line = q.stdout.readline()
        if line.strip() != "READY":
            raise RuntimeError("cpu baseline worker failed to start: %r" % line)
    t0 = time.perf_counter()
    for q in procs:
        q.stdin.write("go\n"); q.stdin.flush()
    outs = [json.loads(q.stdout.readline()) for q in procs]
    wall = time.perf_counter() - t0
    for q in procs:
        q.wait()
    return sum(o["berg_steps"] for o in outs), wall


def _host_core_share():
    """(cores this process may use, how that was found).  A GPU box hands a job one GPU's share of the host: the cgroup CPU quota
    when there is one, else the visible cores divided by the node's 8 GPUs."""
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:            # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()
            if q != "max":
                quota = max(1, int(round(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:   # v1
                q, per = int(f.read()), int(g.read())
                if q > 0:
                    quota = max(1, int(round(q / per)))
        except (OSError, ValueError):
            pass
    if quota is not None:
        return min(visible, quota), visible, "cgroup CPU quota"
    return max(1, visible // 8), visible, "visible cores / 8 GPUs of the node (no cgroup quota)"


def cpu_baseline(nbergs, nsteps):
    """The CPU oracle on the host cores of this node, on a bounded cut of the same workload: one core, then P independent
    shards on P cores, P = this job's share of the host (_host_core_share; KID_CPU_BASELINE_CORES overrides).
    Runs in child processes that never touch the GPU, before this process initialises it."""
    share, visible, rule = _host_core_share()
    P = max(1, min(visible, int(os.environ.get("KID_CPU_BASELINE_CORES", str(share)))))
    w1, t1 = _run_cpu_shards(1, nbergs, nsteps)
    per = max(nbergs // 2, 1)
    wP, tP = _run_cpu_shards(P, per, nsteps)
    return {"value": wP / tP, "unit": "berg-steps/s", "cores": P, "kind": "port",
            "one_core_value": w1 / t1, "cores_visible": visible, "cores_rule": rule,
            "sample": "config 2 (same generator): 1 core: %d bergs x %d steps (%.1f s); %d cores: %d independent shards of %d bergs x %d steps, "
                      "started together (%.1f s wall); oracle/kid_oracle.c -O2, one thread per shard; %d cores visible to this process, "
                      "%d used: %s" % (nbergs, nsteps, t1, P, P, per, nsteps, tP, visible, P, rule)}


def other_configs(np, torch, S, T, Icebergs, budget_s=75.0):
    """BASELINE configs 3 and 4 at their full size, timed here so that the driver's record holds them too (informational: the
    headline `value` is configs[4]'s per-GPU share above).  Each on a fresh handle, a few steps, inputs resident."""
    out, t_start = {}, time.perf_counter()
    # ---- config 3: 1e7 bergs, footloose profile (tests/footloose_tests/input.nml scaled to a 2000 x 1000 km periodic channel) ----
    try:
        n = 10_000_000
        grid, p, b = S.config_c3(n=n, seed=3, ni=2000, nj=1000, fl_style="fl_bits", capacity_factor=1.3, dt=10.0, spread=True, displace=True, periodic=True)
        ib = Icebergs(grid, p, capacity=len(b["lon"]), device=torch.cuda.current_device())
        ib.upload_bergs(b)
        ib.set_store_environment(False)   # ignore_traj=T: nobody reads berg%uo..hi back (the fused step interpolates for itself)
        # one whole re-binning interval is timed, its re-binning included: the kernel slows as the cell order decays between two
        # of them (1.03 -> 1.3 ms over 16 steps) and the re-binning itself is 3.4 ms at this size.  24 steps between two: measured
        # 1.39 ms/step against 1.43 at the library's default of 16 (tools/profiling/ab_c3_long.sh)
        interval = 24
        ib.set_resort_interval(interval)
        ib.move_berg_between_cells()   # (the first re-binning allocates its 6 GB of double buffers: 0.2 s on some boxes, not part of a step)
        ib.run(3); ib.sync()
        ib.profile(True)
        steps = interval
        t0 = time.perf_counter(); ib.run(steps); ib.sync(); dt = time.perf_counter() - t0
        ms, launches, _ = ib.profile_get(); ib.profile(False)
        n_slots, n_alive = ib.num_bergs()
        kern = ms / steps      # per step: the fused launch + the short launch over the step's new children
        out["c3"] = {"workload": "BASELINE configs[2]: 1e7 bergs, footloose (fl_bits, displaced children), Verlet, 2000x1000 periodic 1 km grid, dt=10 s",
                     "ms_per_step": 1e3 * dt / steps, "berg_steps_per_s": n * steps / dt, "steps": steps, "bergs_alive_at_end": int(n_alive),
                     "store_environment": False, "rebin_interval": interval,
                     "roofline": {"bound": "hbm", "algorithmic_bytes_per_berg_step": 320, "kernel_ms_per_step": kern, "kernel_launches": int(launches),
                                  "achieved": 320.0 * n / (kern * 1e-3) / 1e9 if kern > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": (320.0 * n / (kern * 1e-3) / 1e9 / HBM_PEAK_GBS) if kern > 0 else 0.0}}
        ib.close()
        del ib, grid, p, b
    except Exception as e:   # informational: never takes the headline down with it
        out["c3"] = {"error": repr(e)}
    # ---- config 4: 224 x 224 bonded DEM elements, 90 sub-steps per step ----
    try:
        if time.perf_counter() - t_start < budget_s:
            grid, p, b, bd = S.config_c4(nx=224, ny=224, hexagonal=False, radius=1500.0, ni=60, nj=60, gridres=20000.0, sub_steps=90,
                                         origin=(100137.0, 100211.0), bump=(900.0e3, 440.0e3))
            ib = Icebergs(grid, p, capacity=len(b["lon"]), device=torch.cuda.current_device())
            ib.upload_bergs(b); ib.upload_bonds(bd)
            ib.run(2); ib.sync()
            steps = 20
            t0 = time.perf_counter(); ib.run(steps); ib.sync(); dt = time.perf_counter() - t0
            acc, outp, scal = ib.fetch()
            ne = len(b["lon"])
            out["c4"] = {"workload": "BASELINE configs[3]: %d square-packed DEM elements (%d bond sides), MTS with 90 explicit sub-steps, dt=1800 s" % (ne, int(bd["count"].sum())),
                         "ms_per_step": 1e3 * dt / steps, "element_substeps_per_s": ne * p.mts_sub_steps * steps / dt, "berg_steps_per_s": ne * steps / dt,
                         "steps": steps, "error_count": float(scal[T.SCALAR_NAMES["error_count"]]),
                         "algorithmic_bytes_per_element_substep": 256 + 104 * 4,
                         "achieved_GBps": (256 + 104 * 4) * ne * p.mts_sub_steps * steps / dt / 1e9}
            ib.close()
        else:
            out["c4"] = {"skipped": "time budget"}
    except Exception as e:
        out["c4"] = {"error": repr(e)}
    out["seconds"] = time.perf_counter() - t_start
    return out


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-worker":
        return _cpu_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--spinup", type=int, default=24, help="untimed steps before the warm-up: the synthetic bergs start at rest and take ~20 steps "
                    "to reach their drift (per-launch time of the hot build over those steps at 1e7 bergs: 0.81, 0.89, 0.98, 1.06, 1.02, 0.96 ... 0.88, then 0.82 flat); 0 = start timing in the transient")
    ap.add_argument("--bergs", type=int, default=10_000_000, help="bergs per GPU (BASELINE target: >= 1e7 bergs stepped; configs[4]: 1e7 per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="N>1: keep the all-reduce and the gather on the critical path")
    ap.add_argument("--split-general", action="store_true", help="experiment: hot build in two halves, general build on the side stream")
    ap.add_argument("--force-collective", action="store_true", help="rehearsal: run the N>1 code path (RCCL all-reduce) with one rank")
    ap.add_argument("--resort", type=int, default=16, help="re-bin the bergs by cell every this many steps (move_berg_between_cells)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="N>1 rehearsal without N GPUs: every rank uses GPU 0 and the exchange goes through gloo (what is exercised is the N>1 code path, not its speed)")
    ap.add_argument("--advance-clock", action="store_true", help="kid_set_params with an advancing current_yearday before every step, as a model run does")
    ap.add_argument("--slow-lane", action="store_true", help="step the bergs the hot build hands over on a second stream for two steps (round 2's default: it paid while the "
                    "hot build left room for other kernels on a SIMD; with three waves of 160 registers it does not: 1.09 against 1.05 ms/step at 1e7 bergs)")
    ap.add_argument("--no-slow-lane", action="store_true", help="(default since round 3) keep the general build between two hot builds")
    ap.add_argument("--no-other-configs", action="store_true", help="headline only: skip the informational runs after the timed region (store-on figure, configs 3 and 4)")
    ap.add_argument("--cpu-bergs", type=int, default=500_000)
    ap.add_argument("--cpu-steps", type=int, default=8)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # CPU baseline first, in GPU-free child processes, before this process touches the GPU (rank 0 at N=1 only)
    cpu_line = cpu_baseline(args.cpu_bergs, args.cpu_steps) if (world == 1 and rank == 0 and not args.no_cpu_baseline) else None

    import numpy as np
    import torch
    from icebergs_amd import synthetic as S, types as T
    from icebergs_amd.framework import Icebergs

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
    slow_lane = args.slow_lane and not args.no_slow_lane and not args.split_general and not args.no_pipeline
    pipelined = (multi and not args.no_pipeline) or args.split_general or slow_lane
    if pipelined:
        # the all-reduce (N>1) and the gather of step k run on a second stream under the per-berg kernels of step k+1;
        # slow lane: so do the general-build launches (bergs that crossed a cell edge or bounced)
        stepper = PipelinedStepper(ib, params, dist, force_collective=args.force_collective, split_general=args.split_general, slow_lane=slow_lane, resort_interval=args.resort)
        nreduced = stepper.views[0][0].numel() - T.NSCALAR
    else:
        acc_t = torch.zeros(count, dtype=torch.float64, device=dev)
        ib.bind_accum_buffer(acc_t.data_ptr(), count)
        stepper = ShardedStepper(ib, acc_t, ib.ncell, params.diag_mask, dist, params=params, force_collective=args.force_collective, resort_interval=args.resort)
        nreduced = stepper.planes.numel() - T.NSCALAR

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

    for _ in range(args.spinup + args.warmup):
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
    elapsed_min = elapsed
    if dist is not None:
        tt = torch.tensor([elapsed, -elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, elapsed_min = float(tt[0].item()), -float(tt[1].item())
    n_slots, n_alive = ib.num_bergs()

    if rank == 0:
        total_bergs = args.bergs * world
        value = total_bergs * args.steps / elapsed
        # the dominant kernel is the hot build of berg_kernel: one launch per step, or two (one per half-population)
        # when the pipelined schedule is on; every berg goes through exactly one of them per step
        kern_ms = berg_ms / max(launches, 1)
        bergs_per_launch = args.bergs * args.steps / max(launches, 1)
        achieved = ALGO_BYTES_PER_BERG_STEP * bergs_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        # HBM bytes and VALU instructions per launch of the dominant kernel: PMC counters cannot be read from inside the
        # process, so these come from the committed rocprofv3 --pmc passes of this same command (profiles/r02_pmc_summary.json,
        # tools/profiling/run_pmc.sh) and are reported only for the population those passes were taken at
        traffic, traffic_src, valu_per_berg_step, pmc, pmc_note = None, None, None, None, None
        try:
            with open(PMC_SUMMARY) as f:
                summary = json.load(f)
            pmc = summary["hot"]
            libv = ib.lib.kid_version().decode()
            if not summary.get("complete", False):
                pmc_note = "counter passes incomplete in " + os.path.basename(PMC_SUMMARY)
            elif summary.get("library") != libv:
                pmc_note = "counters were taken on another build (%s), this run is %s" % (summary.get("library"), libv)
            elif abs(pmc["grid_size_mean"] - bergs_per_launch) > 256:
                pmc_note = "counters were taken at another population"
            else:
                traffic = pmc["hbm"]["traffic_bytes_per_launch"]
                traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE in separate passes; gfx950 fetch correction x2)" % os.path.basename(PMC_SUMMARY)
                valu_per_berg_step = pmc["valu_wave_instr_per_berg_step"]
        except (OSError, KeyError, ValueError) as e:
            pmc_note = "no PMC summary (%r)" % (e,)
        hbm_frac = achieved / HBM_PEAK_GBS
        # the other bound: one wave-instruction (fp64 included) per 4 cycles per SIMD.  instructions per berg-step x waves / time
        valu = None
        if valu_per_berg_step and kern_ms > 0:
            issue_s = valu_per_berg_step * (bergs_per_launch / 64.0) * 4.0 / (N_SIMD * SIMD_CLOCK_HZ)
            valu = {"wave_instr_per_berg_step": valu_per_berg_step, "floor_ms": 1e3 * issue_s, "frac": issue_s / (kern_ms * 1e-3),
                    "peak": "1 wave-instruction / 4 cycles / SIMD, %d SIMDs at %.1f GHz" % (N_SIMD, SIMD_CLOCK_HZ / 1e9),
                    "source": "SQ_INSTS_VALU per launch / waves (profiles/%s)" % os.path.basename(PMC_SUMMARY)}
            clk = pmc.get("shader_clock_ghz")   # GRBM_GUI_ACTIVE / 8 XCDs / the launch's duration in the same PMC pass
            if clk:
                valu["shader_clock_ghz_measured"] = clk
                valu["frac_at_measured_clock"] = valu["frac"] * (SIMD_CLOCK_HZ / 1e9) / clk
        binding = "valu_fp64_issue" if (valu and valu["frac"] > hbm_frac) else "hbm"
        line = {
            "metric": "berg_steps_per_sec", "value": value, "unit": "berg-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "spinup_steps": args.spinup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4] per-GPU share with configs[1] physics: %d synthetic bergs/GPU (random mass classes), 360x200 lat-lon ocean grid, "
                                   "RK4 drag+Coriolis+melt+mass spreading, dt=1800 s, ignore_traj=T" % args.bergs,
                       "bergs_per_gpu": args.bergs, "grid": "360x200", "sharding": "particle index, replicated grid",
                       "exchange": ("one RCCL all-reduce of %d per-cell planes (%.1f MB) + %d scalars per step%s" % (nreduced // ib.ncell, nreduced * 8 / 1e6, T.NSCALAR, ", overlapped with the next step's kernels" if pipelined else "")) if multi else "none (1 GPU)",
                       "planes_reduced_per_step": (nreduced // ib.ncell) if multi else 0, "MB_reduced_per_step": (nreduced * 8 / 1e6) if multi else 0.0,
                       "bergs_alive_at_end": n_alive},
            "library": ib.lib.kid_version().decode(),   # names the build switches (an experiment or exact-math build says so)
            "per_gpu_value": value / world, "host_submit_ms_per_step": 1e3 * t_submit / args.steps,
            "ms_per_step_rank_max": 1e3 * elapsed / args.steps, "ms_per_step_rank_min": 1e3 * elapsed_min / args.steps,
            # `bound` = the bound that binds the dominant kernel.  achieved / peak / frac are the HBM figures (algorithmic bytes over
            # the kernel's launch time, the quantity BASELINE.json's target is stated in); `valu_fp64_issue` carries the other one.
            "roofline": {"bound": binding, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac,
                         "frac_of": "hbm", "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_frac},
                         "valu_fp64_issue": valu,
                         "traffic": traffic, "traffic_source": traffic_src, "traffic_note": pmc_note,
                         "kernel": "berg_kernel<true, true, 14u, true, 1> (RK4, old interp order, evolve|thermo|spread, hot build, plain namelist)",
                         "kernel_ms_avg": kern_ms, "kernel_launches": launches, "bergs_per_launch": bergs_per_launch,
                         "algorithmic_bytes_per_berg_step": ALGO_BYTES_PER_BERG_STEP},
        }
        # The headline runs the way config 2 is configured (ignore_traj=T: berg%uo..hi are not stored back, IB:2890-2894 has no
        # reader).  A host that samples trajectories, prints bergs_chksum or migrates bergs needs those 11 members: the same
        # population with the store on (the plain hot build again, K = 3: the eleven members written right after the
        # thermodynamics' interpolation) -- informational, not `value`.
        line["roofline"]["store_environment"] = False
        if world == 1 and not args.force_collective and not args.no_other_configs:
            ib.set_store_environment(True)
            for _ in range(2):
                step()
            fence()
            ib.profile(True)
            t1 = time.perf_counter()
            nst = max(2, min(args.steps, 10))
            for _ in range(nst):
                step()
            fence()
            el1 = time.perf_counter() - t1
            bms, bl, _ = ib.profile_get()
            ib.profile(False)
            ib.set_store_environment(False)
            k_ms = bms / max(bl, 1)
            per_launch = args.bergs * nst / max(bl, 1)
            line["roofline"]["with_store_environment"] = {
                "ms_per_step": 1e3 * el1 / nst, "value": args.bergs * nst / el1, "kernel_ms_avg": k_ms, "steps": nst,
                "algorithmic_bytes_per_berg_step": ALGO_BYTES_PER_BERG_STEP + 11 * 8,
                "frac": ((ALGO_BYTES_PER_BERG_STEP + 11 * 8) * per_launch / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_ms > 0 else None,
                "note": "berg%uo,vo,ui,vi,ua,va,ssh_x,ssh_y,sst,cn,hi stored every step (IB:2890-2894): +88 B per berg-step; berg_kernel<true, true, 14u, true, 3>, the plain hot build that stores them"}
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line
        if world == 1 and not args.no_other_configs and not args.force_collective:
            stepper = None
            ib.close()   # (3.9 GB of berg state + the re-binning buffers: config 3 needs the room to be quick to allocate)
            torch.cuda.set_stream(torch.cuda.default_stream(dev))
            line["other_configs"] = other_configs(np, torch, S, T, Icebergs)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ib.close()


if __name__ == "__main__":
    main()

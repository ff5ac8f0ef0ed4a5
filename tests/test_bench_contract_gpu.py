"""bench.py prints ONE JSON line with the keys the driver reads (a small population here; the default run is the real one)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--bergs", "50000",
                        "--cpu-bergs", "2000", "--cpu-steps", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "berg_steps_per_sec" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["dtype"] == "f64"
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 50000 * 6 / (d["ms_per_step"] * 1e-3 * 6)) / d["value"] < 1e-9
    rf = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] in ("hbm", "valu_fp64_issue") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["frac_of"] == "hbm" and abs(rf["hbm"]["frac"] - rf["frac"]) < 1e-15
    assert rf["kernel_launches"] == 6 and rf["kernel_ms_avg"] > 0          # HIP events around every hot-build launch of the timed region
    cb = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["one_core_value"] > 0   # one core and all-core shards


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal():
    """the N>1 code path of bench.py (per-rank shards, the three-stream slow-lane stepper with an exchange, barrier + MAX over
    ranks, one line from rank 0) with two ranks on GPU 0 and gloo instead of RCCL: exercised, not timed"""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--bergs", "20000", "--rehearse-on-one-gpu"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stderr[-2000:], r.stdout[-500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["bergs_per_gpu"] == 20000
    assert abs(d["value"] - 2 * 20000 * 5 / (d["ms_per_step"] * 1e-3 * 5)) / d["value"] < 1e-9      # whole-job aggregate
    assert "all-reduce" in d["config"]["exchange"] and "cpu_baseline" not in d

#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc passes (directories given on the command line): one JSON object on stdout with, for
the hot and the general build of berg_kernel, every counter's mean per launch, the launch grid and the derived figures
DESIGN.md quotes (VALU wave-instructions per berg-step, stalled share of wave cycles, HBM bytes per launch with the
gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md)."""
import collections, csv, glob, json, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list)); grid = collections.defaultdict(list); names = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "berg_kernel" not in k: continue
            key = "hot" if (", true>" in k or ", true, 1>" in k or ", true, 0>" in k) else "general"
            names.setdefault(key, k)
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            grid[key].append(int(r["Grid_Size"]))
out = {}
for key, cs in agg.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    o = {"kernel": names[key], "launches_seen": max(len(v) for v in cs.values()), "grid_size_mean": sum(grid[key]) / len(grid[key]), "counters": m}
    n = o["grid_size_mean"]
    if "SQ_INSTS_VALU" in m: o["valu_wave_instr_per_berg_step"] = m["SQ_INSTS_VALU"] * 64.0 / n if key == "hot" else None
    if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_ANY" in m: o["wait_any_share_of_wave_cycles"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
    if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_INST_ANY" in m: o["wait_inst_share_of_wave_cycles"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
    if "SQ_BUSY_CYCLES" in m and "SQ_ACTIVE_INST_VALU" in m: o["valu_active_per_busy_cycle"] = m["SQ_ACTIVE_INST_VALU"] / m["SQ_BUSY_CYCLES"]
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        o["hbm"] = {"fetch_bytes_raw": m["FETCH_SIZE"] * 1024, "fetch_bytes_corrected": 2 * m["FETCH_SIZE"] * 1024, "write_bytes": m["WRITE_SIZE"] * 1024,
                    "traffic_bytes_per_launch": 2 * m["FETCH_SIZE"] * 1024 + m["WRITE_SIZE"] * 1024,
                    "correction": "FETCH_SIZE (KB) counts 128-B read requests at 64 B on gfx950 -> doubled; WRITE_SIZE (KB) as is (MI355X_MICROARCH.md)"}
    out[key] = o
# the shader clock during the hot build: GRBM_GUI_ACTIVE (summed over the 8 XCDs) per launch over the launch's duration in the
# kernel trace of the same pass
for d in sys.argv[1:]:
    durs = []
    for f in glob.glob(d + "/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "berg_kernel" in k and (", true>" in k or ", true, 1>" in k or ", true, 0>" in k) and int(r["Grid_Size_X"]) > 1000000:
                durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    has = any("GRBM_GUI_ACTIVE" in open(f).read(200000) for f in glob.glob(d + "/*/*counter_collection.csv"))
    if durs and has and "hot" in out and "GRBM_GUI_ACTIVE" in out["hot"]["counters"]:
        mean_ns = sum(durs) / len(durs)
        out["hot"]["kernel_ns_under_pmc"] = mean_ns
        out["hot"]["shader_clock_ghz"] = out["hot"]["counters"]["GRBM_GUI_ACTIVE"] / 8.0 / mean_ns
import ctypes, os
try:   # which build was measured (kid_version names the sources' hash): bench.py quotes these counters only for that build
    so = os.environ.get("KID_HIP_SO") or os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "icebergs_amd", "csrc", "libkid_hip.so")
    L = ctypes.CDLL(so); L.kid_version.restype = ctypes.c_char_p
    out["library"] = L.kid_version().decode()
except OSError as e:
    out["library"] = None
need = {"SQ_INSTS_VALU", "SQ_WAVES", "FETCH_SIZE", "WRITE_SIZE"}
out["complete"] = bool("hot" in out and need <= set(out["hot"]["counters"]))   # every pass produced its counters for the hot build
print(json.dumps(out, indent=1))

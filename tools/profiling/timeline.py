#!/usr/bin/env python3
"""Prints a steady-state window of a rocprofv3 --kernel-trace CSV of bench.py as a timeline:
start offset (us), duration (us), queue, kernel.  usage: timeline.py <kernel_trace.csv> [first_hot_launch] [n_hot_launches]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def is_hot(n):   # berg_kernel<RK, OLD_ORDER, PH, FAST, K>
    a = n.split("berg_kernel<")[1].split(">")[0].split(", ")
    return len(a) > 3 and a[3] == "true"


def short(n):
    if "berg_kernel" in n:
        tpl = n.split("berg_kernel")[1].split(">")[0]
        return "berg_kernel" + tpl + ">" + ("  (hot build)" if is_hot(n) else "  (general build)")
    for k in ("pack_forcing_kernel", "gather_kernel", "permute_all_kernel", "cell_rank_kernel", "cell_place_kernel", "translate_list_kernel",
              "set_berg_table_kernel", "set_params_kernel", "set_grid_kernel", "fillBufferAligned", "copyBuffer", "scan"):
        if k in n:
            return k
    return n.split("(")[0][-50:]


hot = [i for i, r in enumerate(rows) if "berg_kernel<" in r["Kernel_Name"] and is_hot(r["Kernel_Name"])]
first = min(int(sys.argv[2]) if len(sys.argv) > 2 else 8, max(len(hot) - 4, 0))
count = int(sys.argv[3]) if len(sys.argv) > 3 else 3
i0 = max(hot[first] - 1, 0)
t0 = int(rows[i0]["Start_Timestamp"])
print("%10s %9s  %-5s %s" % ("start_us", "dur_us", "queue", "kernel"))
for r in rows[i0:hot[first + count]]:
    print("%10.1f %9.1f  %-5s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                     r.get("Queue_Id", "?"), short(r["Kernel_Name"])))

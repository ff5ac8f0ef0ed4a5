#!/usr/bin/env python3
"""Prints the last few steps of a rocprofv3 --kernel-trace CSV as a timeline (start offset, duration, stream, kernel)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-nshow:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"]
    short = name.split("(")[0][-60:]
    if "berg_kernel" in name:
        short = "berg_kernel" + name.split("berg_kernel")[1].split("(")[0]
    print("%9.1f %8.1f  q=%s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                     r.get("Queue_Id", "?"), short))

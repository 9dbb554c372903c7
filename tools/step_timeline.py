#!/usr/bin/env python3
"""Kernel timeline of a few steps from a rocprofv3 --kernel-trace CSV: start (us, relative), duration, queue, name.
usage: step_timeline.py <dir with *kernel_trace.csv> [first_force_launch_index] [n_force_launches]"""
import csv
import glob
import sys

d = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 60
count = int(sys.argv[3]) if len(sys.argv) > 3 else 6
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
force = [i for i, r in enumerate(rows) if "k_force_lj_verlet" in r["Kernel_Name"]]
a, b = force[first], force[min(first + count, len(force) - 1)]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3s}  {name}")
    prev_end = max(prev_end, e)

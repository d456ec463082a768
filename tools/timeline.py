#!/usr/bin/env python
"""Print the kernel timeline of the last N steps of a rocprofv3 --kernel-trace csv (start / end in us relative to the
step's batch_gather launch, queue / stream id), to see what overlaps what.  usage: timeline.py DIR [steps]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getsize)
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?"),
                 r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Workgroup_Size_X", "?")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "batch_gather_kernel" in r[2] or "refresh_assemble" in r[2]]
if sum(1 for r in rows if "step_ahead" in r[2]) > len(marks):  # step-ahead form: a step starts at its gather
    marks = [i for i, r in enumerate(rows) if "embed_fwd_kernel" in r[2]]
lo = marks[-(n + 1)]
t0 = rows[lo][0]
for st, en, name, q, s, gx, gy, wx in rows[lo:marks[-1] + 1]:
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("rechub::", "").split("(")[0][:44]
    print(f"{(st - t0) / 1e3:9.1f} {(en - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f}  q{q} s{s}  {short:44s} grid {gx}x{gy}/{wx}")

#!/usr/bin/env python
"""Per-step statistics over a rocprofv3 --kernel-trace csv of `bench.py --trace-inner`: the step period (refresh_assemble /
batch_gather launch to the next one) and the duration of every kernel of the step, as median / mean / max over the steady-state
steps, plus the steps whose period is more than 15 % above the median.  usage: step_stats.py DIR [last_n_steps]"""
import csv
import glob
import os
import statistics as st
import sys

d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 100
f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getsize)
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"))
              for r in csv.DictReader(open(f)))


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").replace("rechub::", "").split("(")[0][:52]


marks = [i for i, r in enumerate(rows) if "batch_gather_kernel" in r[2] or "refresh_assemble" in r[2]]
if sum(1 for r in rows if "step_ahead" in r[2]) > len(marks):  # step-ahead form: a step starts at its gather
    marks = [i for i, r in enumerate(rows) if "embed_fwd_kernel" in r[2]]
marks = marks[-(last + 1):]
per, dur, start = [], {}, {}
for a, b in zip(marks[:-1], marks[1:]):
    t0 = rows[a][0]
    per.append((rows[b][0] - t0) / 1e3)
    seen = {}
    for s, e, n, q in rows[a:b]:
        k = short(n)
        seen[k] = seen.get(k, 0) + 1
        key = k if seen[k] == 1 else f"{k} #{seen[k]}"
        dur.setdefault(key, []).append((e - s) / 1e3)
        start.setdefault(key, []).append((s - t0) / 1e3)
med = st.median(per)
print(f"steps {len(per)}  period us: median {med:.1f} mean {st.mean(per):.1f} min {min(per):.1f} max {max(per):.1f}")
print(f"{'kernel':56s} {'n':>4s} {'start':>8s} {'median':>8s} {'mean':>8s} {'max':>8s}")
for k in sorted(dur, key=lambda k: st.median(start[k])):
    v = dur[k]
    print(f"{k:56s} {len(v):4d} {st.median(start[k]):8.1f} {st.median(v):8.1f} {st.mean(v):8.1f} {max(v):8.1f}")
if os.environ.get("STEP_DETAIL"):  # per step: start of the side-stream sweep and start / duration of the kernels named in STEP_DETAIL
    keys = [k for k in dur if any(w in k for w in os.environ["STEP_DETAIL"].split(";"))]
    print("step period " + " | ".join(f"{k[:34]:34s}" for k in keys))
    for i in range(len(per)):
        print(f"{i:4d} {per[i]:6.1f} " + " | ".join(f"start {start[k][i]:7.1f} dur {dur[k][i]:7.1f}       " if i < len(dur[k]) else " " * 34
                                                   for k in keys))
slow = [i for i, p in enumerate(per) if p > 1.15 * med]
print(f"slow steps (> 1.15 x median): {len(slow)} of {len(per)}: " + " ".join(f"{i}:{per[i]:.0f}" for i in slow[:40]))

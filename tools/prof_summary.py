#!/usr/bin/env python
"""Summarise a rocprofv3 output directory (csv format): per-kernel count / avg / min / total from the kernel trace,
or per-kernel mean counter value with --pmc NAME.  Only our rh_* kernels and the top ATen / hipBLASLt ones are listed."""
import argparse
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    if "adam_lazy_sweep_kernel" in name:
        # <LPR, MERGED>: false = the window sweep alone (in round 3: the DEFERRED side-stream sweep), true = the merged
        # end-of-step launch (touched rows + dense tables / window)
        tmpl = name[name.index("adam_lazy_sweep_kernel"):].split("(")[0]
        return "rechub::" + tmpl
    for key in ("embed_fwd_kernel", "embed_bwd_kernel", "adam_dense_kernel", "adam_lazy_sweep_kernel",
                "adam_lazy_touched_kernel", "adam_prepare_kernel", "batch_gather_kernel", "batch_advance_kernel",
                "cross_fwd_kernel", "cross_bwd_kernel", "seq_pool_kernel", "fm_fwd_kernel", "fm_bwd_kernel"):
        if key in name:
            return "rechub::" + key
    return name[:100]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--pmc", default=None)
    ap.add_argument("--tail", type=int, default=0, help="with --pmc: average only the LAST n dispatches of each kernel")
    a = ap.parse_args()
    if a.pmc:
        files = glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            sys.exit("no counter_collection.csv under " + a.dir)
        vals = defaultdict(list)
        for f in files:
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") != a.pmc:
                    continue
                vals[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
        acc = {}
        for k, v in vals.items():
            if a.tail:
                v = v[-a.tail:]
            acc[k] = [len(v), sum(v)]
        print(f"# rocprofv3 --pmc {a.pmc}: mean counter value per dispatch (raw counter units)")
        print(f"{'kernel':60s} {'dispatches':>10s} {'mean':>16s}")
        for k, (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
            print(f"{k:60s} {n:10d} {tot / n:16.1f}")
        return
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no kernel_trace.csv under " + a.dir)
    acc = defaultdict(list)
    by_shape = defaultdict(list)  # the gather kernels per launch shape = per batch size (bench.py's gather_kernel_sweep)
    for f in files:
        for row in csv.DictReader(open(f)):
            dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            acc[short(row["Kernel_Name"])].append(dur)
            if "embed_fwd" in row["Kernel_Name"] or "embed_bwd" in row["Kernel_Name"]:
                name = "embed_fwd_uniform" if "uniform" in row["Kernel_Name"] else short(row["Kernel_Name"]).split("::")[-1]
                by_shape[(name, int(row["Grid_Size_X"]) // max(1, int(row["Workgroup_Size_X"])))].append(dur)
    total = sum(sum(v) for v in acc.values())
    print("# rocprofv3 --kernel-trace --stats summary (durations in us).  median_us = the steady-state launch: the mean also")
    print("# holds the short first sweeps after a flush, the 6 ms flush itself and the B = 16384 / 65536 launches of bench.py's")
    print("# gather-kernel sweep")
    print(f"{'kernel':100s} {'calls':>7s} {'avg_us':>10s} {'median_us':>10s} {'min_us':>10s} {'total_ms':>10s} {'pct':>6s}")
    tail = []
    for (name, blocks), v in sorted(by_shape.items()):
        if len(v) >= 8:
            med = sorted(v)[len(v) // 2]
            tail.append(f"#   {name:22s} workgroups {blocks:7d}  launches {len(v):5d}  avg {sum(v) / len(v) / 1e3:8.2f} us  median {med / 1e3:8.2f} us")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:45]:
        med = sorted(v)[len(v) // 2]
        print(f"{k:100s} {len(v):7d} {sum(v) / len(v) / 1e3:10.2f} {med / 1e3:10.2f} {min(v) / 1e3:10.2f} {sum(v) / 1e6:10.3f} "
              f"{100.0 * sum(v) / total:6.2f}")
    for k, v in acc.items():
        if ("adam_lazy_sweep_kernel<4, false>" in k or "adam_lazy_sweep_wide_kernel" in k) and len(v) >= 40:
            # after a flush the first lazy_k sweeps replay 1, 2, ... steps (a ramp of lazy_k launches), the eager passes at the end
            # of bench.py run theirs without a chain beside them: the steady-state launch is the plateau in between
            p75 = sorted(v)[int(0.75 * len(v))]
            pl = [x for x in v if abs(x - p75) <= 0.1 * p75]
            print(f"# {k}: steady-state launches (within 10 % of the 75th percentile, {p75 / 1e3:.1f} us): {len(pl)} of {len(v)}, "
                  f"mean {sum(pl) / len(pl) / 1e3:.2f} us  <- compare with roofline.avg_launch_ms of the same run")
    if tail:
        print("# gather kernels per launch shape (forward: lane-split kernel 16 samples / workgroup, field-uniform kernel 16 samples /")
        print("# workgroup of 4 wavefronts; backward: ceil8(B / 256) x 26 workgroups):")
        print("\n".join(tail))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Per-variant FETCH_SIZE / WRITE_SIZE of `tools/bwd_ceiling_probe.py --pmc` from two rocprofv3 --pmc passes.
usage: bwd_ceiling_pmc.py DIR_FETCH DIR_WRITE SETS      (dispatch order per batch: fwd x (2 + SETS), then product, plain-wide,
plain-16B, no-sink x (2 + SETS) each; the two warm-up launches of a variant are dropped)"""
import csv
import glob
import os
import sys


def per_dispatch(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = {}
    for path in f:
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") != counter:
                continue
            k = (int(r["Dispatch_Id"]), r["Kernel_Name"])
            rows[k] = rows.get(k, 0.0) + float(r["Counter_Value"])
    return [(n, v) for (_, n), v in sorted(rows.items())]


fetch, write, sets = per_dispatch(sys.argv[1], "FETCH_SIZE"), per_dispatch(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
print("# KiB per launch (raw counter units; MI355X_MICROARCH.md: FETCH_SIZE counts wide coalesced reads at one half -> 2 x FETCH + WRITE")
print("# is the figure bench.py's roofline.traffic uses).  Float atomics are memory-side read-modify-writes: whether and how the")
print("# counters see them is not calibrated -- compare `product` with `plain-wide` (the same requests as plain stores).")
for what, key in (("embed_fwd", "embed_fwd"), ("embed_bwd", "embed_bwd_kernel")):
    fe = [v for n, v in fetch if key in n]
    wr = [v for n, v in write if key in n]
    names = ["fwd"] if what == "embed_fwd" else ["product", "plain-wide", "plain-16B", "no-sink"]
    per = 2 + sets
    # the probe runs its batches in order; report the LAST batch (the largest)
    fe, wr = fe[-per * len(names):], wr[-per * len(names):]
    for i, nm in enumerate(names):
        a = fe[i * per + 2:(i + 1) * per]
        b = wr[i * per + 2:(i + 1) * per]
        if not a or not b:
            continue
        fa, wb = sum(a) / len(a), sum(b) / len(b)
        print(f"{nm:10s} FETCH_SIZE {fa:12.1f}  WRITE_SIZE {wb:12.1f}  2*FETCH+WRITE {(2 * fa + wb) * 1024 / 1e6:9.1f} MB")

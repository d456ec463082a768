#!/usr/bin/env python
"""When does a captured side branch START?  (round 6; DCN-v2 / DIN timelines: in the BACKWARD the side branch of
ops.run_beside began 170-250 us after the node it depends on, in the forward at once.)

    rocprofv3 --kernel-trace --output-format csv -d /tmp/bo -o t -- python tools/probe/branch_order_probe.py
    python tools/probe/branch_order_probe.py --report /tmp/bo

One graph per case: k0 on the origin stream O, a fork (event), NM kernels "main" and one kernel "s1" whose only dependency is k0,
a join and a marker kernel on O.  The kernels are in-place scalings of tensors of different sizes (the report tells them apart by
their grid): k0 8 Mi floats, main 48 Mi each, s1 24 Mi, join 4 Mi.
  side_first   s1 on a side stream S, captured BEFORE main (on O)     -- the forward of run_beside as rounds 3-5 had it
  main_first   main on O captured first, then s1 on S                 -- its backward: autograd runs the later-created branch first
  both_side    main on a second side stream S2 captured first, then s1 on S; O only forks and joins
"""
import argparse
import csv
import glob
import os

NM = 4
SIZES = {"k0": 8 << 20, "main": 48 << 20, "s1": 24 << 20, "join": 4 << 20}


def run():
    import torch
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    buf = {k: torch.ones(n, device=dev) for k, n in SIZES.items()}
    mains = [torch.ones(SIZES["main"], device=dev) for _ in range(NM)]
    S, S2, O = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for case in ("side_first", "main_first", "both_side"):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(O):
            g.capture_begin(capture_error_mode="thread_local")
            buf["k0"].mul_(1.0001)
            S.wait_stream(O)
            S2.wait_stream(O)

            def main_part(stream):
                with torch.cuda.stream(stream):
                    for m in mains:
                        m.mul_(1.0001)

            def side_part():
                with torch.cuda.stream(S):
                    buf["s1"].mul_(1.0001)

            if case == "side_first":
                side_part()
                main_part(O)
            elif case == "main_first":
                main_part(O)
                side_part()
            else:
                main_part(S2)
                side_part()
            O.wait_stream(S)
            O.wait_stream(S2)
            buf["join"].mul_(1.0001)
            g.capture_end()
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()


def report(d):
    f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getsize)
    by_grid = {}
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), int(r.get("Grid_Size_X", 0)) ))
    rows.sort()
    grids = sorted({g for _, _, _, g in rows})
    # elementwise kernels: grid proportional to the element count
    unit = min(g for g in grids if g > 0)
    names = {}
    for k, n in SIZES.items():
        names[n // (4 << 20)] = k
    joins = [i for i, r in enumerate(rows) if r[3] == unit]
    case_names = ["side_first"] * 3 + ["main_first"] * 3 + ["both_side"] * 3
    start = 0
    for ci, j in enumerate(joins[-9:]):
        seg = rows[(joins[-9:][ci - 1] + 1) if ci else max(0, j - NM - 2):j + 1]
        t0 = seg[0][0]
        print(f"--- {case_names[ci]} replay {ci % 3 + 1}")
        for st, en, q, g in seg:
            print(f"{(st - t0) / 1e3:9.1f} {(en - t0) / 1e3:9.1f} {(en - st) / 1e3:7.1f}  q{q}  {names.get(g // unit, g)}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--report")
    a = ap.parse_args()
    report(a.report) if a.report else run()

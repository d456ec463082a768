#!/usr/bin/env python
"""Does the BACKWARD of ops.run_beside fork at once?  (round 6; the side branch's backward began 170-250 us late in the DCN-v2 /
DIN timelines.)  Two branches of NOPS element-wise ops on tensors of different sizes (main 32 Mi floats, side 16 Mi: the trace tells
them apart by grid), forward + backward captured into one hipGraph.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/bf -o t -- python tools/probe/backward_fork_probe.py [--mode M]
    python tools/timeline-like report: --report /tmp/bf
"""
import argparse
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
NOPS = 4


def run(mode):
    import torch
    from torch_rechub_amd import graphs, ops
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    xm = torch.ones(32 << 20, device=dev, requires_grad=True)
    xs = torch.ones(16 << 20, device=dev, requires_grad=True)

    def branch(x):
        y = x
        for _ in range(NOPS):
            y = y * 1.0001
        return y

    def step():
        xm.grad = xs.grad = None
        a, b = ops.run_beside(lambda: branch(xm), lambda: branch(xs), side_inputs=(xs,))
        loss = a.sum() + b.sum()
        loss.backward()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    g = graphs.SegmentedGraph()
    g.capture(step)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()


def report(d):
    f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getsize)
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), int(r.get("Grid_Size_X", 0)),
                   r["Kernel_Name"][:60]) for r in csv.DictReader(open(f)))
    rows = rows[-(4 * NOPS + 12):]
    t0 = rows[0][0]
    for st, en, q, g, n in rows:
        print(f"{(st - t0) / 1e3:9.1f} {(en - t0) / 1e3:9.1f} {(en - st) / 1e3:7.1f}  q{q}  grid {g:10d}  {n}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--report")
    ap.add_argument("--mode", default="")
    a = ap.parse_args()
    report(a.report) if a.report else run(a.mode)

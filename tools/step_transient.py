#!/usr/bin/env python
"""Per-step GPU time of the first steps after a device synchronisation (HIP events after every replay): where does the
fixed ~0.2 ms of a short timed region go?   python tools/step_transient.py [--form deferred|inline]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--form", default="deferred")
    a = ap.parse_args()
    os.environ["RECHUB_STEP_FORM"] = a.form
    sys.argv = ["bench.py", "--rows", "4000000", "--no-cpu-baseline", "--brief"]
    args = bench.parse()
    dev = torch.device("cuda:0")
    wl = bench.Workload(args, dev, 0)
    model, trainer, loader = wl.build(None, True, batch=args.batch)
    for _ in range(120):
        trainer._graphed_step(loader)
    for trial in range(3):
        torch.cuda.synchronize()
        time.sleep(0.01 * trial)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(25)]
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(24):
            trainer._graphed_step(loader)
            ev[i + 1].record()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(24)]
        print(f"{a.form} trial {trial}: wall {dt * 1e3:.3f} ms, enqueue {t_enq * 1e3:.3f} ms, per-step us:", " ".join(f"{s:.0f}" for s in steps))


main()

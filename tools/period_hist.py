#!/usr/bin/env python
"""Distribution of the step period WITHOUT a profiler: the headline workload in its steady state (as bench.py --trace-inner),
one timing event recorded in front of every replay (each costs the chain a few us -- the figures are for comparing forms and
for spotting slow steps, not the headline).  usage: period_hist.py [bench.py arguments] ; prints percentiles in us."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    args = bench.parse()
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    wl = bench.Workload(args, device, 0)
    model, trainer, loader = wl.build(None, True, batch=args.batch)
    trainer._graphed_step(loader)
    warm = max(args.warmup, args.lazy_k + 8) + len(trainer.TUNE_CANDIDATES) * (trainer.TUNE_SETTLE + trainer.TUNE_STEPS) + 4
    for _ in range(warm):
        trainer._graphed_step(loader)
    n = args.steps
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    for i in range(n):
        evs[i].record()
        trainer._graphed_step(loader)
    evs[n].record()
    torch.cuda.synchronize()
    per = sorted(1e3 * evs[i].elapsed_time(evs[i + 1]) for i in range(n))
    q = lambda f: per[min(n - 1, int(f * n))]
    print(f"form {getattr(trainer, '_tune', {}).get('chosen')} steps {n}  period us: mean {sum(per) / n:.1f} min {per[0]:.1f} p10 {q(.1):.1f} "
          f"p50 {q(.5):.1f} p90 {q(.9):.1f} p99 {q(.99):.1f} max {per[-1]:.1f}  slow(>1.15 x p50) {sum(p > 1.15 * q(.5) for p in per)}")


if __name__ == "__main__":
    main()

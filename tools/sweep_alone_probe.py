#!/usr/bin/env python
"""The deferred window sweep ALONE (no step beside it) at several persistent grid sizes: is it starved for instruction-level
parallelism at 2 wavefronts per SIMD (512 workgroups), or is the 235 us it takes beside the step contention?
Every sweep replays K = 64 zero-gradient steps for the rows of its window (steady state, nothing touched).
    python tools/sweep_alone_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CRITEO_VOCABS  # noqa: E402


def main():
    from torch_rechub_amd import _lib, ops
    from torch_rechub_amd.optim import SWEEP_LAZY_TABLES, TableAdam
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    tables = [torch.nn.Parameter(torch.randn(v, 16, device=dev, generator=g) * 1e-2) for v in CRITEO_VOCABS]
    opt = TableAdam(tables, table_params=tables, lr=1e-3, weight_decay=1e-5, lazy_k=64)
    opt.sync_hyper()
    opt._lazy_setup()
    t = 0

    def one(timed=None):
        nonlocal t
        _lib.call("rh_adam_prepare", ops._p(opt._t_hyper), ops._p(opt._t_step), ops._p(opt._t_ring), opt.RING, ops._stream())
        t += 1
        if timed is not None:
            timed[0].record()
        opt._sweep(SWEEP_LAZY_TABLES, ops._stream(), t_value=t)
        if timed is not None:
            timed[1].record()

    for _ in range(72):
        one()
    torch.cuda.synchronize()
    for grid in (256, 384, 512, 768, 1024, 2048, 8192):
        _lib.call("rh_set_tuning", 8, grid)
        for _ in range(4):
            one()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for e in evs:
            one(e)
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)
        print(f"deferred sweep alone, {grid:5d} workgroups: median {ms[len(ms) // 2] * 1e3:7.1f} us  (min {ms[0] * 1e3:.1f})", flush=True)


if __name__ == "__main__":
    main()

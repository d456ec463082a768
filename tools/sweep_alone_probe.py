#!/usr/bin/env python
"""The deferred window sweep ALONE (no step beside it) at several persistent grid sizes: is it starved for instruction-level
parallelism at 2 wavefronts per SIMD (512 workgroups), or is the 235 us it takes beside the step contention?
Every sweep replays K zero-gradient steps for the rows of its window (steady state, nothing touched).  Round 5: both forms of
the kernel -- one float4 per lane (RH_TUNE_SWEEP_WIDE = 1, rounds 1-4) and two (the default) -- and the fraction of the
4.626 T element-steps/s VALU ceiling each reaches.
    python tools/sweep_alone_probe.py [lazy_k = 128]"""
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
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    opt = TableAdam(tables, table_params=tables, lr=1e-3, weight_decay=1e-5, lazy_k=K)
    lazy_elems = sum(p.numel() for p in tables if opt.table_k(p) != 1)
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

    for _ in range(K + 8):
        one()
    torch.cuda.synchronize()
    peak = 1024 * 2.4e9 / 136 * 256  # element-steps / s
    for wide in (1, 2):
        _lib.call("rh_set_tuning", 14, wide)
        for grid in (256, 512, 768, 1024, 2048, 8192):
            _lib.call("rh_set_tuning", 8, grid)
            for _ in range(4):
                one()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for e in evs:
                one(e)
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in evs)
            med = ms[len(ms) // 2]
            print(f"deferred sweep alone, K = {K}, {wide} float4 per lane, {grid:5d} workgroups: median {med * 1e3:7.1f} us  "
                  f"(min {ms[0] * 1e3:.1f})  = {lazy_elems / (med * 1e-3) / peak:.3f} of the VALU ceiling", flush=True)


if __name__ == "__main__":
    main()

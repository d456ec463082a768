#!/usr/bin/env python
"""Feasibility probe: does the VALU-bound lazy-Adam sweep overlap with the latency-bound fwd+bwd kernels when the two
run on different HIP streams?  A = hipGraph replay of DeepFM forward+backward (no optimizer), B = back-to-back sweeps of an
independent copy of the Criteo tables.  Prints A alone, B alone, A || B.
    python tools/overlap_probe.py [--prio]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CRITEO_VOCABS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prio", action="store_true", help="run A on a high-priority stream")
    ap.add_argument("--grid", type=int, default=0, help="sweep grid (rh_set_tuning key 2)")
    ap.add_argument("--pad", type=int, default=0, help="extra LDS bytes per sweep workgroup (rh_set_tuning key 3)")
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--cus", type=int, default=0, help="run B on a stream masked to this many CUs of every XCD")
    ap.add_argument("--cus-a", type=int, default=0, help="run A on a stream masked to the LAST n CUs of every XCD")
    a = ap.parse_args()
    from torch_rechub_amd import _lib, ops
    from torch_rechub_amd.basic.features import DenseFeature, SparseFeature
    from torch_rechub_amd.models.ranking import DeepFM
    from torch_rechub_amd.optim import TableAdam
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    B = 4096
    dense = [DenseFeature(f"I{i}") for i in range(13)]
    sparse = [SparseFeature(f"C{i}", v, 16) for i, v in enumerate(CRITEO_VOCABS)]
    with torch.device(dev):
        model = DeepFM(dense + sparse, sparse, {"dims": [256, 128], "dropout": 0.2, "activation": "relu"})
    model.train()
    x = {f.name: torch.randint(0, v, (B,), device=dev, generator=g) for f, v in zip(sparse, CRITEO_VOCABS)}
    x.update({f.name: torch.rand(B, device=dev, generator=g) for f in dense})
    y = (torch.rand(B, device=dev, generator=g) < 0.25).float()

    def fwd_bwd():
        loss = ops.bce_mean(model(x), y)
        loss.backward()

    sa = torch.cuda.Stream(priority=-1 if a.prio else 0)
    sb = torch.cuda.Stream()
    import ctypes
    if a.cus:
        ptr = ctypes.c_void_p()
        _lib.call("rh_stream_create_cumask", a.cus, 0, ctypes.byref(ptr))
        sb = torch.cuda.ExternalStream(ptr.value, device=dev)
    if a.cus_a:
        ptr = ctypes.c_void_p()
        _lib.call("rh_stream_create_cumask", a.cus_a, 1, ctypes.byref(ptr))
        sa = torch.cuda.ExternalStream(ptr.value, device=dev)
    with torch.cuda.stream(sa):
        for _ in range(3):
            fwd_bwd()
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=sa):
            fwd_bwd()
    tables = [torch.nn.Parameter(torch.randn(v, 16, device=dev, generator=g) * 1e-2) for v in CRITEO_VOCABS]
    lazy = TableAdam(tables, table_params=tables, lr=1e-3, weight_decay=1e-5, lazy_k=64)
    lazy.sync_hyper()
    if a.grid:
        _lib.call("rh_set_tuning", 2, a.grid)
    if a.pad:
        _lib.call("rh_set_tuning", 3, a.pad)
    with torch.cuda.stream(sb):
        for _ in range(40):
            lazy.step_tables()
    torch.cuda.synchronize()

    def run(do_a, do_b, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            if do_a:
                with torch.cuda.stream(sa):
                    gph.replay()
            if do_b:
                with torch.cuda.stream(sb):
                    lazy.step_tables()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    for _ in range(1):
        ta, tb, tab = run(True, False, a.iters), run(False, True, a.iters), run(True, True, a.iters)
        print(f"A (fwd+bwd graph) {ta:7.1f} us | B (sweep) {tb:7.1f} us | A||B {tab:7.1f} us  (sum {ta + tb:7.1f}, max {max(ta, tb):7.1f})",
              flush=True)


if __name__ == "__main__":
    main()

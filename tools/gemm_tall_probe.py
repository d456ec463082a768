#!/usr/bin/env python
"""The DIN attention MLP's tall products (M = B * L = 409600 rows): csrc/gemm.hip's tile kernel vs the library GEMM.
    python tools/gemm_tall_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    from torch_rechub_amd import _lib, ops
    dev = torch.device("cuda:0")
    M = int(os.environ.get("PROBE_M", "409600"))
    for N, K in ((256, 64), (128, 256), (64, 256), (256, 128)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.1
        b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        g = torch.randn(M, N, device=dev)
        gx = torch.empty(M, K, device=dev)
        t_lib = timeit(lambda: torch.nn.functional.linear(x, w, b))
        t_own = timeit(lambda: _lib.call("rh_linear_fwd", ops._p(x), K, ops._p(w), K, ops._p(b), M, N, K, ops._p(y), N,
                                         ops._p(None), ops._p(None), ops._p(None), ops._p(None), ops._stream()))
        ref = torch.nn.functional.linear(x, w, b)
        err = (y - ref).abs().max().item()
        t_dlib = timeit(lambda: torch.mm(g, w))
        t_down = timeit(lambda: _lib.call("rh_linear_dgrad", ops._p(g), N, ops._p(w), K, M, N, K, ops._p(gx), K, ops._stream()))
        derr = (gx - torch.mm(g, w)).abs().max().item()
        fl = 2.0 * M * N * K
        print(f"M={M} N={N} K={K}: fwd lib {t_lib:7.1f} us ({fl / t_lib / 1e6:5.1f} TF) own {t_own:7.1f} us ({fl / t_own / 1e6:5.1f} TF) "
              f"err {err:.1e} | dgrad lib {t_dlib:7.1f} us own {t_down:7.1f} us ({fl / t_down / 1e6:5.1f} TF) err {derr:.1e}", flush=True)


if __name__ == "__main__":
    main()

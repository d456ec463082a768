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


def att_l1():
    """The fused first attention layer (operand built in registers + statistics epilogue) against its three-launch form."""
    from torch_rechub_amd import _lib, ops
    dev = torch.device("cuda:0")
    B, L, D, N = 4096, 100, 16, 256
    hist, tgt = torch.randn(B, L, D, device=dev), torch.randn(B, D, device=dev)
    W, b = torch.randn(N, 4 * D, device=dev) * 0.1, torch.randn(N, device=dev)
    t_fused = timeit(lambda: ops.din_att_l1(hist, tgt, W, b, True))
    t_nostat = timeit(lambda: ops.din_att_l1(hist, tgt, W, b, False))

    def three():
        att = ops.din_att_input(hist, tgt)
        return torch.nn.functional.linear(att, W, b)

    t_three = timeit(three)
    print(f"att layer 1 (B={B}, L={L}, D={D}, N={N}): fused {t_fused:.1f} us (without statistics {t_nostat:.1f}), "
          f"operand kernel + library GEMM {t_three:.1f} us (+ the statistics pass over z)", flush=True)


if __name__ == "__main__":
    att_l1()
    main()

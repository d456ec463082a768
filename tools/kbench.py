#!/usr/bin/env python
"""Micro-benchmark of the hot kernels at the Criteo shape (real 2 GiB tables): per-launch time from HIP events over
back-to-back launches (includes the ~1.5 us inter-kernel gap), swept over batch size and tuning knobs.

    python tools/kbench.py [--what fwd,bwd,adam,cross,gather] [--iters 200]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import BWD_BYTES_PER_SAMPLE, CRITEO_VOCABS, FWD_BYTES_PER_SAMPLE  # noqa: E402


def timeit(fn, iters, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="fwd,bwd,adam,cross")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--batches", default="4096,16384,65536")
    args = ap.parse_args()
    what = set(args.what.split(","))
    from torch_rechub_amd import _lib, ops
    dev = torch.device("cuda:0")
    D, F = 16, len(CRITEO_VOCABS)
    g = torch.Generator(device=dev).manual_seed(1)
    tables = [torch.nn.Parameter(torch.randn(v, D, device=dev, generator=g) * 1e-2) for v in CRITEO_VOCABS]
    lr_w = torch.randn(1, F * D, device=dev)
    lr_b = torch.randn(1, device=dev)
    for B in [int(b) for b in args.batches.split(",")]:
        idx = torch.stack([torch.randint(0, v, (B,), device=dev, generator=g) for v in CRITEO_VOCABS], 1)
        dense = torch.rand(B, 13, device=dev, generator=g)
        cols = [idx[:, f] for f in range(F)]
        dcols = [dense[:, j] for j in range(13)]
        if "fwd" in what:
            for fs in (1, 2, 4, 8):
                call = ops.EmbedCall(tables, [None] * F, cols, dcols, want_fm=True, want_lr=True, field_split=fs)
                with torch.no_grad():
                    us = timeit(lambda: ops.fused_embedding(call, lr_w, lr_b), args.iters)
                gbs = FWD_BYTES_PER_SAMPLE * B / us / 1e3
                print(f"fwd  B={B:6d} fs={fs}  {us:8.2f} us  {gbs:7.0f} GB/s (alg)  [incl. 4 torch.empty]", flush=True)
            # raw launch (no torch allocations): reuse outputs
            call = ops.EmbedCall(tables, [None] * F, cols, dcols, want_fm=True, want_lr=True)
            out = torch.empty(B, F * D + 13, device=dev)
            fm = torch.empty(B, device=dev)
            lr = torch.empty(B, device=dev)
            ssum = torch.empty(B, D, device=dev)
            fdesc, idesc, ddesc = call.fdesc(False), call.idesc(), call.ddesc()
            for fs in (1, 2, 4, 8):
                def raw():
                    _lib.call("rh_embed_fwd", ops._p(fdesc), ops._p(idesc), 1, B, F, D, ops._p(ddesc), 13, F * D,
                              ops._p(out), out.stride(0), ops._p(lr_w), ops._p(lr_b), ops._p(lr), ops._p(fm),
                              ops._p(ssum), fs, ops._p(ops.err_flag(dev)), ops._stream())
                us = timeit(raw, args.iters)
                print(f"fwd* B={B:6d} fs={fs}  {us:8.2f} us  {FWD_BYTES_PER_SAMPLE * B / us / 1e3:7.0f} GB/s (alg)  [raw launch]",
                      flush=True)
        if "bwd" in what:
            call = ops.EmbedCall(tables, [None] * F, cols, dcols, want_fm=True, want_lr=True)
            out = torch.randn(B, F * D + 13, device=dev)
            ssum = out[:, :F * D].reshape(B, F, D).sum(1).contiguous()
            g_out = torch.randn(B, F * D + 13, device=dev)
            g_y = torch.randn(B, device=dev)
            fdesc, idesc = call.fdesc(True), call.idesc()
            for wide, spb in [(w, s) for w in (0, 1) for s in (128, 256, 512, 1024)]:
                _lib.call("rh_set_tuning", 1, wide)
                nch = _lib.call("rh_embed_bwd_nchunks", B, spb)
                partial = torch.empty(nch, F * D, device=dev)

                def raw():
                    _lib.call("rh_embed_bwd", ops._p(fdesc), ops._p(idesc), 1, B, F, D, ops._p(g_out), g_out.stride(0),
                              ops._p(out), out.stride(0), ops._p(ssum), ops._p(g_y), ops._p(g_y), ops._p(lr_w),
                              ops._p(partial), 1.0, 0, ops._p(None), spb, ops._p(ops.err_flag(dev)), ops._stream())
                us = timeit(raw, args.iters)
                print(f"bwd* B={B:6d} wide={wide} spb={spb:4d} {us:8.2f} us  "
                      f"{BWD_BYTES_PER_SAMPLE * B / us / 1e3:7.0f} GB/s (alg)", flush=True)
            for w in tables:
                ops.grad_buffer(w).zero_()
        if "cross" in what:
            d = 429
            x = torch.randn(B, d, device=dev)
            W = torch.randn(3, d, device=dev) / 20
            Bv = torch.randn(3, d, device=dev) / 10
            o = torch.empty(B, d, device=dev)
            us = timeit(lambda: _lib.call("rh_cross_fwd", ops._p(x), d, ops._p(x), d, ops._p(W), ops._p(Bv), B, d, 3,
                                          ops._p(o), d, ops._stream()), args.iters)
            print(f"cross fwd B={B:6d} {us:8.2f} us  {2 * B * d * 4 / us / 1e3:7.0f} GB/s", flush=True)
            nb = _lib.call("rh_cross_bwd_nblocks", B)
            part = torch.empty(nb, 2, 3, d, device=dev)
            gx = torch.empty(B, d, device=dev)
            us = timeit(lambda: _lib.call("rh_cross_bwd", ops._p(x), d, ops._p(x), d, ops._p(W), ops._p(Bv), B, d, 3,
                                          ops._p(o), d, ops._p(None), ops._p(gx), d, 1, ops._p(part), ops._stream()),
                        args.iters)
            print(f"cross bwd B={B:6d} {us:8.2f} us  {3 * B * d * 4 / us / 1e3:7.0f} GB/s", flush=True)
    if "mlp" in what:
        # DeepFM MLP pieces at B = 4096: library GEMM backward vs the split-batch MFMA weight gradient, and the head
        for (B, N, K) in [(4096, 256, 429), (4096, 128, 256), (16384, 256, 429)]:
            gg = torch.randn(B, N, device=dev)
            xx = torch.randn(B, K, device=dev)
            us_lib = timeit(lambda: (gg.t().mm(xx), gg.sum(0)), args.iters)
            us = timeit(lambda: ops.linear_wgrad(gg, xx), args.iters)
            fl = 2.0 * B * N * K
            print(f"wgrad B={B} N={N} K={K}: library mm+sum {us_lib:7.2f} us | rh_linear_wgrad {us:7.2f} us "
                  f"({fl / us / 1e6:6.1f} TFLOP/s f32 MFMA)", flush=True)
        B, K = 4096, 128
        h = torch.randn(B, K, device=dev, requires_grad=True)
        lin = torch.nn.Linear(K, 1).to(dev)
        e0 = torch.randn(B, 1, device=dev, requires_grad=True)
        e1 = torch.randn(B, 1, device=dev, requires_grad=True)
        t = (torch.rand(B, device=dev) < 0.3).float()

        def ref_head():
            y = torch.sigmoid((lin(h) + e0 + e1).squeeze(1))
            torch.nn.BCELoss()(y, t).backward()

        def my_head():
            y = ops.head_sigmoid(h, lin.weight, lin.bias, e0, e1)
            ops.bce_mean(y, t).backward()

        print(f"head+bce fwd+bwd B={B} K={K}: torch {timeit(ref_head, args.iters):7.2f} us | fused {timeit(my_head, args.iters):7.2f} us",
              flush=True)
    if "adam" in what:
        from torch_rechub_amd.optim import TableAdam
        opt = TableAdam(tables, table_params=tables, lr=1e-3, weight_decay=1e-5)
        opt.sync_hyper()
        n = sum(t.numel() for t in tables)
        us = timeit(opt.step_tables, 20, warm=3)
        print(f"adam dense {n} elems {us:9.1f} us  {28 * n / us / 1e3:7.0f} GB/s (alg 28 B/elem)", flush=True)
        for k in (16, 32, 64):
            for grid in (1024, 2048, 4096, 8192):
                lazy = TableAdam(tables, table_params=tables, lr=1e-3, weight_decay=1e-5, lazy_k=k)
                lazy.sync_hyper()
                _lib.call("rh_set_tuning", 2, grid)
                for _ in range(k + 2):  # reach the steady state: every swept row lags exactly k steps
                    lazy.step_tables()
                us = timeit(lazy.step_tables, 40, warm=0)
                print(f"adam lazy  K={k:3d} grid={grid:5d} {us:9.1f} us per step (sweep only, no touched rows)", flush=True)
                del lazy
        _lib.call("rh_set_tuning", 2, 0)


if __name__ == "__main__":
    main()

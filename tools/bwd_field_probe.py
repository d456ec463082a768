#!/usr/bin/env python
"""rh_embed_bwd on ONE field at a time (F = 1) per vocabulary size: which accumulation path costs what.
    python tools/bwd_field_probe.py [--batch 65536]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.kbench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--fields", type=int, default=1, help="copies of the field in one launch (blocks in flight)")
    args = ap.parse_args()
    from torch_rechub_amd import _lib, ops
    dev = torch.device("cuda:0")
    D, B, F = 16, args.batch, args.fields
    g = torch.Generator(device=dev).manual_seed(1)
    print(f"B={B} F={F} (same vocab x F)   columns: auto/slabs64 | auto/slabs1e5 | no-reg(LDS)/1e5 | global atomics | no sink")
    for V in [3, 4, 10, 27, 32, 105, 305, 512, 583, 1460, 14992, 286181, 10131227]:
        tables = [torch.nn.Parameter(torch.randn(V, D, device=dev, generator=g) * 1e-2) for _ in range(F)]
        idx = torch.stack([torch.randint(0, V, (B,), device=dev, generator=g) for _ in range(F)], 1)
        call = ops.EmbedCall(tables, [None] * F, [idx[:, f] for f in range(F)], [], want_fm=True, want_lr=True)
        lr_w = torch.randn(1, F * D, device=dev)
        out = torch.randn(B, F * D, device=dev)
        ssum = torch.randn(B, D, device=dev)
        g_out = torch.randn(B, F * D, device=dev)
        g_y = torch.randn(B, device=dev)
        fdesc, idesc = call.fdesc(True), call.idesc()
        nch = _lib.call("rh_embed_bwd_nchunks", B, 0)
        partial = torch.empty(nch, F * D, device=dev)

        def bwd():
            _lib.call("rh_embed_bwd", ops._p(fdesc), ops._p(idesc), 1, B, F, D, ops._p(g_out), g_out.stride(0),
                      ops._p(out), out.stride(0), ops._p(ssum), ops._p(g_y), ops._p(g_y), ops._p(lr_w), ops._p(partial),
                      1.0, 0, ops._p(None), 0, ops._p(ops.err_flag(dev)), ops._stream())

        res = []
        for path, slabs in [(0, 64), (0, 100000), (2, 100000), (1, 64), (3, 64)]:
            _lib.call("rh_set_tuning", 6, path)
            _lib.call("rh_set_tuning", 5, slabs)
            res.append(timeit(bwd, args.iters))
        _lib.call("rh_set_tuning", 6, 0)
        _lib.call("rh_set_tuning", 5, 64)
        print(f"V={V:9d}  " + "  ".join(f"{u:8.1f}" for u in res), flush=True)
        del tables


if __name__ == "__main__":
    main()

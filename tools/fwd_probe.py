#!/usr/bin/env python
"""rh_embed_fwd kernels (RH_TUNE_FWD_PATH) on the DeepFM / Criteo call (26 tables, D = 16, 13 dense columns, FM + LR), with the index set
CYCLED between launches (6 sets) so that the 256 MiB Infinity Cache cannot keep the rows of the previous launch:
    python tools/fwd_probe.py [--batches 4096,65536] [--vars 0,1,2,3] [--splits 0]
Prints us per launch and the fraction of the 8 TB/s HBM peak on the algorithmic bytes (bench.py's figure)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CRITEO_VOCABS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="4096,8192,16384,32768,65536")
    ap.add_argument("--vars", default="1,2,3,4", help="RH_TUNE_FWD_PATH values: 1 lane-split kernel, 2 / 3 / 4 field-uniform kernel, 4 / 2 / 1 wavefronts per sample group")
    ap.add_argument("--splits", default="0")
    ap.add_argument("--sets", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=30)
    ap.add_argument("--index", default="random", help="random | seq (row = sample number: perfect locality) | hot (random inside the first 4096 rows)")
    args = ap.parse_args()
    from torch_rechub_amd import _lib, ops
    dev = torch.device("cuda:0")
    D, F, ND = 16, len(CRITEO_VOCABS), 13
    g = torch.Generator(device=dev).manual_seed(1)
    tables = [torch.nn.Parameter(torch.randn(v, D, device=dev, generator=g) * 1e-2) for v in CRITEO_VOCABS]
    lr_w, lr_b = torch.randn(1, F * D, device=dev), torch.randn(1, device=dev)
    for B in [int(b) for b in args.batches.split(",")]:
        calls = []
        for _ in range(args.sets):
            if args.index == "seq":
                base = torch.arange(B, device=dev) + len(calls) * B
                idx = torch.stack([base % v for v in CRITEO_VOCABS], 1)
            elif args.index == "hot":
                idx = torch.stack([torch.randint(0, min(v, 4096), (B,), device=dev, generator=g) for v in CRITEO_VOCABS], 1)
            else:
                idx = torch.stack([torch.randint(0, v, (B,), device=dev, generator=g) for v in CRITEO_VOCABS], 1)
            dense = torch.rand(B, ND, device=dev, generator=g)
            call = ops.EmbedCall(tables, [None] * F, [idx[:, f] for f in range(F)], [dense[:, j] for j in range(ND)],
                                 want_fm=True, want_lr=True)
            calls.append((call, call.fdesc(False), call.idesc(), call.ddesc(), idx, dense))
        pitch = ((F * D + ND + 15) // 16) * 16
        out = torch.empty(B, pitch, device=dev)
        fm, lr, ssum = torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, D, device=dev)
        alg = B * (F * (8 + 64 + 64) + ND * 8 + 8 + D * 4)
        for split in [int(x) for x in args.splits.split(",")]:
            for var in [int(x) for x in args.vars.split(",")]:
                _lib.call("rh_set_tuning", 7, var)

                def one(c):
                    _lib.call("rh_embed_fwd", ops._p(c[1]), ops._p(c[2]), 1, B, F, D, ops._p(c[3]), ND, F * D, ops._p(out),
                              out.stride(0), ops._p(lr_w), ops._p(lr_b), ops._p(lr), ops._p(fm), ops._p(ssum), split,
                              ops._p(ops.err_flag(dev)), ops._stream())

                for c in calls:
                    one(c)
                torch.cuda.synchronize()
                if var == int(args.vars.split(",")[0]):
                    ref = (out.clone(), fm.clone(), lr.clone(), ssum.clone())
                else:  # same last call: the gathered block must be identical, the sums equal up to summation order
                    ok = (torch.equal(out[:, :F * D + ND], ref[0][:, :F * D + ND]) and
                          torch.allclose(fm, ref[1], rtol=1e-4, atol=1e-6) and torch.allclose(lr, ref[2], rtol=1e-4, atol=1e-5)
                          and torch.allclose(ssum, ref[3], rtol=1e-5, atol=1e-6))
                    if not ok:
                        print(f"  !! var {var} differs from var {args.vars.split(',')[0]}: out {torch.equal(out[:, :F*D+ND], ref[0][:, :F*D+ND])} "
                              f"fm {(fm - ref[1]).abs().max().item():.3e} lr {(lr - ref[2]).abs().max().item():.3e} "
                              f"S {(ssum - ref[3]).abs().max().item():.3e}")
                best = []
                for rep in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for r in range(args.rounds):
                        for c in calls:
                            one(c)
                    e1.record()
                    torch.cuda.synchronize()
                    best.append(e0.elapsed_time(e1) * 1e3 / (args.rounds * len(calls)))
                us = min(best)
                print(f"B={B:6d} split={split} var={var:2d}  {us:7.2f} us  (runs {', '.join(f'{b:.2f}' for b in best)})  "
                      f"{alg / us / 1e3:6.0f} GB/s  frac {alg / us / 1e3 / 8000:.3f}", flush=True)
        _lib.call("rh_set_tuning", 7, 0)


if __name__ == "__main__":
    main()

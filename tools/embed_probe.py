#!/usr/bin/env python
"""Where does the time of rh_embed_fwd / rh_embed_bwd go?  Times the raw launches on subsets of the Criteo fields
(tiny <= 512 rows: LDS-aggregated in the backward; mid 583..14992 rows: L2-resident; large >= 93145 rows: HBM) at
several batch sizes, int64 vs int32 indices.   python tools/embed_probe.py [--iters 100]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CRITEO_VOCABS  # noqa: E402
from tools.kbench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--batches", default="4096,65536")
    ap.add_argument("--slabs", default="0", help="RH_TUNE_BWD_PATH values to sweep for the backward (0 auto, 4 = chunk-fastest block order)")
    ap.add_argument("--dtypes", default="i64")
    args = ap.parse_args()
    from torch_rechub_amd import _lib, ops
    dev = torch.device("cuda:0")
    D = 16
    g = torch.Generator(device=dev).manual_seed(1)
    all_tables = [torch.nn.Parameter(torch.randn(v, D, device=dev, generator=g) * 1e-2) for v in CRITEO_VOCABS]
    groups = {"all": list(range(26)),
              "tiny": [i for i, v in enumerate(CRITEO_VOCABS) if v <= 512],
              "mid": [i for i, v in enumerate(CRITEO_VOCABS) if 512 < v <= 20000],
              "large": [i for i, v in enumerate(CRITEO_VOCABS) if v > 20000]}
    groups["tiny+mid"] = groups["tiny"] + groups["mid"]
    groups["mid+large"] = groups["mid"] + groups["large"]
    for B in [int(b) for b in args.batches.split(",")]:
        for name, members in groups.items():
            F = len(members)
            tables = [all_tables[i] for i in members]
            vocabs = [CRITEO_VOCABS[i] for i in members]
            for idt in [dict(i64=torch.int64, i32=torch.int32)[d] for d in args.dtypes.split(",")]:
                idx = torch.stack([torch.randint(0, v, (B,), device=dev, generator=g) for v in vocabs], 1).to(idt)
                cols = [idx[:, f] for f in range(F)]
                lr_w = torch.randn(1, F * D, device=dev)
                lr_b = torch.randn(1, device=dev)
                nd = 13 if name == "all" else 0  # the DeepFM call: 13 dense columns appended
                dense = torch.rand(B, 13, device=dev, generator=g)
                call = ops.EmbedCall(tables, [None] * F, cols, [dense[:, j] for j in range(nd)], want_fm=True, want_lr=True)
                out = torch.empty(B, F * D + nd, device=dev)
                fm, lr, ssum = torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, D, device=dev)
                fdesc, idesc = call.fdesc(False), call.idesc()
                i64 = 1 if idt == torch.int64 else 0

                def fwd():
                    _lib.call("rh_embed_fwd", ops._p(fdesc), ops._p(idesc), i64, B, F, D, ops._p(call.ddesc()), nd, F * D,
                              ops._p(out), out.stride(0), ops._p(lr_w), ops._p(lr_b), ops._p(lr), ops._p(fm),
                              ops._p(ssum), 0, ops._p(ops.err_flag(dev)), ops._stream())

                us_f = timeit(fwd, args.iters)
                g_out = torch.randn(B, F * D + nd, device=dev)
                g_y = torch.randn(B, device=dev)
                fdesc_g = call.fdesc(True)
                SPB = int(os.environ.get("PROBE_SPB", "0"))
                nch = _lib.call("rh_embed_bwd_nchunks", B, SPB)
                partial = torch.empty(nch, F * D, device=dev)

                def bwd():
                    _lib.call("rh_embed_bwd", ops._p(fdesc_g), ops._p(idesc), i64, B, F, D, ops._p(g_out),
                              g_out.stride(0), ops._p(out), out.stride(0), ops._p(ssum), ops._p(g_y), ops._p(g_y),
                              ops._p(lr_w), ops._p(partial), 1.0, 0, ops._p(None), SPB, ops._p(ops.err_flag(dev)),
                              ops._stream())

                us_b = []
                for sl in [int(x) for x in args.slabs.split(",")]:
                    _lib.call("rh_set_tuning", 6, sl)
                    us_b.append(timeit(bwd, args.iters))
                us_all, us_b = us_b, us_b[0]
                ib = 8 if i64 else 4
                fb = F * (ib + 128) + 8 + 8 * nd
                bb = F * (ib + 192) + 4
                print(f"B={B:6d} {name:9s} F={F:2d} idx={'i64' if i64 else 'i32'}  fwd {us_f:8.2f} us {fb * B / us_f / 1e3:6.0f} GB/s | "
                      f"bwd {us_b:8.2f} us {bb * B / us_b / 1e3:6.0f} GB/s  paths {args.slabs}: " +
                      " ".join(f"{u:.1f}" for u in us_all), flush=True)
        for w in all_tables:
            ops.grad_buffer(w).zero_()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Where is the ceiling of rh_embed_bwd at large batch?  (north_star: >= 50 % of the 8 TB/s HBM peak on the fused gather +
interaction pair; SURVEY 8(d): 5 204 algorithmic bytes per sample in the backward at F = 26, D = 16, int64 indices.)

    python tools/bwd_ceiling_probe.py [--batches 16384,65536] [--sets 8] [--iters 40]
    rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python tools/bwd_ceiling_probe.py --pmc      (one launch per variant and set)

The full DeepFM backward (26 Criteo tables, 2 GiB; uniform indices -- the worst case) in five forms:
    product      path 0: row-wide float atomics for large tables, LDS-parked sums for tables of <= 32 rows
    plain-wide   path 5: the SAME memory requests with every atomic replaced by a plain store            (timing only)
    plain-16B    path 6: one 16-byte store per lane, no re-layout -- the cheapest scatter there is        (timing only)
    no-sink      path 3: everything but the scatter (index, upstream gradient, embedding row, S read; LR partials written)
    fwd          rh_embed_fwd on the same lookups, for the pair's figure
Every launch of a variant takes the NEXT of `--sets` independent index sets (sets x B x 26 rows of 64 B: past the 256 MiB
Infinity Cache from 4 sets of 65536 samples up), so no launch finds its table-gradient rows on die.  plain-* bound every
scheme that writes each lookup's gradient row ONCE -- a sort-ahead / segmented reduction included: with uniform indices over
10 M-row tables a batch holds next to no duplicates to merge, and a sorted form reads the upstream gradient rows through a
permutation (a random 64-byte read stream on top).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import BWD_BYTES_PER_SAMPLE, CRITEO_VOCABS, FWD_BYTES_PER_SAMPLE, HBM_PEAK_GBS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="16384,65536")
    ap.add_argument("--sets", type=int, default=8)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--pmc", action="store_true", help="one pass of launches only (run under rocprofv3 --pmc)")
    args = ap.parse_args()
    from torch_rechub_amd import _lib, ops
    dev = torch.device("cuda:0")
    D, F, ND = 16, len(CRITEO_VOCABS), 13
    g = torch.Generator(device=dev).manual_seed(1)
    tables = [torch.nn.Parameter(torch.randn(v, D, device=dev, generator=g) * 1e-2) for v in CRITEO_VOCABS]
    lr_w = torch.randn(1, F * D, device=dev)
    lr_b = torch.randn(1, device=dev)
    err = ops.err_flag(dev)
    print(f"# rh_embed_bwd ceiling probe: F={F} D={D} int64 indices, uniform, {args.sets} index sets cycled; "
          f"algorithmic bytes / sample fwd {FWD_BYTES_PER_SAMPLE} bwd {BWD_BYTES_PER_SAMPLE}; peak {HBM_PEAK_GBS} GB/s")
    for B in [int(b) for b in args.batches.split(",")]:
        sets = []
        for _ in range(args.sets):
            idx = torch.stack([torch.randint(0, v, (B,), device=dev, generator=g) for v in CRITEO_VOCABS], 1)
            dense = torch.rand(B, ND, device=dev, generator=g)
            call = ops.EmbedCall(tables, [None] * F, [idx[:, f] for f in range(F)], [dense[:, j] for j in range(ND)],
                                 want_fm=True, want_lr=True)
            sets.append(dict(call=call, idx=idx, dense=dense, fdesc=call.fdesc(False), fdesc_g=call.fdesc(True), idesc=call.idesc(),
                             ddesc=call.ddesc()))
        out = torch.empty(B, F * D + ND, device=dev)
        fm, lr, ssum = torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, D, device=dev)
        g_out = torch.randn(B, F * D + ND, device=dev)
        g_y = torch.randn(B, device=dev)
        nch = _lib.call("rh_embed_bwd_nchunks", B, 0)
        partial = torch.empty(nch, F * D, device=dev)

        def fwd(s):
            _lib.call("rh_embed_fwd", ops._p(s["fdesc"]), ops._p(s["idesc"]), 1, B, F, D, ops._p(s["ddesc"]), ND, F * D,
                      ops._p(out), out.stride(0), ops._p(lr_w), ops._p(lr_b), ops._p(lr), ops._p(fm), ops._p(ssum), 0,
                      ops._p(err), ops._stream())

        def bwd(s):
            _lib.call("rh_embed_bwd", ops._p(s["fdesc_g"]), ops._p(s["idesc"]), 1, B, F, D, ops._p(g_out), g_out.stride(0),
                      ops._p(out), out.stride(0), ops._p(ssum), ops._p(g_y), ops._p(g_y), ops._p(lr_w), ops._p(partial), 1.0,
                      0, ops._p(None), 0, ops._p(err), ops._stream())

        def time_variant(fn, path):
            _lib.call("rh_set_tuning", 6, path)
            for s in sets[:2]:
                fn(s)
            torch.cuda.synchronize()
            if args.pmc:
                for s in sets:
                    fn(s)
                torch.cuda.synchronize()
                return float("nan")
            # one hipGraph of `sets` launches (launch-to-launch gaps of eager launches would be part of the figure otherwise)
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for s in sets:
                        fn(s)
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / (args.iters * len(sets)) * 1e3  # us per launch

        fwd(sets[0])
        rows = []
        us_f = time_variant(fwd, 0)
        rows.append(("fwd", us_f, FWD_BYTES_PER_SAMPLE))
        for name, path in (("product", 0), ("plain-wide", 5), ("plain-16B", 6), ("no-sink", 3)):
            rows.append((name, time_variant(bwd, path), BWD_BYTES_PER_SAMPLE))
        _lib.call("rh_set_tuning", 6, 0)
        for w in tables:
            ops.grad_buffer(w).zero_()
        if args.pmc:
            print(f"B={B}: pmc pass done ({len(sets)} launches per variant, order: fwd, product, plain-wide, plain-16B, no-sink)")
            continue
        for name, us, nbytes in rows:
            gbs = nbytes * B / us / 1e3
            print(f"B={B:6d}  {name:10s} {us:8.2f} us  {gbs:7.0f} GB/s (algorithmic)  = {gbs / HBM_PEAK_GBS:.3f} of the {HBM_PEAK_GBS:.0f} GB/s peak",
                  flush=True)
        prod = rows[1][1]
        pair = (FWD_BYTES_PER_SAMPLE + BWD_BYTES_PER_SAMPLE) * B / (us_f + prod) / 1e3
        print(f"B={B:6d}  pair fwd + product: {pair:7.0f} GB/s = {pair / HBM_PEAK_GBS:.3f} of peak", flush=True)


if __name__ == "__main__":
    main()

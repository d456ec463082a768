#!/usr/bin/env python
"""Why is the pre-gather refresh of DSSM's 204 800 history lookups (100 M-row table) 4x slower per lookup than DeepFM's?
Times rh_adam_lazy_touched(refresh) alone for index sets of several shapes / table sizes, rows lagging uniformly in [0, K).
    python tools/refresh_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_rechub_amd import ops
from torch_rechub_amd.optim import TableAdam

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
K = 64
for V, B, L, pad_frac in ((100_000_000, 4096, 50, 0.5), (100_000_000, 4096, 50, 0.0), (10_000_000, 4096, 50, 0.0),
                          (10_000_000, 4096, 26, 0.0), (100_000_000, 4096, 1, 0.0), (100_000_000, 65536, 1, 0.0)):
    table = torch.nn.Parameter(torch.randn(V, 16, device=dev, generator=g) * 1e-2)
    opt = TableAdam([table], table_params=[table], lr=1e-3, weight_decay=1e-5, lazy_k=K)
    opt.overlap_sweep = False
    opt.sync_hyper()
    ops.grad_buffer(table)
    opt._k_decided = True
    # rows "last" uniformly behind: emulate the steady state (device step counter K, last in [0, K))
    for _ in range(K):
        opt.step_tables()
    torch.cuda.synchronize()
    times = []
    for rep in range(6):
        idx = torch.randint(1, V, (B, L), device=dev, generator=g)
        if pad_frac:
            lens = torch.randint(1, L + 1, (B,), device=dev, generator=g)
            idx.masked_fill_(torch.arange(L, device=dev)[None, :] >= lens[:, None], 0)
        flat = os.environ.get("FLAT", "0") == "1"
        if flat:  # the sequence-feature call shape: ONE field of B * L lookups (idx read as a flat column)
            key = (idx.data_ptr(), 1, 0)
        else:
            cols = [idx[:, j] for j in range(L)]
            key = tuple([c.data_ptr() for c in cols] + [idx.stride(0)] * L + list(range(L)))
        idesc = ops.EmbedCall._icache.get(key, dev)
        opt.step_tables()  # one more step: everything not refreshed lags
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if flat:
            ops._pre_gather([table], [0 if pad_frac else None], idesc, 1, B * L, 1, 16, training=True)
        else:
            ops._pre_gather([table] * L, [0 if pad_frac else None] * L, idesc, 1, B, L, 16, training=True)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3)
    n = B * L * (1 - pad_frac / 2 if pad_frac else 1)
    print(f"V={V:>11,d} B={B} L={L} pad={pad_frac}: refresh {min(times[1:]):7.1f} us (median {sorted(times[1:])[len(times[1:]) // 2]:7.1f})  "
          f"= {min(times[1:]) * 1e3 / n:5.2f} ns per live lookup", flush=True)
    del opt, table
    torch.cuda.empty_cache()

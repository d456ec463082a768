#!/usr/bin/env python
"""One fixed configuration per kernel, so that `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE` gives a clean mean per launch:
  rh_adam_dense (known byte count: calibration of the counters), rh_embed_fwd / rh_embed_bwd at B=4096 (Criteo shape),
  rh_adam_lazy_step (touched rows of a batch + window sweep, one launch) in steady state (K=64, no flush inside the
  measured launches).
Run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python tools/pmc_probe.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CRITEO_VOCABS  # noqa: E402


def main():
    from torch_rechub_amd import ops
    from torch_rechub_amd.optim import TableAdam
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    F, D, B = len(CRITEO_VOCABS), 16, int(os.environ.get("PMC_B", "4096"))
    tables = [torch.nn.Parameter(torch.randn(v, D, device=dev, generator=g) * 1e-2) for v in CRITEO_VOCABS]
    lr_w = torch.nn.Parameter(torch.randn(1, F * D, device=dev))
    lr_b = torch.nn.Parameter(torch.randn(1, device=dev))
    dense = TableAdam(tables, table_params=tables, lr=1e-3, weight_decay=1e-5)
    for _ in range(5):
        dense.step_tables()
    torch.cuda.synchronize()
    del dense
    for it in range(20):
        idx = torch.stack([torch.randint(0, v, (B,), device=dev, generator=g) for v in CRITEO_VOCABS], 1)
        dn = torch.rand(B, 13, device=dev, generator=g)
        call = ops.EmbedCall(tables, [None] * F, [idx[:, f] for f in range(F)], [dn[:, j] for j in range(13)], want_fm=True,
                             want_lr=True)
        out, fm, lr = ops.fused_embedding(call, lr_w, lr_b)
        torch.autograd.backward([out, fm, lr], [torch.randn_like(out), torch.randn_like(fm), torch.randn_like(lr)])
        for w in tables:
            ops.grad_buffer(w).zero_()
    lazy = TableAdam(tables, table_params=tables, lr=1e-3, weight_decay=1e-5, lazy_k=64)
    idx = torch.stack([torch.randint(0, v, (B,), device=dev, generator=g) for v in CRITEO_VOCABS], 1)
    call = ops.EmbedCall(tables, [None] * F, [idx[:, f] for f in range(F)], [], want_fm=False, want_lr=False)
    for it in range(64 + 20):  # first 64 launches reach the steady state (every swept row lags 64 steps)
        # every step has ONE index batch over the tables, as the DeepFM step: the end-of-step launch is then the merged one
        # (rh_adam_lazy_step = touched rows + window sweep, `adam_lazy_sweep_kernel<4, true>` in the trace)
        ops._log_touch(call.weights, call.pads, call.idesc(), call.idx_is_i64, B, F, D, call.idx)
        lazy.step_tables()
    torch.cuda.synchronize()
    print("pmc probe done")


if __name__ == "__main__":
    main()

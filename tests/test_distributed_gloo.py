"""Data-parallel exchange on CPU tensors over gloo, world_size 2 (the N>1 path of bench.py uses the same code
over RCCL).  Checks that (flat dense all-reduce) + (sparse row all-gather + scatter) reproduce the gradients of
one process seeing the concatenated global batch — i.e. what nn.DataParallel computes in the reference
(trainers/ctr_trainer.py:53-55)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from oracle import ctr_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net():
    torch.manual_seed(7)
    return nn.Sequential(nn.Linear(5, 4), nn.ReLU(), nn.Linear(4, 1))


class _FakeCall(object):

    def __init__(self, idx_cols):
        self.idx = idx_cols


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch_rechub_amd import ops
        from torch_rechub_amd.distributed import DataParallelContext
        net = _net()
        if rank == 1:  # replicas start different: the context must broadcast rank 0's weights
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(1.0)
        ctx = DataParallelContext(net)
        g = torch.Generator().manual_seed(100 + rank)
        x = torch.randn(6, 5, generator=g)
        ctx.bucket.zero()
        loss = net(x).mean() / world
        loss.backward()
        assert ctx.bucket.all_present()
        # the first layer's weight gradient is "late": flush() part-way must only pack / reduce what exists
        late = net[0].weight.grad
        net[0].weight.grad = None
        ctx.bucket.flush()
        assert ctx.bucket.packed[1] and not ctx.bucket.packed[0]
        net[0].weight.grad = late
        ctx.bucket.finish(assign_views=True)
        dense = [p.grad.clone() for p in net.parameters()]
        # sparse exchange: per-rank index columns + gradient rows
        B, F, D = 6, 3, 4
        idx = torch.randint(0, 5, (B, F), generator=g)
        rows = torch.randn(B, F, D, generator=g)
        idx_all, rows_all = ctx.sparse_exchange(_FakeCall([idx[:, f] for f in range(F)]), rows)
        ctx.close()
        assert ops._sparse_exchange is None
        if rank == 0:
            torch.save({"dense": dense, "idx_all": idx_all, "rows_all": rows_all,
                        "w0": [p.detach().clone() for p in net.parameters()]}, out)
    finally:
        dist.destroy_process_group()


def test_world2_gradients_equal_global_batch(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    # single process, global batch = concat of both ranks' local batches
    net = _net()
    xs, idxs, rows = [], [], []
    for r in range(2):
        g = torch.Generator().manual_seed(100 + r)
        xs.append(torch.randn(6, 5, generator=g))
        idxs.append(torch.randint(0, 5, (6, 3), generator=g))
        rows.append(torch.randn(6, 3, 4, generator=g))
    for p, w in zip(net.parameters(), got["w0"]):
        assert torch.equal(p, w)  # broadcast from rank 0 happened
    net(torch.cat(xs)).mean().backward()
    for p, d in zip(net.parameters(), got["dense"]):
        np.testing.assert_allclose(d.numpy(), p.grad.numpy(), rtol=1e-5, atol=1e-6)
    assert torch.equal(got["idx_all"], torch.cat(idxs))
    assert torch.equal(got["rows_all"], torch.cat(rows))
    shapes = [(5, 4)] * 3
    a = O.embedding_backward(shapes, got["idx_all"].numpy(), got["rows_all"].numpy().astype(np.float64))
    b = O.embedding_backward(shapes, torch.cat(idxs).numpy(), torch.cat(rows).numpy().astype(np.float64))
    for ga, gb in zip(a, b):
        np.testing.assert_array_equal(ga, gb)


def test_bench_gpus_n_starts_its_own_ranks_dry_run_on_gloo():
    """`python bench.py --gpus 2` with no torchrun environment must start two ranks by itself (round-3 verdict: it ran ONE
    rank and printed n_gpus = 1).  The launcher + rendezvous path is exercised on CPU with gloo; stdout is exactly one
    JSON line from rank 0 with n_gpus = 2."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--launch-dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["gloo_ranks"] == 2 and line["dry_run"] is True
    assert "launching 2 ranks" in r.stderr


def test_bench_refuses_to_measure_fewer_ranks_than_asked_for():
    """Without enough visible devices `--gpus N` is an error, not a one-rank line (no GPU in the CPU container: 0 < 2);
    a torchrun environment whose WORLD_SIZE differs from --gpus is refused too."""
    import os
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "refusing to measure fewer" in r.stderr and r.stdout.strip() == ""
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--launch-dry-run"],
                       env=env2, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""

"""Row-sharded tables and cross-rank in-batch negatives: host logic and collectives on CPU tensors over gloo (the GPU
path runs the same Python over RCCL; the lookups themselves are HIP only and are covered by tests/test_gpu_world2.py
and tests/test_gpu_kernels.py)."""
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


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_oracle_sharded_gather_is_the_plain_gather(world):
    """The decomposition the HIP path relies on: per-rank gathers at the localised indices sum to the lookup of the
    full tables, bit for bit (one non-zero term per element), padding rows and ragged shard sizes included."""
    rng = np.random.default_rng(world)
    vocabs = [1, 3, 10, 17, 64, 1000]
    pads = [None, 0, 2, None, 63, None]
    tables = [rng.standard_normal((v, 8)).astype(np.float32) for v in vocabs]
    for t, p in zip(tables, pads):
        if p is not None:
            t[p] = 0  # nn.Embedding(padding_idx) keeps that row zero (initializers.py:17-20)
    idx = np.stack([rng.integers(0, v, 300) for v in vocabs], axis=1)
    got = O.sharded_embedding_gather(tables, idx, world, pads)
    assert np.array_equal(got, O.embedding_gather(tables, idx))
    for r in range(world):
        loc = O.shard_localize(idx, vocabs, pads, world, r)
        assert loc.dtype == np.int32 and (loc >= 0).all()
        assert (loc <= np.array([-(-v // world) for v in vocabs])).all()
    with pytest.raises(IndexError):
        O.shard_localize(np.array([[1, 0, 0, 0, 0, 0]]), vocabs, pads, world, 0)


def test_oracle_sampler_rows_of_a_global_batch():
    """Slices of rows drawn by different 'ranks' are the rows one process draws for the global batch."""
    full = O.inbatch_sample_rows(seed=7, ctr=3, B=12, cols=12, row0=0, K=5)
    for row0, B in ((0, 4), (4, 4), (8, 4), (3, 9)):
        assert np.array_equal(O.inbatch_sample_rows(7, 3, B, 12, row0, 5), full[row0:row0 + B])
    for i, row in enumerate(full):
        assert i not in row and len(set(row.tolist())) == 5 and row.min() >= 0 and row.max() < 12
    assert not np.array_equal(full, O.inbatch_sample_rows(7, 4, 12, 12, 0, 5))  # the call counter moves the stream
    everything = O.inbatch_sample_rows(1, 0, 6, 6, 0, 5)
    assert all(sorted(r.tolist()) == [c for c in range(6) if c != i] for i, r in enumerate(everything))


def test_rectangular_inbatch_sampling_on_cpu_tensors():
    from torch_rechub_amd.utils.match import gather_inbatch_logits, inbatch_negative_sampling
    g = torch.Generator().manual_seed(0)
    scores = torch.randn(4, 12, generator=g)  # rows 4..7 of a 12 x 12 global batch
    hard = inbatch_negative_sampling(scores, neg_ratio=3, hard_negative=True, row_offset=4)
    masked = scores.clone()
    masked[torch.arange(4), torch.arange(4) + 4] = float("-inf")
    assert torch.equal(hard, torch.topk(masked, 3, dim=1).indices)
    rnd = inbatch_negative_sampling(scores, neg_ratio=6, generator=torch.Generator().manual_seed(1), row_offset=4)
    assert rnd.shape == (4, 6) and not (rnd == (torch.arange(4) + 4).unsqueeze(1)).any()
    assert all(len(set(r.tolist())) == 6 for r in rnd)
    assert inbatch_negative_sampling(scores, row_offset=4).shape == (4, 11)
    logits = gather_inbatch_logits(scores, hard, row_offset=4)
    assert torch.equal(logits[:, 0], scores[torch.arange(4), torch.arange(4) + 4]) and logits.shape == (4, 4)
    with pytest.raises(ValueError):
        inbatch_negative_sampling(scores, row_offset=9)


def _tables():
    torch.manual_seed(5)
    a = nn.Embedding(11, 4)
    b = nn.Embedding(7, 4, padding_idx=3)
    return nn.ModuleDict({"a": a, "b": b})


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch_rechub_amd import sharding
        model = _tables()
        full = {k: v.clone() for k, v in model.state_dict().items()}
        mods = sharding.shard_tables(model)
        assert len(mods) == 2 and all(sharding.is_sharded(m) for m in mods)
        sh = model["a"]._rh_shard
        assert (sh.world, sh.rank, sh.sink) == (world, rank, -(-11 // world))
        # the shard is the oracle's: owned rows, zero tail, zero sink row
        for name in ("a", "b"):
            want = O.shard_rows(full[name + ".weight"].numpy(), world, rank)
            assert np.array_equal(model[name].weight.detach().numpy(), want)
        with pytest.raises(RuntimeError):
            model["a"](torch.tensor([0]))  # a shard must not be indexed with global ids
        # full_state_dict reassembles the reference layout on every rank
        sd = sharding.full_state_dict(model)
        for k, v in full.items():
            assert torch.equal(sd[k], v), k
        # ... and load_full_state_dict deals a reference-layout checkpoint out again
        bumped = {k: v + 1.0 for k, v in full.items()}
        sharding.load_full_state_dict(model, bumped)
        want = O.shard_rows(bumped["a.weight"].numpy(), world, rank)
        got = model["a"].weight.detach().numpy()
        assert np.array_equal(got[:len(range(rank, 11, world))], want[:len(range(rank, 11, world))])
        # the data-parallel context in sharded mode: replicas made equal first, then dealt out; only the dense
        # parameters go into the all-reduce bucket; a flagged small table (BST's positions) stays a dense parameter
        from torch_rechub_amd import ops
        from torch_rechub_amd.distributed import DataParallelContext, table_parameters
        net = nn.ModuleDict({"emb": nn.Embedding(9, 4), "pos": nn.Embedding(5, 4), "fc": nn.Linear(4, 2)})
        net["pos"]._rh_dense = True
        if rank == 1:
            with torch.no_grad():
                net["emb"].weight.add_(1.0)
        ctx = DataParallelContext(net, shard_tables=True)
        assert [m is net["emb"] for m in ctx.sharded] == [True]
        assert net["emb"].weight.shape == (-(-9 // world) + 1, 4) and net["pos"].weight.shape == (5, 4)
        assert [tuple(p.shape) for p in table_parameters(net)] == [tuple(net["emb"].weight.shape)]
        assert {id(p) for p in ctx.bucket.params} == {id(net["pos"].weight), id(net["fc"].weight), id(net["fc"].bias)}
        assert ops._sparse_exchange is not None
        rows = sharding.full_table(net["emb"])
        torch.save(rows, os.path.join(outdir, f"emb{rank}.pt"))  # rank 0's values everywhere (broadcast before dealing)
        ctx.close()
        assert ops._sparse_exchange is None
        # differentiable collectives: scatter_rows_sum (forward reduce-scatter, backward all-gather) and gather_rows
        B, C = 3, 5
        g = torch.Generator().manual_seed(40 + rank)
        x_all = torch.randn(world * B, C, generator=g, requires_grad=True)
        w = torch.randn(B, C, generator=g)
        y = sharding.scatter_rows_sum(x_all)
        (y * w).sum().backward()
        x = torch.randn(B, C, generator=g, requires_grad=True)
        w_all = torch.randn(world * B, C, generator=g)
        z = sharding.gather_rows(x)
        (z * w_all).sum().backward()
        torch.save({"y": y.detach(), "gx_all": x_all.grad, "z": z.detach(), "gx": x.grad},
                   os.path.join(outdir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_roundtrip_and_row_collectives_over_gloo(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    tables = [torch.load(os.path.join(tmp_path, f"emb{r}.pt")) for r in range(world)]
    assert all(t.shape == (9, 4) and torch.equal(t, tables[0]) for t in tables)
    B, C = 3, 5
    xs, ws, x2, w2 = [], [], [], []
    for r in range(world):
        g = torch.Generator().manual_seed(40 + r)
        xs.append(torch.randn(world * B, C, generator=g))
        ws.append(torch.randn(B, C, generator=g))
        x2.append(torch.randn(B, C, generator=g))
        w2.append(torch.randn(world * B, C, generator=g))
    total = sum(xs)
    for r in range(world):
        np.testing.assert_allclose(got[r]["y"].numpy(), total[r * B:(r + 1) * B].numpy(), rtol=1e-6, atol=1e-6)
        # d/dx_all of sum_r' <y_r', w_r'> on rank r: block r' of x_all(r) feeds y_r' -> w_r'
        assert torch.equal(got[r]["gx_all"], torch.cat(ws))
        assert torch.equal(got[r]["z"], torch.cat(x2))
        # rows of rank r appear in every rank's gathered copy: the gradients add up
        want = sum(w2[q][r * B:(r + 1) * B] for q in range(world))
        np.testing.assert_allclose(got[r]["gx"].numpy(), want.numpy(), rtol=1e-6, atol=1e-6)


# -- cross-rank in-batch negatives on CPU tensors: two ranks == one process on the global batch -----------------------
def _towers():
    torch.manual_seed(21)
    return nn.Linear(6, 4), nn.Linear(5, 4)


def _negatives_loss(user, item, rows, world, rank, gather):
    """The in-batch step of MatchTrainer._compute_loss (mode 0, hard negatives: deterministic) on plain tensors."""
    import torch.nn.functional as F
    from torch_rechub_amd.utils.match import gather_inbatch_logits, inbatch_negative_sampling
    u, v = F.normalize(user, dim=1), F.normalize(item, dim=1)
    row0 = 0
    if gather is not None:
        row0 = rank * rows
        v = gather(v)
    scores = u @ v.t()
    neg = inbatch_negative_sampling(scores, neg_ratio=3, hard_negative=True, row_offset=row0)
    logits = gather_inbatch_logits(scores, neg, row_offset=row0)
    return F.cross_entropy(logits, torch.zeros(logits.shape[0], dtype=torch.long))


def _neg_worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch_rechub_amd import sharding
        fu, fi = _towers()
        g = torch.Generator().manual_seed(50)
        xu, xi = torch.randn(world * 8, 6, generator=g), torch.randn(world * 8, 5, generator=g)
        sl = slice(rank * 8, (rank + 1) * 8)
        loss = _negatives_loss(fu(xu[sl]), fi(xi[sl]), 8, world, rank, sharding.gather_rows)
        (loss / world).backward()  # the trainer's scaling: gradients are summed over ranks
        grads = [p.grad.clone() for p in list(fu.parameters()) + list(fi.parameters())]
        for t in grads:
            dist.all_reduce(t)
        torch.save({"loss": loss.detach(), "grads": grads}, os.path.join(outdir, f"n{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_cross_rank_negatives_equal_one_process_on_the_global_batch(tmp_path):
    world = 2
    mp.spawn(_neg_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(tmp_path, f"n{r}.pt")) for r in range(world)]
    fu, fi = _towers()
    g = torch.Generator().manual_seed(50)
    xu, xi = torch.randn(world * 8, 6, generator=g), torch.randn(world * 8, 5, generator=g)
    loss = _negatives_loss(fu(xu), fi(xi), world * 8, 1, 0, None)
    loss.backward()
    np.testing.assert_allclose(sum(float(r["loss"]) for r in got) / world, float(loss.detach()), rtol=1e-6)
    for a, b, p in zip(got[0]["grads"], got[1]["grads"], list(fu.parameters()) + list(fi.parameters())):
        assert torch.equal(a, b)  # all-reduced: both ranks hold the same sum
        np.testing.assert_allclose(a.numpy(), p.grad.numpy(), rtol=1e-5, atol=1e-7)

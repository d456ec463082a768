"""Two data-parallel ranks sharing the one GPU of the test box (gloo carrying the HIP tensors, since RCCL refuses two
ranks on one device): the complete N > 1 path -- loss / world, dense bucket all-reduce, all-gather of (indices, gradient
rows), rh_embed_scatter_rows, lazy Adam's touched pass over the gathered indices -- must reproduce ONE process training
on the concatenated global batches (what nn.DataParallel computes, trainers/ctr_trainer.py:53-55), and leave both
replicas equal.  The same is required of row-sharded tables (tables="shard": each rank keeps every second row of every
table; indices all-gathered, rows reduce-scattered), whose reassembled checkpoint must be the single-process one, and
of the two-tower step with sequence features and cross-rank in-batch negatives."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

# noise_tolerant: replicas (and the one-process twin) sum the same gradient rows with float atomics in their own order, and
# Adam turns the rounding noise of near-cancelling elements into lr-sized steps -- compared to a tolerance with an outlier
# budget (tests/test_gpu_models.py::noisy_twin_tolerance states where the budget comes from) and ordered LAST in the suite
pytestmark = [pytest.mark.gpu, pytest.mark.noise_tolerant]
OUTLIERS = 0.08  # share of a tensor's elements allowed beyond atol + rtol |x| (at least 3 elements)

VOCABS = [3, 4, 10, 27, 105, 305, 583 * 40, 40, 1460 * 40, 24, 18, 15, 633 * 40]
STEPS, B = 6, 64
DEVICE = "cuda:0"  # the two gloo ranks share the box's one GPU; the RCCL variant below gives each rank its own


def _data(seed=11):
    g = torch.Generator().manual_seed(seed)
    n = STEPS * 2 * B
    sparse = torch.stack([torch.randint(0, v, (n,), generator=g) for v in VOCABS], 1)
    dense = torch.rand(n, 4, generator=g)
    label = (torch.rand(n, generator=g) < 0.25).float()
    return sparse, dense, label


def _model():
    from torch_rechub_amd.basic.features import DenseFeature, SparseFeature
    from torch_rechub_amd.models.ranking import DeepFM
    torch.manual_seed(3)
    dense = [DenseFeature(f"I{i}") for i in range(4)]
    sparse = [SparseFeature(f"C{i}", v, 16) for i, v in enumerate(VOCABS)]
    # no hidden layer: BatchNorm statistics are per replica by design and would differ from the
    # single-process global-batch statistics this test compares against
    return DeepFM(dense + sparse, sparse, {"dims": [], "dropout": 0.0}).to(DEVICE), dense, sparse


def _train(rows_of_step, world, tables="replicate"):
    from torch_rechub_amd import sharding
    from torch_rechub_amd.trainers import CTRTrainer
    model, dfe, sfe = _model()
    # "shard>=300": only the tables with at least 300 rows are sharded, the small ones stay replicated (their
    # gradient rows go through the all-gather exchange): both mechanisms inside one lookup list
    min_rows = 300 if tables == "shard>=300" else 0
    l2 = tables.endswith("+l2")  # embedding regulariser on: its table gradient is a LOCAL dense term on every replica
    tables = tables.replace("+l2", "")
    placement = "shard" if tables.startswith("shard") else tables
    trainer = CTRTrainer(model, optimizer_params={"lr": 1e-2, "weight_decay": 1e-4, "lazy_small_rows": 64},
                         device=DEVICE, show_progress=False, lazy_k=4, tables=placement if world > 1 else None,
                         shard_min_rows=min_rows,
                         regularization_params={"embedding_l1": 0.0, "embedding_l2": 2e-2, "dense_l1": 0.0,
                                                "dense_l2": 1e-2} if l2 else None)
    assert (trainer.dp is not None) == (world > 1)
    if world > 1 and placement == "shard":
        emb = model.embedding.embed_dict["C5"]
        assert sharding.is_sharded(emb) and emb.weight.shape[0] == -(-305 // world) + 1
        assert sharding.is_sharded(model.embedding.embed_dict["C0"]) == (min_rows == 0)
    sparse, dense, label = _data()
    model.train()
    losses = []
    for s in range(STEPS):
        rows = rows_of_step(s)
        x = {f.name: sparse[rows, j].to(DEVICE) for j, f in enumerate(sfe)}
        x.update({f.name: dense[rows, j].to(DEVICE) for j, f in enumerate(dfe)})
        losses.append(float(trainer.train_step(x, label[rows].to(DEVICE))))
    trainer.flush()
    torch.cuda.synchronize()
    full = sharding.full_state_dict(model) if trainer.tables == "shard" else model.state_dict()
    sd = {k: v.detach().cpu() for k, v in full.items()}
    if trainer.dp is not None:
        trainer.dp.close()
    return sd, losses


def _worker(rank, port, outdir, train, arg, backend="gloo"):
    import torch.distributed as dist
    global DEVICE
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    if backend == "nccl":  # RCCL over xGMI: one device per rank
        DEVICE = f"cuda:{rank}"
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device(DEVICE))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        torch.cuda.set_device(rank if backend == "nccl" else 0)
        sd, losses = train(lambda s: slice(s * 2 * B + rank * B, s * 2 * B + (rank + 1) * B), 2, arg)
        torch.save({"sd": sd, "losses": losses}, os.path.join(outdir, f"rank{rank}.pt"))
        torch.cuda.synchronize()
        dist.barrier()  # neither rank tears its transport down while the other may still be inside a collective
    finally:
        dist.destroy_process_group()
    # The result is on disk; leave without the interpreter's finalisation.  One run in five of round 6's full suite lost rank 0
    # to "terminate called without an active exception" (SIGABRT, a joinable C++ thread destroyed at exit) about where this
    # function returns; the training itself had completed.
    import sys
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)


def _two_ranks(tmp_path, train, arg, backend="gloo"):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(port, str(tmp_path), train, arg, backend), nprocs=2, join=True)
    return torch.load(os.path.join(tmp_path, "rank0.pt")), torch.load(os.path.join(tmp_path, "rank1.pt"))


@pytest.mark.parametrize("tables", ["replicate", "shard"])
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X: the world-2 parity check over real RCCL / xGMI")
def test_two_devices_over_rccl_reproduce_one_process_on_the_global_batch(tmp_path, tables):
    """The same parity check with the production transport: backend nccl (= RCCL), one device per rank -- dense bucket
    all-reduce on the side stream, all-gather of (indices, gradient rows) / index all-gather + row reduce-scatter.
    Skipped on the 1-GPU test box (RCCL refuses two ranks on one device); runs wherever two devices are visible."""
    _check_against_one_process(tmp_path, tables, "nccl")


@pytest.mark.parametrize("tables", ["replicate", "shard", "shard>=300", "replicate+l2", "shard+l2"])
def test_two_ranks_reproduce_one_process_on_the_global_batch(tmp_path, tables):
    _check_against_one_process(tmp_path, tables, "gloo")


def _check_against_one_process(tmp_path, tables, backend):
    r0, r1 = _two_ranks(tmp_path, _train, tables, backend)
    single, losses = _train(lambda s: slice(s * 2 * B, (s + 1) * 2 * B), world=1,
                            tables="replicate+l2" if tables.endswith("+l2") else "replicate")
    assert set(single) == set(r0["sd"]) and all(single[k].shape == r0["sd"][k].shape for k in single)
    # the mean of the two per-rank losses is the global-batch loss (with the regulariser on, a rank's reported loss holds
    # the penalty of ITS shard only in sharded mode: compared on the weights below instead)
    if not tables.startswith("shard+"):
        np.testing.assert_allclose((np.array(r0["losses"]) + np.array(r1["losses"])) / 2, losses, rtol=2e-5, atol=1e-6)
    travel = 1e-2 * STEPS
    for k, want in single.items():
        a, b, w = r0["sd"][k].numpy(), r1["sd"][k].numpy(), want.numpy()
        # replicas: same data, same arithmetic; only the order of the atomic row sums differs
        bad = np.abs(a - b) > 2e-5 + 1e-4 * np.abs(w)
        assert bad.sum() <= max(3, OUTLIERS * bad.size) and np.abs(a - b).max() <= 0.25 * travel, f"replicas diverged in {k}"
        bad = np.abs(a - w) > 3e-4 + 1e-3 * np.abs(w)
        assert bad.sum() <= max(3, OUTLIERS * bad.size), f"{k}: {bad.sum()} / {bad.size} elements differ from single-process training"
        assert np.abs(a - w).max() <= 0.25 * travel + 3e-4, k


def _bucket_after_two_frozen_steps(rows_of_step, world, arg):
    """The dense gradient bucket after the SECOND step of a training whose learning rate and weight decay are zero (so the
    second step sees the first step's parameters, bit for bit, and runs with the bucket attached: at N > 1 the round-5 path --
    slabs packed by the bucket's own launches in the middle of the backward, then all-reduced)."""
    from torch_rechub_amd.trainers import CTRTrainer
    model, dfe, sfe = _model()
    trainer = CTRTrainer(model, optimizer_params={"lr": 0.0, "weight_decay": 0.0, "lazy_small_rows": 64}, device=DEVICE,
                         show_progress=False, lazy_k=4, tables="replicate" if world > 1 else None)
    assert (trainer.dp is not None) == (world > 1)
    sparse, dense, label = _data()
    model.train()
    rows = rows_of_step(0)
    x = {f.name: sparse[rows, j].to(DEVICE) for j, f in enumerate(sfe)}
    x.update({f.name: dense[rows, j].to(DEVICE) for j, f in enumerate(dfe)})
    y = label[rows].to(DEVICE)
    before = {k: v.detach().clone() for k, v in model.state_dict().items() if "embed_dict" not in k}
    for _ in range(2):
        trainer.train_step(x, y)
    torch.cuda.synchronize()
    for k, v in before.items():
        assert torch.equal(model.state_dict()[k], v), k  # frozen, as intended
    assert trainer.optimizer._bucket is not None and all(trainer.bucket.packed)
    flat = trainer.bucket.flat.detach().cpu().clone()
    if trainer.dp is not None:
        trainer.dp.close()
    return {"flat": flat}, [float(arg)]


def test_two_ranks_dense_bucket_is_the_mean_of_the_ranks_single_gpu_gradients_bitwise(tmp_path):
    """Round-5 advisor finding: the world-2 comparisons above carry an 8 % outlier budget and Adam's scale invariance hides a
    wrong 1/world factor, so a partially reduced dense gradient could pass them.  This one has no tolerance: the dense bucket
    of either rank after its all-reduce must equal  0.5 g_0 + 0.5 g_1  bit for bit, g_r = the bucket of a SINGLE-GPU trainer on
    rank r's batch (dense gradients are sums in a fixed order on every path -- per-block partials, split slabs -- and a
    factor of one half is exact)."""
    r0, r1 = _two_ranks(tmp_path, _bucket_after_two_frozen_steps, 0.0)
    singles = [_bucket_after_two_frozen_steps(lambda s, r=r: slice(r * B, (r + 1) * B), 1, 0.0)[0]["flat"] for r in (0, 1)]
    want = 0.5 * singles[0] + 0.5 * singles[1]
    assert want.abs().max() > 0
    assert torch.equal(r0["sd"]["flat"], want)
    assert torch.equal(r1["sd"]["flat"], want)


# -- two towers: sequence features (replicated: gradient rows of the history lookups are exchanged; sharded: pooled
#    partial sums are reduce-scattered) and in-batch negatives drawn over the items of BOTH ranks ---------------------
N_USER, N_ITEM, N_CATE, HIST = 50, 400, 12, 8


def _match_data(seed=23):
    g = torch.Generator().manual_seed(seed)
    n = STEPS * 2 * B
    hist = torch.randint(1, N_ITEM, (n, HIST), generator=g)
    hist[torch.rand(n, HIST, generator=g) < 0.3] = 0  # padding
    return {"user_id": torch.randint(0, N_USER, (n,), generator=g), "hist_item_id": hist,
            "item_id": torch.randint(1, N_ITEM, (n,), generator=g), "cate_id": torch.randint(0, N_CATE, (n,), generator=g)}


def _train_match(rows_of_step, world, arg):
    from torch_rechub_amd import sharding
    from torch_rechub_amd.basic.features import SequenceFeature, SparseFeature
    from torch_rechub_amd.models.matching import DSSM
    from torch_rechub_amd.trainers import MatchTrainer
    from torch_rechub_amd import ops
    tables, pooling = arg
    ops._sample_rng.clear()  # the sampler's call counter starts at 0, as in the freshly spawned ranks
    torch.manual_seed(9)
    user = [SparseFeature("user_id", N_USER, 16),
            SequenceFeature("hist_item_id", N_ITEM, 16, pooling=pooling, shared_with="item_id")]
    item = [SparseFeature("item_id", N_ITEM, 16, padding_idx=0), SparseFeature("cate_id", N_CATE, 16)]
    # no hidden layers: BatchNorm statistics are per replica by design
    model = DSSM(user, item, {"dims": []}, {"dims": []}).to("cuda:0")
    trainer = MatchTrainer(model, mode=0, in_batch_neg=True, in_batch_neg_ratio=5, sampler_seed=77,
                           global_negatives=True, optimizer_params={"lr": 1e-2, "weight_decay": 1e-4,
                                                                    "lazy_small_rows": 16},
                           device="cuda:0", show_progress=False, lazy_k=4, tables=tables if world > 1 else None)
    data = _match_data()
    model.train()
    losses = []
    for s in range(STEPS):
        rows = rows_of_step(s)
        x = {k: v[rows].to("cuda:0") for k, v in data.items()}
        losses.append(float(trainer.train_step(x, torch.zeros(x["user_id"].shape[0], device="cuda:0"))))
    trainer.flush()
    torch.cuda.synchronize()
    full = sharding.full_state_dict(model) if trainer.tables == "shard" else model.state_dict()
    sd = {k: v.detach().cpu() for k, v in full.items()}
    if trainer.dp is not None:
        trainer.dp.close()
    return sd, losses


@pytest.mark.parametrize("tables,pooling", [("replicate", "mean"), ("shard", "mean"), ("shard", "sum")])
def test_two_tower_ranks_with_global_negatives_reproduce_one_process(tmp_path, tables, pooling):
    r0, r1 = _two_ranks(tmp_path, _train_match, (tables, pooling))
    single, losses = _train_match(lambda s: slice(s * 2 * B, (s + 1) * 2 * B), 1, (tables, pooling))
    np.testing.assert_allclose((np.array(r0["losses"]) + np.array(r1["losses"])) / 2, losses, rtol=2e-5, atol=1e-6)
    travel = 1e-2 * STEPS
    for k, want in single.items():
        a, b, w = r0["sd"][k].numpy(), r1["sd"][k].numpy(), want.numpy()
        assert a.shape == w.shape, k
        bad = np.abs(a - b) > 2e-5 + 1e-4 * np.abs(w)
        assert bad.sum() <= max(3, OUTLIERS * bad.size) and np.abs(a - b).max() <= 0.25 * travel, f"replicas diverged in {k}"
        bad = np.abs(a - w) > 3e-4 + 1e-3 * np.abs(w)
        assert bad.sum() <= max(3, OUTLIERS * bad.size), f"{k}: {bad.sum()} / {bad.size} elements differ from single-process training"
        assert np.abs(a - w).max() <= 0.25 * travel + 3e-4, k

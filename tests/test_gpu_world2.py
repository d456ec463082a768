"""Two data-parallel ranks sharing the one GPU of the test box (gloo carrying the HIP tensors, since RCCL refuses two
ranks on one device): the complete N > 1 path -- loss / world, dense bucket all-reduce, all-gather of (indices, gradient
rows), rh_embed_scatter_rows, lazy Adam's touched pass over the gathered indices -- must reproduce ONE process training
on the concatenated global batches (what nn.DataParallel computes, trainers/ctr_trainer.py:53-55), and leave both
replicas equal."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

VOCABS = [3, 4, 10, 27, 105, 305, 583 * 40, 40, 1460 * 40, 24, 18, 15, 633 * 40]
STEPS, B = 6, 64


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
    return DeepFM(dense + sparse, sparse, {"dims": [], "dropout": 0.0}).to("cuda:0"), dense, sparse


def _train(rows_of_step, world):
    from torch_rechub_amd.trainers import CTRTrainer
    model, dfe, sfe = _model()
    trainer = CTRTrainer(model, optimizer_params={"lr": 1e-2, "weight_decay": 1e-4, "lazy_small_rows": 64},
                         device="cuda:0", show_progress=False, lazy_k=4)
    assert (trainer.dp is not None) == (world > 1)
    sparse, dense, label = _data()
    model.train()
    losses = []
    for s in range(STEPS):
        rows = rows_of_step(s)
        x = {f.name: sparse[rows, j].to("cuda:0") for j, f in enumerate(sfe)}
        x.update({f.name: dense[rows, j].to("cuda:0") for j, f in enumerate(dfe)})
        losses.append(float(trainer.train_step(x, label[rows].to("cuda:0"))))
    trainer.flush()
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    if trainer.dp is not None:
        trainer.dp.close()
    return sd, losses


def _worker(rank, port, outdir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        torch.cuda.set_device(0)
        sd, losses = _train(lambda s: slice(s * 2 * B + rank * B, s * 2 * B + (rank + 1) * B), world=2)
        torch.save({"sd": sd, "losses": losses}, os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_ranks_reproduce_one_process_on_the_global_batch(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    single, losses = _train(lambda s: slice(s * 2 * B, (s + 1) * 2 * B), world=1)
    # the mean of the two per-rank losses is the global-batch loss
    np.testing.assert_allclose((np.array(r0["losses"]) + np.array(r1["losses"])) / 2, losses, rtol=2e-5, atol=1e-6)
    travel = 1e-2 * STEPS
    for k, want in single.items():
        a, b, w = r0["sd"][k].numpy(), r1["sd"][k].numpy(), want.numpy()
        # replicas: same data, same arithmetic; only the order of the atomic row sums differs
        bad = np.abs(a - b) > 2e-5 + 1e-4 * np.abs(w)
        assert bad.mean() <= 5e-3 and np.abs(a - b).max() <= 0.25 * travel, f"replicas diverged in {k}"
        bad = np.abs(a - w) > 3e-4 + 1e-3 * np.abs(w)
        assert bad.mean() <= 5e-3, f"{k}: {bad.sum()} / {bad.size} elements differ from single-process training"
        assert np.abs(a - w).max() <= 0.25 * travel + 3e-4, k

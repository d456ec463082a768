"""CPU port (plain eager PyTorch, fp32) of the reference's DeepFM / DCN training step — the ``cpu_baseline`` that
``bench.py`` times on the GPU box's host cores, where /root/reference does not exist.

TEST / BASELINE INFRASTRUCTURE ONLY: never imported by ``torch_rechub_amd``.

It restates the reference's op chain one-for-one, so that its cost profile is the reference's:
  * one ``nn.Embedding`` per sparse feature, looked up per feature and concatenated
    (EmbeddingLayer.forward, torch_rechub/basic/layers.py:77-127) — and DeepFM gathers TWICE
    (models/ranking/deepfm.py:35,37);
  * FM = 2 sums + pow + sub + sum + mul (layers.py:313-319); LR = nn.Linear (layers.py:183-189);
  * MLP = [Linear, BatchNorm1d, ReLU, Dropout] x k + Linear (layers.py:276-292);
  * CrossNetwork loop (layers.py:412-420);
  * step = BCELoss on probabilities, model.zero_grad(), backward (dense embedding gradients), torch.optim.Adam over
    every parameter incl. all table rows (trainers/ctr_trainer.py:59-61, 87-99), loss.item() twice per step.
  * end-to-end leg: the reference's input contract -- ``TorchDataset.__getitem__`` builds one dict PER SAMPLE,
    ``DataGenerator.generate_dataloader`` does ``random_split`` + ``DataLoader(shuffle=True, num_workers=0)``
    (utils/data.py:14-25, 61-83), the loop moves every column to the device and calls ``y.float()``
    (trainers/ctr_trainer.py:83-85) -- SURVEY 8(d) "the reference CPU CTRTrainer".
Pinned by tests/test_oracle_golden.py::test_cpu_port_* against the reference's golden vectors.
"""
import time

import numpy as np
import torch
from torch import nn
from torch.utils.data import DataLoader, Dataset, random_split


class PortMLP(nn.Module):

    def __init__(self, input_dim, dims, dropout=0.0, output_layer=True):
        super().__init__()
        layers = []
        for w in dims:
            layers += [nn.Linear(input_dim, w), nn.BatchNorm1d(w), nn.ReLU(inplace=True), nn.Dropout(p=dropout)]
            input_dim = w
        if output_layer:
            layers.append(nn.Linear(input_dim, 1))
        self.mlp = nn.Sequential(*layers)

    def forward(self, x):
        return self.mlp(x)


class PortEmbedding(nn.Module):
    """Per-feature tables in a ModuleDict; forward(x, names, dense_names, squeeze) as layers.py:77-127."""

    def __init__(self, vocabs, embed_dim, init_std=1e-4):
        super().__init__()
        self.embed_dict = nn.ModuleDict()
        for name, v in vocabs.items():
            e = nn.Embedding(v, embed_dim)
            nn.init.normal_(e.weight, 0.0, init_std)  # RandomNormal(0, 1e-4), features.py:54
            self.embed_dict[name] = e

    def forward(self, x, sparse_names, dense_names=(), squeeze_dim=False):
        sparse_emb = [self.embed_dict[n](x[n].long()).unsqueeze(1) for n in sparse_names]
        dense_values = [x[n].float().unsqueeze(1) for n in dense_names]
        emb = torch.cat(sparse_emb, dim=1) if sparse_emb else None
        if not squeeze_dim:
            return emb
        if emb is None:
            return torch.cat(dense_values, dim=1)
        if dense_values:
            return torch.cat((emb.flatten(start_dim=1), torch.cat(dense_values, dim=1)), dim=1)
        return emb.flatten(start_dim=1)


def fm_port(x):
    square_of_sum = torch.sum(x, dim=1)**2
    sum_of_square = torch.sum(x**2, dim=1)
    return 0.5 * torch.sum(square_of_sum - sum_of_square, dim=1, keepdim=True)


class PortDeepFM(nn.Module):
    """tutorials/00 wiring: deep = dense + sparse, fm = sparse (parameter names as the reference's state_dict)."""

    def __init__(self, vocabs, dense_names, embed_dim=16, dims=(256, 128), dropout=0.2, deep_uses_sparse=True,
                 init_std=1e-4):
        super().__init__()
        self.sparse_names = list(vocabs)
        self.dense_names = list(dense_names)
        self.deep_uses_sparse = deep_uses_sparse
        fm_dims = len(self.sparse_names) * embed_dim
        deep_dims = len(self.dense_names) + (fm_dims if deep_uses_sparse else 0)

        class _LR(nn.Module):

            def __init__(self, n):
                super().__init__()
                self.fc = nn.Linear(n, 1, bias=True)

            def forward(self, x):
                return self.fc(x)

        self.linear = _LR(fm_dims)
        self.embedding = PortEmbedding(vocabs, embed_dim, init_std)
        self.mlp = PortMLP(deep_dims, list(dims), dropout)

    def forward(self, x):
        deep_sparse = self.sparse_names if self.deep_uses_sparse else []
        input_deep = self.embedding(x, deep_sparse, self.dense_names, squeeze_dim=True)  # gather #1
        input_fm = self.embedding(x, self.sparse_names, (), squeeze_dim=False)  # gather #2 (deepfm.py:37)
        y = self.linear(input_fm.flatten(start_dim=1)) + fm_port(input_fm) + self.mlp(input_deep)
        return torch.sigmoid(y.squeeze(1))


def train_step(model, optimizer, criterion, x, y):
    """Body of CTRTrainer.train_one_epoch (ctr_trainer.py:84-101), regularisation off (python 0.0)."""
    y_pred = model(x)
    loss = criterion(y_pred, y.float())
    model.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.item() + 0.0 * loss.item()  # the reference syncs twice per step (:100-101)


def time_cpu_baseline(vocabs, n_dense, batch_size, budget_s=20.0, min_steps=3, max_steps=50, seed=2022, threads=None):
    """Times model-step-only training (pre-collated batches) of the DeepFM port at the given shape.

    Returns dict(samples_per_s, steps, ms_per_step, cores, build_s).  Roughly ``budget_s`` seconds of steps.
    """
    if threads:
        torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(seed)
    t0 = time.perf_counter()
    names = {f"C{i + 1}": int(v) for i, v in enumerate(vocabs)}
    dense_names = [f"I{i + 1}" for i in range(n_dense)]
    model = PortDeepFM(names, dense_names)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)  # trainer default, ctr_trainer.py:60
    crit = nn.BCELoss()
    build_s = time.perf_counter() - t0

    def batch():
        x = {n: torch.randint(0, v, (batch_size,), generator=g) for n, v in names.items()}
        x.update({n: torch.rand(batch_size, generator=g) for n in dense_names})
        return x, (torch.rand(batch_size, generator=g) < 0.25).long()

    batches = [batch() for _ in range(4)]
    train_step(model, opt, crit, *batches[0])  # warm-up: allocates the dense gradients and Adam state
    steps, t_start = 0, time.perf_counter()
    while steps < max_steps:
        train_step(model, opt, crit, *batches[steps % len(batches)])
        steps += 1
        if steps >= min_steps and time.perf_counter() - t_start > budget_s:
            break
    dt = time.perf_counter() - t_start
    return dict(samples_per_s=steps * batch_size / dt, steps=steps, ms_per_step=1e3 * dt / steps,
                cores=torch.get_num_threads(), build_s=build_s)


class PortDataset(Dataset):
    """TorchDataset (utils/data.py:14-25): x = dict of column arrays, one python dict per sample."""

    def __init__(self, x, y):
        self.x, self.y = x, y

    def __getitem__(self, index):
        return {k: v[index] for k, v in self.x.items()}, self.y[index]

    def __len__(self):
        return len(self.y)


def port_dataloaders(x, y, split_ratio, batch_size):
    """DataGenerator.generate_dataloader with split_ratio (utils/data.py:61-83)."""
    ds = PortDataset(x, y)
    n = len(ds)
    n_train, n_val = int(n * split_ratio[0]), int(n * split_ratio[1])
    train, val, test = random_split(ds, (n_train, n_val, n - n_train - n_val))
    return (DataLoader(train, batch_size=batch_size, shuffle=True, num_workers=0),
            DataLoader(val, batch_size=batch_size, shuffle=False, num_workers=0),
            DataLoader(test, batch_size=batch_size, shuffle=False, num_workers=0))


def time_cpu_end_to_end(vocabs, n_dense, batch_size, rows=200_000, budget_s=12.0, min_steps=2, max_steps=50, seed=2022,
                        threads=None):
    """Times the reference's train_one_epoch loop END TO END (loader included) on a ``rows``-row slice given as a dict
    of numpy columns (int64 sparse as its LabelEncoder emits, float32 dense, int64 labels; tutorial-00 usage).

    Returns dict(samples_per_s, steps, ms_per_step, loader_ms_per_step, cores, rows)."""
    if threads:
        torch.set_num_threads(threads)
    rng = np.random.default_rng(seed)
    names = {f"C{i + 1}": int(v) for i, v in enumerate(vocabs)}
    dense_names = [f"I{i + 1}" for i in range(n_dense)]
    x = {n: rng.integers(0, v, rows, dtype=np.int64) for n, v in names.items()}
    x.update({n: rng.random(rows, dtype=np.float32) for n in dense_names})
    y = (rng.random(rows) < 0.25).astype(np.int64)
    torch.manual_seed(seed)
    train_dl, _, _ = port_dataloaders(x, y, [0.7, 0.1], batch_size)
    model = PortDeepFM(names, dense_names)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)  # trainer default, ctr_trainer.py:60
    crit = nn.BCELoss()
    device = torch.device("cpu")
    it = iter(train_dl)
    xb, yb = next(it)
    train_step(model, opt, crit, {k: v.to(device) for k, v in xb.items()}, yb.to(device))  # warm-up: Adam state
    steps, loader_s, t_start = 0, 0.0, time.perf_counter()
    while steps < max_steps:
        t0 = time.perf_counter()
        try:
            xb, yb = next(it)
        except StopIteration:
            it = iter(train_dl)
            xb, yb = next(it)
        xb = {k: v.to(device) for k, v in xb.items()}  # ctr_trainer.py:84
        yb = yb.to(device)
        loader_s += time.perf_counter() - t0
        train_step(model, opt, crit, xb, yb)
        steps += 1
        if steps >= min_steps and time.perf_counter() - t_start > budget_s:
            break
    dt = time.perf_counter() - t_start
    return dict(samples_per_s=steps * batch_size / dt, steps=steps, ms_per_step=1e3 * dt / steps,
                loader_ms_per_step=1e3 * loader_s / steps, cores=torch.get_num_threads(), rows=rows)


def time_cpu_legs(vocabs, n_dense, batch_size, budget_s=24.0, full=False, rows=200_000, seed=2022, threads=None,
                  model_step_leg=True):
    """Both legs of SURVEY 8(d)'s CPU baseline on ONE model / optimizer (the 8.4 GB of p / g / m / v are built once):
    (i) the reference's ``train_one_epoch`` loop END TO END over a DataGenerator-style loader (dict of numpy columns),
    (ii) model-step-only on pre-collated batches.  Every step is timed on its own; the MEDIAN is reported.

    ``full``: the protocol as written (3 warm-up + 10 timed steps per leg; ~90 s at the Criteo shape on 128 threads);
    ``"auto"``: the full protocol unless the first step shows it would take longer than ~8x ``budget_s``.
    Otherwise the run is bounded to about ``budget_s`` seconds of steps: 1 warm-up step (it allocates the dense
    gradients and the Adam state) and as many timed steps per leg as fit, at least 3 -- the record says which.
    ``model_step_leg=False`` (bench.py's default run since round 6: ONE 3 + 10 leg, ~45 s instead of ~100 s of a 2-minute
    run): leg (ii) is skipped and the model step is taken as end-to-end minus loader, per step."""
    if threads:
        torch.set_num_threads(threads)
    rng = np.random.default_rng(seed)
    names = {f"C{i + 1}": int(v) for i, v in enumerate(vocabs)}
    dense_names = [f"I{i + 1}" for i in range(n_dense)]
    x = {n: rng.integers(0, v, rows, dtype=np.int64) for n, v in names.items()}
    x.update({n: rng.random(rows, dtype=np.float32) for n in dense_names})
    y = (rng.random(rows) < 0.25).astype(np.int64)
    torch.manual_seed(seed)
    t0 = time.perf_counter()
    train_dl, _, _ = port_dataloaders(x, y, [0.7, 0.1], batch_size)
    model = PortDeepFM(names, dense_names)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)  # trainer default, ctr_trainer.py:60
    crit = nn.BCELoss()
    build_s = time.perf_counter() - t0
    device = torch.device("cpu")
    it = iter(train_dl)

    def next_batch():
        nonlocal it
        try:
            xb, yb = next(it)
        except StopIteration:
            it = iter(train_dl)
            xb, yb = next(it)
        return {k: v.to(device) for k, v in xb.items()}, yb.to(device)  # ctr_trainer.py:84-85

    t0 = time.perf_counter()
    train_step(model, opt, crit, *next_batch())  # first warm-up step: allocates the dense gradients and the Adam state
    first_s = time.perf_counter() - t0
    if full == "auto":  # the full protocol is 25 more steps: only where a step is short enough for the run to stay bounded
        full = first_s * 2 * 13 <= max(budget_s, 1.0) * 8
    n_warm, n_timed, n_min = (3, 10, 10) if full else (1, 10, 3)
    for _ in range(n_warm - 1):
        train_step(model, opt, crit, *next_batch())
    leg_budget = budget_s / 2
    e2e_ms, loader_ms, t_leg = [], [], time.perf_counter()
    while len(e2e_ms) < n_timed:
        t0 = time.perf_counter()
        xb, yb = next_batch()
        t1 = time.perf_counter()
        train_step(model, opt, crit, xb, yb)
        t2 = time.perf_counter()
        e2e_ms.append(1e3 * (t2 - t0))
        loader_ms.append(1e3 * (t1 - t0))
        if not full and len(e2e_ms) >= n_min and time.perf_counter() - t_leg > leg_budget:
            break
    g = torch.Generator().manual_seed(seed)

    def batch():
        xb = {n: torch.randint(0, v, (batch_size,), generator=g) for n, v in names.items()}
        xb.update({n: torch.rand(batch_size, generator=g) for n in dense_names})
        return xb, (torch.rand(batch_size, generator=g) < 0.25).long()

    batches = [batch() for _ in range(4)] if model_step_leg else []
    for i in range(n_warm if (full and model_step_leg) else 0):  # the model and the optimizer state are warm from leg (i) already
        train_step(model, opt, crit, *batches[i % 4])
    step_ms, t_leg = [], time.perf_counter()
    if not model_step_leg:
        step_ms = [a - b for a, b in zip(e2e_ms, loader_ms)]
    while model_step_leg and len(step_ms) < n_timed:
        t0 = time.perf_counter()
        train_step(model, opt, crit, *batches[len(step_ms) % 4])
        step_ms.append(1e3 * (time.perf_counter() - t0))
        if not full and len(step_ms) >= n_min and time.perf_counter() - t_leg > leg_budget:
            break

    def leg(ms, warm):
        med = float(np.median(ms))
        return dict(median_ms_per_step=med, samples_per_s=batch_size / (med * 1e-3), timed_steps=len(ms), warmup_steps=warm,
                    step_ms=[round(v, 1) for v in ms])

    e2e = leg(e2e_ms, n_warm)
    e2e["loader_median_ms_per_step"] = float(np.median(loader_ms))
    # SURVEY 8(d), "once as DataFrame, as tutorial 00 does": x handed to DataGenerator as a pandas DataFrame (y a Series).
    # TorchDataset.__getitem__ (utils/data.py:21-22) then indexes 39 Series per SAMPLE; the model step behind it is the one
    # timed above, so this leg times the LOADER alone (n_df batches, at least 1 warm-up) and adds the model-step median.
    import pandas as pd
    df = pd.DataFrame(x)
    df_dl, _, _ = port_dataloaders(df, pd.Series(y), [0.7, 0.1], batch_size)
    it_df = iter(df_dl)
    next(it_df)
    df_ms = []
    for _ in range(3):
        t0 = time.perf_counter()
        xb, yb = next(it_df)
        {k: v.to(device) for k, v in xb.items()}, yb.to(device)
        df_ms.append(1e3 * (time.perf_counter() - t0))
    step_med = float(np.median(step_ms))
    df_leg = dict(loader_median_ms_per_step=float(np.median(df_ms)), loader_ms=[round(v, 1) for v in df_ms],
                  median_ms_per_step=float(np.median(df_ms)) + step_med,
                  samples_per_s=batch_size / ((float(np.median(df_ms)) + step_med) * 1e-3),
                  note="loader timed alone (1 warm-up + 3 batches), model-step median of this run added")
    return dict(end_to_end=e2e, end_to_end_dataframe=df_leg,
                model_step=leg(step_ms, n_warm + len(e2e_ms) + (n_warm if full else 0)) if model_step_leg else None,
                cores=torch.get_num_threads(), rows=rows, build_s=build_s, full_protocol=bool(full))

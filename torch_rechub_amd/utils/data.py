"""Input pipeline.

Two loaders with the same batch contract ``(x_dict, y)`` as the reference:

* ``TorchDataset`` / ``PredictDataset`` / ``DataGenerator`` — API mirrors of
  torch_rechub/utils/data.py:14-25, 28-38, 61-83 (host DataLoader: per-sample dict, default_collate).
  The reference's end-to-end CPU run is bound by this loader (3-20 k samples/s).
* ``DeviceDataLoader`` — the columnar dataset is resident in HBM; one HIP launch (``rh_batch_gather``)
  assembles each shuffled minibatch into STATIC buffers, so there is no per-sample Python, no
  host->device copy in the step and the whole train step can be captured in a hipGraph.
  ``x_dict`` values are column views of the static (B,F) / (B,ND) buffers.
"""

import torch
from torch.utils.data import DataLoader, Dataset, random_split

from .. import _lib, ops


def get_loss_func(task_type="classification"):
    """Default criterion of a task (reference utils/data.py:104-110): BCELoss on probabilities / MSELoss."""
    if task_type == "classification":
        return torch.nn.BCELoss()
    if task_type == "regression":
        return torch.nn.MSELoss()
    raise ValueError("task_type must be classification or regression")


def get_metric_func(task_type="classification"):
    """Default validation metric of a task (reference utils/data.py:113-119): ROC AUC / mean squared error."""
    from sklearn.metrics import mean_squared_error, roc_auc_score
    if task_type == "classification":
        return roc_auc_score
    if task_type == "regression":
        return mean_squared_error
    raise ValueError("task_type must be classification or regression")


class TorchDataset(Dataset):

    def __init__(self, x, y):
        super().__init__()
        self.x = x
        self.y = y

    def __getitem__(self, index):
        return {k: v[index] for k, v in self.x.items()}, self.y[index]

    def __len__(self):
        return len(self.y)


class PredictDataset(Dataset):

    def __init__(self, x):
        super().__init__()
        self.x = x

    def __getitem__(self, index):
        return {k: v[index] for k, v in self.x.items()}

    def __len__(self):
        return len(self.x[list(self.x.keys())[0]])


class MatchDataGenerator(object):
    """Loaders for two-tower training / retrieval evaluation (API mirror of torch_rechub/utils/data.py:41-58)."""

    def __init__(self, x, y=[]):
        super().__init__()
        self.dataset = TorchDataset(x, y) if len(y) != 0 else PredictDataset(x)

    def generate_dataloader(self, x_test_user, x_all_item, batch_size, num_workers=8):
        train_dataloader = DataLoader(self.dataset, batch_size=batch_size, shuffle=True, num_workers=num_workers)
        test_dataloader = DataLoader(PredictDataset(x_test_user), batch_size=batch_size, shuffle=False,
                                     num_workers=num_workers)
        item_dataloader = DataLoader(PredictDataset(x_all_item), batch_size=batch_size, shuffle=False,
                                     num_workers=num_workers)
        return train_dataloader, test_dataloader, item_dataloader


class DataGenerator(object):
    """Host loader factory with the reference's signature and split semantics (utils/data.py:61-83)."""

    def __init__(self, x, y):
        super().__init__()
        self.dataset = TorchDataset(x, y)
        self.length = len(self.dataset)

    def generate_dataloader(self, x_val=None, y_val=None, x_test=None, y_test=None, split_ratio=None, batch_size=16,
                            num_workers=0):
        if split_ratio is not None:
            train_length = int(self.length * split_ratio[0])
            val_length = int(self.length * split_ratio[1])
            test_length = self.length - train_length - val_length
            print("the samples of train : val : test are  %d : %d : %d" % (train_length, val_length, test_length))
            train_dataset, val_dataset, test_dataset = random_split(self.dataset,
                                                                    (train_length, val_length, test_length))
        else:
            train_dataset = self.dataset
            val_dataset = TorchDataset(x_val, y_val)
            test_dataset = TorchDataset(x_test, y_test)
        train_dataloader = DataLoader(train_dataset, batch_size=batch_size, shuffle=True, num_workers=num_workers)
        val_dataloader = DataLoader(val_dataset, batch_size=batch_size, shuffle=False, num_workers=num_workers)
        test_dataloader = DataLoader(test_dataset, batch_size=batch_size, shuffle=False, num_workers=num_workers)
        return train_dataloader, val_dataloader, test_dataloader


class DeviceBatch(dict):
    """``x_dict`` whose sparse values are columns of one packed (B, F) index matrix (``.sparse``)."""
    sparse = None
    sparse_names = ()
    dense = None
    dense_names = ()


class DeviceDataLoader(object):
    """HBM-resident columnar dataset + on-device shuffled batch assembly.

    sparse : (N, F) int64 on the device, columns named ``sparse_names``.  An entry of ``sparse_names`` may be a
             ``(name, L)`` pair: the next L columns are one padded sequence feature (what ``generate_seq_feature`` /
             ``pad_sequences`` produce per row, utils/data.py:122-214); the batch then carries it as a contiguous
             (B, L) index matrix.
    dense  : (N, ND) float32 on the device (or None), columns named ``dense_names``
    label  : (N,) float32 on the device, or (N, n_task) for the multi-task trainer (``ys[:, i]`` = task i)
    Iterating yields ``(DeviceBatch, y)`` whose tensors are the same static buffers every step.  With
    ``drop_last=False`` (reference DataLoader default) the tail batch is smaller and uses its own buffers.
    """

    def __init__(self, sparse, sparse_names, dense, dense_names, label, batch_size, shuffle=True, drop_last=False,
                 generator=None):
        ops.require_hip(sparse, dense, label)
        if sparse.dtype != torch.int64 or sparse.dim() != 2 or not sparse.is_contiguous():
            raise ValueError("sparse must be a contiguous int64 (N, F) matrix")
        if dense is not None and (dense.dtype != torch.float32 or dense.dim() != 2 or not dense.is_contiguous()):
            raise ValueError("dense must be a contiguous float32 (N, ND) matrix")
        self.sparse_names, self.dense_names = list(sparse_names), list(dense_names or [])
        self.N, self.F = sparse.shape
        self.ND = 0 if dense is None else dense.shape[1]
        # column layout of the sparse block: (name, first column, width); width > 1 = a padded sequence feature
        self.columns, at = [], 0
        for entry in self.sparse_names:
            name, width = (entry, 1) if isinstance(entry, str) else (entry[0], int(entry[1]))
            self.columns.append((name, at, width))
            at += width
        if at != self.F:
            raise ValueError(f"sparse has {self.F} columns, sparse_names describe {at}")
        label = label.float()
        self.n_label = 1 if label.dim() == 1 else int(label.shape[1])
        if label.dim() == 2:
            # multi-task labels ride behind the dense block: one gather moves features and every task's label
            dense = label.contiguous() if dense is None else torch.cat([dense, label], dim=1).contiguous()
            label = None
        elif label.dim() != 1:
            raise ValueError("label must be (N,) or (N, n_task)")
        self.sparse, self.dense, self.label = sparse, dense, None if label is None else label.contiguous()
        self.NDL = 0 if dense is None else dense.shape[1]  # dense columns + label columns riding with them
        self.batch_size = int(batch_size)
        self.shuffle, self.drop_last = shuffle, drop_last
        self.generator = generator
        dev = sparse.device
        self.pos = torch.zeros(1, dtype=torch.int64, device=dev)
        self.perm = torch.arange(self.N, dtype=torch.int64, device=dev)
        self._bufs = {}
        # bumped whenever perm / pos change other than by the training step's own fused advance (pos += B): a consumer that
        # looked one batch ahead (optim.TableAdam's relaxed join) then knows its preview is void
        self.generation = 0

    @classmethod
    def from_parquet(cls, file_paths, sparse_names, dense_names, label_name, batch_size, device="cuda:0", shuffle=True,
                     drop_last=False, generator=None, rank=None, world=None, chunk_rows=1 << 20):
        """Stream Parquet files straight into the HBM-resident columnar dataset.

        The files are partitioned over the ranks of the data-parallel job with the reference's per-worker rule
        (data/dataset.py:88-107: contiguous runs of ceil(n / parts) files); this rank's run is scanned once with
        pyarrow in record batches of ``chunk_rows`` rows, each packed on the host into one (rows, F) int64 and one
        (rows, ND) float32 block in pinned double buffers and copied to its slice of the device tensors on a side
        stream while the next record batch is being decoded.  Integer columns stay exact (int64), unlike the
        reference's float32 cast.
        """
        import numpy as np
        import pyarrow.dataset as pads
        import pyarrow.parquet as pq

        import torch.distributed as dist

        from ..data.convert import pa_column_to_numpy
        from ..data.dataset import partition_files
        if world is None:
            world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
            rank = dist.get_rank() if world > 1 else 0
        mine = partition_files(file_paths, world, rank or 0)
        sparse_names, dense_names = list(sparse_names), list(dense_names or [])
        n_rows = sum(pq.ParquetFile(p).metadata.num_rows for p in mine)
        dev = torch.device(device)
        if world > 1 and dist.is_available() and dist.is_initialized() and dist.get_world_size() == world:
            # every training step is a collective: all ranks must run the same number of equally sized batches.  Each
            # rank keeps the first min_r(rows_r) rows of its run of files (the tail of the longer runs is dropped, as a
            # drop_last over ranks), and a rank without files makes EVERY rank raise instead of leaving its peers
            # blocked in the first collective.
            t = torch.tensor([n_rows], dtype=torch.int64, device=dev if dist.get_backend() != "gloo" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            n_rows = int(t.item())
            if n_rows == 0:
                raise ValueError(f"a rank of {world} received no Parquet rows ({len(file_paths)} files in total): "
                                 "give the job at least one non-empty file per rank")
        elif not mine:
            raise ValueError(f"rank {rank} of {world} received no Parquet file ({len(file_paths)} files in total)")
        F, ND = len(sparse_names), len(dense_names)
        sparse = torch.empty((n_rows, F), dtype=torch.int64, device=dev)
        dense = torch.empty((n_rows, ND), dtype=torch.float32, device=dev) if ND else None
        label = torch.empty((n_rows,), dtype=torch.float32, device=dev)
        from .. import graphs
        copy_stream = graphs.role_stream("copy", dev)
        stage = [None, None]  # pinned (sparse, dense, label) staging blocks, used alternately
        done = [None, None]
        columns = sparse_names + dense_names + [label_name]
        scanner = pads.dataset(list(mine), format="parquet").scanner(columns=columns, batch_size=int(chunk_rows))
        at = 0
        for i, rb in enumerate(scanner.to_batches()):
            n = min(rb.num_rows, n_rows - at)  # n_rows may have been truncated to the minimum over the ranks
            if n <= 0:
                if at >= n_rows:
                    break
                continue
            if n < rb.num_rows:
                rb = rb.slice(0, n)
            k = i & 1
            if done[k] is not None:
                done[k].synchronize()  # the copy that last used this staging block has finished
            if stage[k] is None or stage[k][0].shape[0] < n:
                stage[k] = (torch.empty((max(n, int(chunk_rows)), F), dtype=torch.int64).pin_memory(),
                            torch.empty((max(n, int(chunk_rows)), max(ND, 1)), dtype=torch.float32).pin_memory(),
                            torch.empty((max(n, int(chunk_rows)),), dtype=torch.float32).pin_memory())
            hs, hd, hl = stage[k]
            cols = {name: rb.column(j) for j, name in enumerate(rb.schema.names)}
            hs_np, hd_np = hs.numpy(), hd.numpy()
            for j, name in enumerate(sparse_names):
                np.copyto(hs_np[:n, j], pa_column_to_numpy(cols[name], integer=True))
            for j, name in enumerate(dense_names):
                np.copyto(hd_np[:n, j], pa_column_to_numpy(cols[name], integer=False))
            np.copyto(hl.numpy()[:n], pa_column_to_numpy(cols[label_name], integer=False))
            with torch.cuda.stream(copy_stream):
                sparse[at:at + n].copy_(hs[:n], non_blocking=True)
                if ND:
                    dense[at:at + n].copy_(hd[:n, :ND], non_blocking=True)
                label[at:at + n].copy_(hl[:n], non_blocking=True)
                done[k] = torch.cuda.Event()
                done[k].record(copy_stream)
            at += n
        if at != n_rows:
            raise RuntimeError(f"Parquet metadata announced {n_rows} rows, the scan delivered {at}")
        torch.cuda.current_stream(dev).wait_stream(copy_stream)
        return cls(sparse, sparse_names, dense, dense_names, label, batch_size, shuffle=shuffle, drop_last=drop_last,
                   generator=generator)

    def __len__(self):
        full, rem = divmod(self.N, self.batch_size)
        return full + (1 if rem and not self.drop_last else 0)

    def _buffers(self, B):
        b = self._bufs.get(B)
        if b is None:
            dev = self.sparse.device
            sp = torch.empty((B, self.F), dtype=torch.int64, device=dev)
            sp._rh_static = True  # same address, same role every step: per-site caches may key on it (sharding.py)
            de = torch.empty((B, self.NDL), dtype=torch.float32, device=dev) if self.NDL else None
            y = torch.empty((B,), dtype=torch.float32, device=dev) if self.label is not None else de[:, self.ND:]
            x = DeviceBatch()
            seqs = []
            for name, at, width in self.columns:
                if width == 1:
                    x[name] = sp[:, at]
                else:  # a sequence feature gets its own contiguous (B, L) buffer (the kernels' descriptors are by address)
                    x[name] = torch.empty((B, width), dtype=torch.int64, device=dev)
                    x[name]._rh_static = True
                    seqs.append((x[name], sp[:, at:at + width]))
            for j, n in enumerate(self.dense_names):
                x[n] = de[:, j]
            x.sparse, x.sparse_names, x.dense, x.dense_names = sp, [c[0] for c in self.columns], de, self.dense_names
            b = (x, y, sp, de, seqs)
            self._bufs[B] = b
        return b

    def reshuffle(self):
        if self.shuffle:  # in place: a captured hipGraph keeps reading the same buffer
            self.perm.copy_(torch.randperm(self.N, dtype=torch.int64, device=self.sparse.device,
                                           generator=self.generator))
        self.pos.zero_()
        self.generation += 1

    def assembly_args(self, B=None):
        """The arguments of rh_batch_gather for the batch at the current position -- for a caller that runs the assembly
        inside a launch of its own (optim.TableAdam: rh_adam_lazy_refresh_assemble) after ``load_next(assemble=False)`` --
        or None when this loader's batches need more than that one gather (padded-sequence columns)."""
        B = self.batch_size if B is None else B
        x, y, sp, de, seqs = self._buffers(B)
        if seqs:
            return None
        return dict(perm=self.perm, pos=self.pos, N=self.N, B=B, sparse=self.sparse, F=self.F, dense=self.dense, ND=self.NDL,
                    label=self.label, sparse_out=sp, dense_out=de, label_out=y if self.label is not None else None)

    def load_next(self, B=None, advance=True, assemble=True):
        """Assemble the batch at the current position into the static buffers and advance the position.

        ``advance=False``: the caller advances the position later in the step (``counter()`` -> ops.StepFusion: the
        trainers fold it into the step's single scalar launch instead of a launch of its own).
        ``assemble=False``: only the views are returned; the caller launches the assembly itself (``assembly_args``)."""
        B = self.batch_size if B is None else B
        x, y, sp, de, seqs = self._buffers(B)
        s = ops._stream()
        if assemble:
            self.generation += 1  # the static batch buffers are overwritten: a batch somebody prepared ahead is gone
            _lib.call("rh_batch_gather", ops._p(self.perm), ops._p(self.pos), self.N, B, ops._p(self.sparse), self.F,
                      ops._p(self.dense), self.NDL, ops._p(self.label), ops._p(sp), ops._p(de),
                      ops._p(y if self.label is not None else None), s)
        elif seqs:
            raise ValueError("load_next(assemble=False) on a loader with sequence columns")
        if advance:
            _lib.call("rh_batch_advance", ops._p(self.pos), B, self.N, s)
            self.generation += 1
        for dst, src in seqs:
            dst.copy_(src)
        return x, y

    def counter(self, B=None):
        """(device counter, increment, modulus) of the batch position: pos = (pos + B) % N."""
        return self.pos, int(self.batch_size if B is None else B), int(self.N)

    def __iter__(self):
        self.reshuffle()
        full, rem = divmod(self.N, self.batch_size)
        for _ in range(full):
            yield self.load_next()
        if rem and not self.drop_last:
            yield self.load_next(rem)

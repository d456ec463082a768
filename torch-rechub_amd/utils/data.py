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
import ctypes

import torch
from torch.utils.data import DataLoader, Dataset, random_split

from .. import _lib, ops


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

    sparse : (N, F) int64 on the device, columns named ``sparse_names``
    dense  : (N, ND) float32 on the device (or None), columns named ``dense_names``
    label  : (N,) float32 on the device
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
        self.sparse, self.dense, self.label = sparse, dense, label.float().contiguous()
        self.sparse_names, self.dense_names = list(sparse_names), list(dense_names or [])
        self.N, self.F = sparse.shape
        self.ND = 0 if dense is None else dense.shape[1]
        self.batch_size = int(batch_size)
        self.shuffle, self.drop_last = shuffle, drop_last
        self.generator = generator
        dev = sparse.device
        self.pos = torch.zeros(1, dtype=torch.int64, device=dev)
        self.perm = torch.arange(self.N, dtype=torch.int64, device=dev)
        self._bufs = {}

    def __len__(self):
        full, rem = divmod(self.N, self.batch_size)
        return full + (1 if rem and not self.drop_last else 0)

    def _buffers(self, B):
        b = self._bufs.get(B)
        if b is None:
            dev = self.sparse.device
            sp = torch.empty((B, self.F), dtype=torch.int64, device=dev)
            de = torch.empty((B, self.ND), dtype=torch.float32, device=dev) if self.ND else None
            y = torch.empty((B,), dtype=torch.float32, device=dev)
            x = DeviceBatch()
            for j, n in enumerate(self.sparse_names):
                x[n] = sp[:, j]
            for j, n in enumerate(self.dense_names):
                x[n] = de[:, j]
            x.sparse, x.sparse_names, x.dense, x.dense_names = sp, self.sparse_names, de, self.dense_names
            b = (x, y, sp, de)
            self._bufs[B] = b
        return b

    def reshuffle(self):
        if self.shuffle:  # in place: a captured hipGraph keeps reading the same buffer
            self.perm.copy_(torch.randperm(self.N, dtype=torch.int64, device=self.sparse.device,
                                           generator=self.generator))
        self.pos.zero_()

    def load_next(self, B=None):
        """Assemble the batch at the current position into the static buffers and advance (2 launches)."""
        B = self.batch_size if B is None else B
        x, y, sp, de = self._buffers(B)
        s = ops._stream()
        _lib.call("rh_batch_gather", ops._p(self.perm), ops._p(self.pos), self.N, B, ops._p(self.sparse), self.F,
                  ops._p(self.dense), self.ND, ops._p(self.label), ops._p(sp), ops._p(de), ops._p(y), s)
        _lib.call("rh_batch_advance", ops._p(self.pos), B, self.N, s)
        return x, y

    def __iter__(self):
        self.reshuffle()
        full, rem = divmod(self.N, self.batch_size)
        for _ in range(full):
            yield self.load_next()
        if rem and not self.drop_last:
            yield self.load_next(rem)

"""Streaming Parquet dataset (API mirror of torch_rechub/data/dataset.py:17-107).

Same constructor (positional-only ``file_paths``, ``columns``, ``batch_size`` default 1024), same iteration contract
(a dict of column tensors per Arrow record batch, conversion by ``pa_array_to_tensor``), same file partitioning rule:
the files are split into ``ceil(n / parts)`` contiguous runs and part ``i`` takes run ``i`` (DataLoader workers in the
reference, dataset.py:88-107; here also ranks of the data-parallel job, see DeviceDataLoader.from_parquet).
"""
import pyarrow.dataset as pads
from torch.utils.data import IterableDataset, get_worker_info

from .convert import pa_array_to_tensor

_DEFAULT_BATCH_SIZE = 1024


def partition_files(file_paths, parts, index):
    """Contiguous run ``index`` of ``parts`` (the reference's per-worker rule): trailing parts may be empty."""
    paths = tuple(map(str, file_paths))
    if parts <= 1:
        return paths
    per = -(-len(paths) // parts)
    return paths[index * per:min(len(paths), (index + 1) * per)]


class ParquetIterableDataset(IterableDataset):

    def __init__(self, file_paths, /, columns=None, batch_size=_DEFAULT_BATCH_SIZE):
        self._file_paths = tuple(map(str, file_paths))
        self._columns = None if columns is None else tuple(columns)
        self._batch_size = batch_size

    def _get_partition(self):
        info = get_worker_info()
        if info is None:
            return self._file_paths
        return partition_files(self._file_paths, info.num_workers, info.id)

    def __iter__(self):
        mine = self._get_partition()
        if not mine:
            return
        scanner = pads.dataset(list(mine), format="parquet").scanner(
            columns=None if self._columns is None else list(self._columns), batch_size=self._batch_size)
        for rb in scanner.to_batches():
            yield {name: pa_array_to_tensor(col) for name, col in zip(rb.column_names, rb.columns)}

"""Parquet / Arrow input.  The known answers are the ones the reference's own tests hold for this path
(tests/test_pa_array_to_tensor.py, tests/test_parquet_dataset.py); the device ingestion is checked against the
file contents on the GPU."""
import os

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest
import torch
from torch.utils.data import DataLoader

from torch_rechub_amd.data import ParquetIterableDataset, pa_array_to_tensor, partition_files

SCALARS = [pa.int16(), pa.int32(), pa.int64(), pa.float16(), pa.float32(), pa.float64()]


def test_scalar_arrays_become_float32_with_nan_for_null():
    t = pa_array_to_tensor(pa.array([], type=pa.null()))
    assert t.dtype == torch.float32 and t.tolist() == []
    t = pa_array_to_tensor(pa.array([True, True, False]))
    assert t.dtype == torch.float32 and t.tolist() == [1.0, 1.0, 0.0]
    for dt in SCALARS:
        vals = [1.0, 2.0, 3.0] if pa.types.is_floating(dt) else [1, 2, 3]
        t = pa_array_to_tensor(pa.array(vals, type=dt))
        assert t.dtype == torch.float32 and t.tolist() == [1.0, 2.0, 3.0]
        t = pa_array_to_tensor(pa.array([None, None, vals[2]], type=dt))
        assert np.allclose(t.tolist(), [np.nan, np.nan, 3.0], equal_nan=True)
        t[0] = 5.0  # writable (torch.from_numpy would refuse a read-only view)


def test_nested_arrays_shapes_raggedness_and_types():
    t = pa_array_to_tensor(pa.array([[]], type=pa.list_(pa.null())))
    assert t.dtype == torch.float32 and t.tolist() == [[]]
    t = pa_array_to_tensor(pa.array([], type=pa.list_(pa.int32())))
    assert tuple(t.shape) == (0, 0)
    for dt in SCALARS + [pa.bool_()]:
        one = True if pa.types.is_boolean(dt) else 1
        for make in (pa.list_, pa.large_list, lambda v: pa.list_(v, 2)):
            t = pa_array_to_tensor(pa.array([[one, one], [one, None]], type=make(dt)))
            assert t.dtype == torch.float32 and tuple(t.shape) == (2, 2)
            assert np.allclose(t.tolist(), [[1.0, 1.0], [1.0, np.nan]], equal_nan=True)
    with pytest.raises(ValueError, match="ragged"):
        pa_array_to_tensor(pa.array([[1], [1, 2]], type=pa.list_(pa.int32())))
    with pytest.raises(TypeError, match="Unsupported array type"):
        pa_array_to_tensor(pa.array(["a"]))
    with pytest.raises(TypeError, match="nested"):
        pa_array_to_tensor(pa.array([["a"]], type=pa.list_(pa.string())))


def write_parts(tmp_path, specs):
    paths = []
    for i, spec in enumerate(specs):
        path = os.path.join(tmp_path, f"part{i}.parquet")
        pq.write_table(pa.table({k: pa.array(list(v)) for k, v in spec.items()}), path)
        paths.append(path)
    return paths


@pytest.mark.parametrize("workers,batch", [(0, 3), (3, 4), (2, 1024)])
def test_streaming_dataset_covers_every_row_once(tmp_path, workers, batch):
    paths = write_parts(str(tmp_path), [{"id": range(0, 7), "x": np.arange(7) * 0.5}, {"id": range(7, 14), "x": np.arange(7)},
                                        {"id": range(14, 21), "x": np.ones(7)}])
    ds = ParquetIterableDataset(paths, columns=["id"], batch_size=batch)
    seen = []
    for b in DataLoader(ds, batch_size=None, num_workers=workers):
        assert isinstance(b, dict) and list(b) == ["id"] and b["id"].dtype == torch.float32 and len(b["id"]) <= batch
        seen.extend(b["id"].tolist())
    assert sorted(seen) == list(map(float, range(21)))
    both = next(iter(ParquetIterableDataset(paths)))  # columns=None: every column
    assert set(both) == {"id", "x"}


def test_file_partition_rule_is_the_reference_worker_rule():
    files = [f"f{i}" for i in range(7)]
    assert partition_files(files, 1, 0) == tuple(files)
    assert [partition_files(files, 3, i) for i in range(3)] == [("f0", "f1", "f2"), ("f3", "f4", "f5"), ("f6",)]
    assert [len(partition_files(files, 8, i)) for i in range(8)] == [1] * 7 + [0]
    assert partition_files(files[:2], 4, 3) == ()


@pytest.mark.gpu
def test_device_loader_from_parquet_holds_the_file_contents(tmp_path):
    from torch_rechub_amd.utils.data import DeviceDataLoader
    rng = np.random.default_rng(0)
    specs, want = [], []
    for n in (1000, 1, 2500, 700):
        spec = {"c1": rng.integers(0, 50_000_000, n), "c2": rng.integers(0, 7, n).astype(np.int32),
                "d1": rng.random(n), "d2": rng.random(n).astype(np.float32), "label": rng.integers(0, 2, n)}
        specs.append(spec)
        want.append(spec)
    paths = write_parts(str(tmp_path), specs)
    for world, rank in ((1, 0), (2, 0), (2, 1)):
        dl = DeviceDataLoader.from_parquet(paths, ["c1", "c2"], ["d1", "d2"], "label", batch_size=512, shuffle=False,
                                           rank=rank, world=world, chunk_rows=600)
        mine = want[rank * 2:(rank + 1) * 2] if world == 2 else want
        ref = {k: np.concatenate([np.asarray(w[k]) for w in mine]) for k in mine[0]}
        assert dl.N == len(ref["label"])
        assert np.array_equal(dl.sparse.cpu().numpy(), np.stack([ref["c1"], ref["c2"]], 1).astype(np.int64))  # exact ids > 2^24
        assert np.array_equal(dl.dense.cpu().numpy(), np.stack([ref["d1"], ref["d2"]], 1).astype(np.float32))
        assert np.array_equal(dl.label.cpu().numpy(), ref["label"].astype(np.float32))
        x, y = next(iter(dl))
        assert torch.equal(x["c1"], dl.sparse[:512, 0]) and torch.equal(x["d2"], dl.dense[:512, 1]) and torch.equal(y, dl.label[:512])
    with pytest.raises(ValueError, match="no Parquet file"):
        DeviceDataLoader.from_parquet(paths[:1], ["c1"], [], "label", 8, rank=1, world=2)

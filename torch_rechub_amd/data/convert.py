"""Arrow array -> torch tensor (API mirror of torch_rechub/data/convert.py:10-44).

Contract of the reference, reproduced (its tests/test_pa_array_to_tensor.py is mirrored in
tests/test_parquet_input.py): every supported array becomes a float32 tensor — bool / integer / floating / null
scalars give (n,), list / large_list / fixed_size_list of those give (n, width); nulls become NaN; a ragged nested
array is a ValueError, anything else a TypeError; an empty nested array has shape (0, 0).

``pa_column_to_numpy`` is the exact-typed variant the device ingestion uses: the reference's float32 cast silently
corrupts integer ids above 2^24 (Criteo's largest vocabularies are ~10^7, so it happens to be harmless there); the
device path keeps integer columns as int64.
"""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pyarrow.types as pt
import torch


def _scalar_ok(t):
    return pt.is_boolean(t) or pt.is_integer(t) or pt.is_floating(t) or pt.is_null(t)


def _list_ok(t):
    return pt.is_list(t) or pt.is_large_list(t) or pt.is_fixed_size_list(t)


def _f32(values):
    """float32 numpy copy of a scalar Arrow array (nulls -> NaN), writable so torch.from_numpy accepts it."""
    return pc.cast(values, pa.float32()).to_numpy(zero_copy_only=False, writable=True)


def pa_array_to_tensor(arr):
    t = arr.type
    if _scalar_ok(t):
        return torch.from_numpy(_f32(arr))
    if not _list_ok(t):
        raise TypeError(f"Unsupported array type: {t}")
    if not _scalar_ok(t.value_type):
        raise TypeError(f"Unsupported value type in the nested array: {t.value_type}")
    n = len(arr)
    if len(pc.unique(pc.list_value_length(arr))) > 1:
        raise ValueError("Cannot convert the ragged nested array.")
    flat = _f32(pc.cast(arr, pa.list_(pa.float32())).values)
    return torch.from_numpy(flat.reshape(n, -1) if n > 0 else flat.reshape(0, 0))


def pa_column_to_numpy(col, integer):
    """One table column (Array or ChunkedArray, scalar type, no nulls expected) as int64 or float32 numpy, exact."""
    if isinstance(col, pa.ChunkedArray):
        col = col.combine_chunks()
    if not _scalar_ok(col.type):
        raise TypeError(f"Unsupported column type for the device loader: {col.type}")
    if col.null_count:
        raise ValueError("the device loader does not accept null values (encode / impute them first)")
    if integer:
        if pt.is_floating(col.type):
            out = col.to_numpy(zero_copy_only=False)
            if not np.all(out == np.floor(out)):
                raise ValueError("a sparse (id) column holds non-integral values")
            return out.astype(np.int64)
        return pc.cast(col, pa.int64()).to_numpy(zero_copy_only=False)
    return pc.cast(col, pa.float32()).to_numpy(zero_copy_only=False)

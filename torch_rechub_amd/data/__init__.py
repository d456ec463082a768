"""Parquet / Arrow input (API mirror of torch_rechub/data): host streaming dataset + conversion helpers.
The device-resident ingestion built on them is ``torch_rechub_amd.utils.data.DeviceDataLoader.from_parquet``."""
from .convert import pa_array_to_tensor  # noqa: F401
from .dataset import ParquetIterableDataset, partition_files  # noqa: F401

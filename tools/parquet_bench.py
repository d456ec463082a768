#!/usr/bin/env python
"""Ingest rate of DeviceDataLoader.from_parquet (Parquet -> pinned staging -> HBM) beside the reference-style host path
(ParquetIterableDataset: Arrow -> float32 tensors on the CPU), Criteo-shape rows (26 int64 ids + 13 float32 + label).
    python tools/parquet_bench.py [--rows 4000000] [--files 8]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CRITEO_VOCABS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4_000_000)
    ap.add_argument("--files", type=int, default=8)
    a = ap.parse_args()
    from torch_rechub_amd.data import ParquetIterableDataset
    from torch_rechub_amd.utils.data import DeviceDataLoader
    rng = np.random.default_rng(2022)
    sparse = [f"C{i + 1}" for i in range(26)]
    dense = [f"I{i + 1}" for i in range(13)]
    with tempfile.TemporaryDirectory() as tmp:
        paths, per = [], a.rows // a.files
        for f in range(a.files):
            cols = {n: rng.integers(0, v, per) for n, v in zip(sparse, CRITEO_VOCABS)}
            cols.update({n: rng.random(per, dtype=np.float32) for n in dense})
            cols["label"] = rng.integers(0, 2, per).astype(np.int8)
            path = os.path.join(tmp, f"part{f}.parquet")
            pq.write_table(pa.table(cols), path)
            paths.append(path)
        size = sum(os.path.getsize(p) for p in paths)
        t0 = time.perf_counter()
        dl = DeviceDataLoader.from_parquet(paths, sparse, dense, "label", batch_size=4096)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"from_parquet: {dl.N} rows ({size / 2**20:.0f} MiB of Parquet, {dl.N * (26 * 8 + 13 * 4 + 4) / 2**20:.0f} MiB in HBM) "
              f"in {dt:.2f} s = {dl.N / dt / 1e6:.2f} M rows/s", flush=True)
        t0 = time.perf_counter()
        n = 0
        for b in ParquetIterableDataset(paths, batch_size=4096):
            n += len(b["label"])
            if n >= min(a.rows, 1_000_000):
                break
        dt = time.perf_counter() - t0
        print(f"reference-style host iteration (float32 dict of 40 tensors per 4096-row batch, CPU): {n / dt / 1e6:.2f} M rows/s")


if __name__ == "__main__":
    main()

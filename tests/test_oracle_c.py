"""The two CPU restatements of the integer / index paths -- numpy (oracle/ctr_oracle.py) and plain C
(oracle/int_paths.c, built with gcc) -- must agree bit for bit with each other and with the reference's golden gather.
The HIP kernels are compared against the numpy side in tests/test_gpu_kernels.py."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import c_oracle as C
from oracle import ctr_oracle as O

pytestmark = pytest.mark.skipif(C.load() is None, reason="no C compiler for oracle/int_paths.c")


def test_c_gather_is_the_numpy_gather_and_the_reference_vectors(layers_golden):
    g = layers_golden
    names = ("s0", "s1", "s2", "s3", "s4")
    t = {n: g[f"emb.table.{n}"] for n in names}
    tables = [t["s0"], t["s1"], t["s2"], t["s3"], t["s4"], t["s1"]]  # field 5 shares s1's table (tests/test_oracle_golden.py)
    idx = g["emb.idx"]
    out, bad = C.embedding_gather(tables, idx)
    assert bad == 0
    assert np.array_equal(out, O.embedding_gather(tables, idx))
    assert np.array_equal(out, g["emb.out_3d"])  # produced by the unmodified reference's EmbeddingLayer
    wrong = idx.copy()
    wrong[3, 2] = tables[2].shape[0]
    assert C.embedding_gather(tables, wrong)[1] == 1
    with pytest.raises(IndexError):
        O.embedding_gather(tables, wrong)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_c_shard_localize_is_the_numpy_one(world):
    rng = np.random.default_rng(world)
    vocabs = [1, 3, 10, 17, 64, 1000, 100003]
    pads = [None, 0, 2, None, 63, None, 5]
    idx = np.stack([rng.integers(0, v, 500) for v in vocabs], axis=1)
    for rank in range(world):
        got, bad = C.shard_localize(idx, vocabs, pads, world, rank)
        assert bad == 0 and np.array_equal(got, O.shard_localize(idx, vocabs, pads, world, rank))
    idx[7, 3] = 17
    assert C.shard_localize(idx, vocabs, pads, world, 0)[1] == 1


def test_c_sampler_stream_is_the_numpy_one():
    for seed, ctr, B, cols, row0, K in ((7, 3, 12, 12, 0, 5), (1234, 0, 32, 96, 64, 7), (1, 9, 4, 3000, 2996, 2999),
                                        (5, 1, 6, 6, 0, 5)):
        assert np.array_equal(C.inbatch_sample_rows(seed, ctr, B, cols, row0, K),
                              O.inbatch_sample_rows(seed, ctr, B, cols, row0, K))
    with pytest.raises(ValueError):
        C.inbatch_sample_rows(1, 0, 8, 10, 5, 3)


def test_c_batch_gather_is_the_numpy_one():
    rng = np.random.default_rng(3)
    N, F, ND, B = 101, 5, 3, 16
    perm = rng.permutation(N)
    sparse, dense, label = rng.integers(0, 50, (N, F)), rng.random((N, ND), dtype=np.float32), rng.random(N).astype(np.float32)
    for pos in (0, 37, 95):  # the last one wraps around the end of the permutation
        so, do, lo = C.batch_gather(perm, pos, B, sparse, dense, label)
        ws, wd, wl = O.batch_gather(perm, pos, B, sparse, dense, label)
        assert np.array_equal(so, ws) and np.array_equal(do, wd) and np.array_equal(lo, wl)

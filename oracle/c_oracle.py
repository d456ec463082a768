"""ctypes access to the plain-C restatement of the integer paths (oracle/int_paths.c).  TEST INFRASTRUCTURE ONLY.

``load()`` builds oracle/_c/liboracle_int.so with ``make -C oracle`` when it is missing (gcc) and returns None when no
C compiler is available, so callers can skip."""
import ctypes
import os
import shutil
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_c", "liboracle_int.so")
_lib = None


def build():
    if shutil.which(os.environ.get("CC", "gcc")) is None and shutil.which("cc") is None:
        return False
    subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
    return os.path.exists(LIB)


def load():
    global _lib
    if _lib is not None:
        return _lib
    src = os.path.join(HERE, "int_paths.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        if not build():
            return None
    lib = ctypes.CDLL(LIB)
    p, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    lib.o_embedding_gather.argtypes = [p, p, p, i64, i32, i32, p]
    lib.o_embedding_gather.restype = i64
    lib.o_batch_gather.argtypes = [p, i64, i64, i64, p, i32, p, i32, p, p, p, p]
    lib.o_batch_gather.restype = None
    lib.o_shard_localize.argtypes = [p, i64, i32, p, p, i32, i32, p]
    lib.o_shard_localize.restype = i64
    lib.o_inbatch_sample_rows.argtypes = [ctypes.c_uint64, ctypes.c_uint64, i32, i32, i32, i32, p]
    lib.o_inbatch_sample_rows.restype = i32
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def embedding_gather(tables, idx):
    """(out (B, F, D) float32, number of out-of-range indices)."""
    lib = load()
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    B, F = idx.shape
    tabs = [np.ascontiguousarray(t, dtype=np.float32) for t in tables]
    D = tabs[0].shape[1]
    ptrs = (ctypes.c_void_p * F)(*[t.ctypes.data for t in tabs])
    vocab = np.array([t.shape[0] for t in tabs], dtype=np.int64)
    out = np.empty((B, F, D), dtype=np.float32)
    bad = lib.o_embedding_gather(ptrs, _ptr(vocab), _ptr(idx), B, F, D, _ptr(out))
    return out, int(bad)


def batch_gather(perm, pos, B, sparse, dense, label):
    lib = load()
    perm = np.ascontiguousarray(perm, dtype=np.int64)
    sparse = np.ascontiguousarray(sparse, dtype=np.int64)
    dense = np.ascontiguousarray(dense, dtype=np.float32)
    label = np.ascontiguousarray(label, dtype=np.float32)
    so = np.empty((B, sparse.shape[1]), dtype=np.int64)
    do = np.empty((B, dense.shape[1]), dtype=np.float32)
    lo = np.empty((B,), dtype=np.float32)
    lib.o_batch_gather(_ptr(perm), int(pos), perm.shape[0], B, _ptr(sparse), sparse.shape[1], _ptr(dense), dense.shape[1],
                       _ptr(label), _ptr(so), _ptr(do), _ptr(lo))
    return so, do, lo


def shard_localize(idx, vocabs, pads, world, rank):
    """(local (N, F) int32, number of out-of-range ids)."""
    lib = load()
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    vocab = np.array(vocabs, dtype=np.int64)
    pad = np.array([-1 if p is None else p for p in pads], dtype=np.int64)
    out = np.empty(idx.shape, dtype=np.int32)
    bad = lib.o_shard_localize(_ptr(idx), idx.shape[0], idx.shape[1], _ptr(vocab), _ptr(pad), world, rank, _ptr(out))
    return out, int(bad)


def inbatch_sample_rows(seed, ctr, B, cols, row0, K):
    lib = load()
    out = np.empty((B, K), dtype=np.int64)
    rc = lib.o_inbatch_sample_rows(seed, ctr, B, cols, row0, K, _ptr(out))
    if rc != 0:
        raise ValueError("bad sizes")
    return out

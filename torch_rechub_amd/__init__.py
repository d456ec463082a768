"""torch_rechub_amd — MI355X (gfx950) native CTR-training hot path behind torch-rechub's layer/trainer API.

The package directory IS the import name; ``torch-rechub_amd`` at the repo root is a symlink onto it (the hyphenated
project name cannot be imported).

Layout (mirrors only what the hot path needs from the reference package):
  csrc/        hand-written HIP kernels + the C ABI (include/rechub_hip.h) -> librechub_hip.so
  _lib.py      ctypes binding; ops.py: autograd functions over raw pointers + the current HIP stream
  basic/       features, initializers, activation, layers, loss_func, callback   (reference basic/*)
  models/ranking/  DeepFM, WideDeep, DCN, DCNv2, DIN, DIEN, BST, AFM, AutoInt, EDCN, FiBiNet   (reference models/ranking/*)
  models/matching/ DSSM; models/multi_task/ SharedBottom, ESMM, MMOE, PLE, AITM
  trainers/    CTRTrainer, MatchTrainer, MTLTrainer                            (reference trainers/*.py)
  utils/data.py    DataGenerator / TorchDataset + the HBM-resident DeviceDataLoader
  optim.py     FusedDenseAdam (torch.optim.Adam semantics, tables stepped by one HIP launch)
  distributed.py   one-process-per-GPU data parallel over RCCL (dense all-reduce + sparse row exchange)
  sharding.py      row-sharded tables (one shard per rank) + differentiable row collectives (cross-rank negatives)
"""
__version__ = "0.1.0"

from . import basic, models, trainers, utils  # noqa: E402,F401

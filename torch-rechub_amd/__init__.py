"""torch-rechub_amd — MI355X (gfx950) native CTR-training hot path behind torch-rechub's layer/trainer API.

Import name: ``torch_rechub_amd`` (the directory carries the hyphenated project name, which Python
cannot import; ``torch_rechub_amd/__init__.py`` at the repo root is a two-line alias onto it).

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

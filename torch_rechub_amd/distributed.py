"""One-process-per-GPU data parallelism over RCCL (xGMI), replacing ``nn.DataParallel``.

Reference: CTRTrainer wraps the model in single-process ``torch.nn.DataParallel`` when len(gpus) > 1
(trainers/ctr_trainer.py:53-55): every forward broadcasts ALL parameters (2 GiB of tables) and every
backward reduces DENSE full-table gradients onto gpus[0].  Here each rank owns a replica and exchanges:

* dense (non-embedding) gradients: packed by one ``torch.cat`` into ONE flat buffer, all-reduced (SUM; the
  trainer scales the loss by 1/world) on a side HIP stream.  The pack + reduction of everything already
  produced is launched when the embedding backward starts (``ops.add_pre_embed_backward_hook``), so it
  overlaps the scatter kernels and the sparse exchange; the few late gradients (e.g. the fused LR weight)
  go in ``finish()``.  The optimizer reads the flat buffer directly (``rh_adam_small``).
* embedding gradients: all-gather of (index matrix (B,F), gradient rows (B,F,D)) followed by a local
  scatter-add (``rh_embed_scatter_rows``) — the same sum ``DataParallel`` computes, without moving
  vocab-sized tensors (68-72 B per lookup instead of 2 GiB per step).

BatchNorm statistics stay per rank, which is what DataParallel replicas do.
Works with the ``nccl`` (= RCCL) backend on GPUs and with ``gloo`` on CPU tensors (tests).
"""
import os

import torch
import torch.distributed as dist
from torch import nn

from . import graphs, ops


def table_parameters(model):
    """Parameters that belong to nn.Embedding modules (deduplicated, model.parameters() order).  A module flagged
    ``_rh_dense`` (a small table read as a slice, e.g. BST's positional table) counts as a dense parameter."""
    ids = set()
    for m in model.modules():
        if isinstance(m, (nn.Embedding, nn.EmbeddingBag)) and not getattr(m, "_rh_dense", False):
            ids.update(id(p) for p in m.parameters())
    return [p for p in model.parameters() if id(p) in ids]


def pack_indices(idx_list):
    """(B,F) contiguous index matrix of per-field (B,) columns; zero-copy when they already are columns of one."""
    first = idx_list[0]
    F, B = len(idx_list), first.shape[0]
    step = first.element_size()
    if all(t.data_ptr() == first.data_ptr() + f * step and t.stride(0) == F for f, t in enumerate(idx_list)):
        return torch.as_strided(first, (B, F), (F, 1))
    return torch.stack(idx_list, dim=1)


# RECHUB_EMULATE_WORLD=N (single-GPU study of the per-rank cost of an N-rank job): the gathers return N copies of the
# local shard, so that the scatter / optimizer passes see the row volume of N ranks.  Gradients are N-fold then: for
# timing only.
_EMULATE_WORLD = int(os.environ.get("RECHUB_EMULATE_WORLD", "0"))


def all_gather_cat(t, group=None, out=None):
    """Concatenate ``t`` from every rank along dim 0 (rank order), optionally into a preallocated ``out``."""
    world = dist.get_world_size(group)
    t = t.contiguous()
    if _EMULATE_WORLD > 1 and world == 1:
        if out is None or out.shape[0] != _EMULATE_WORLD * t.shape[0]:
            out = torch.empty((_EMULATE_WORLD * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        # ONE broadcast copy (a real all-gather is one collective launch, not N copies: seven extra 5 us launches per gather
        # made the emulated 8-rank step look 65 us longer than its kernels)
        out.view((_EMULATE_WORLD,) + tuple(t.shape)).copy_(t.unsqueeze(0).expand((_EMULATE_WORLD,) + tuple(t.shape)))
        return out
    if out is None:
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    try:
        dist.all_gather_into_tensor(out, t, group=group)
    except (RuntimeError, NotImplementedError):
        parts = list(out.chunk(world, dim=0))
        dist.all_gather(parts, t, group=group)
    return out


class DenseGradBucket(object):
    """The dense (non-embedding) gradients of one step, packed into ONE flat buffer.

    ``zero()`` replaces ``model.zero_grad()`` for these parameters (grads -> None, so autograd hands over its freshly
    produced gradient tensors instead of launching one accumulate kernel per parameter); ``flush()`` packs every gradient
    that exists and is not packed yet (one ``torch.cat`` per contiguous run, normally one) and, with world > 1, starts its
    all-reduce (SUM) on a side HIP stream; ``finish()`` packs / reduces the rest and joins.  The optimizer then reads the
    flat buffer (``rh_adam_small``).  ``flush()`` is called from a pre-hook of the embedding backward, so the all-reduce
    of everything the MLP produced overlaps the embedding scatter kernels and the sparse row exchange.
    """

    def __init__(self, params, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.params = [p for p in params if p.requires_grad]
        self.sizes = [p.numel() for p in self.params]
        self.offsets = [0]
        for n in self.sizes:
            self.offsets.append(self.offsets[-1] + n)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(self.offsets[-1], dtype=torch.float32, device=dev)
        self.use_cuda = dev.type == "cuda"
        # (ONE stream per role and process, not a pooled stream per bucket: graphs.role_stream says why)
        self.side = graphs.role_stream("dense_allreduce", dev) if (self.use_cuda and dist.is_available() and
                                                                    dist.is_initialized()) else None
        self.packed = [False] * len(self.params)
        self.pending = []
        self.defer = False  # True: flush() only packs; reduce_deferred() starts the all-reduces (split-graph step)
        self._deferred_runs = []
        self._pack_keep_dp = []  # sources of this step's slab-packing launches (flush with ops.deferred armed)

    def view(self, i):
        return self.flat[self.offsets[i]:self.offsets[i + 1]].view_as(self.params[i])

    def zero(self):
        for p in self.params:
            p.grad = None
        self.packed = [False] * len(self.params)
        self._pack_keep_dp = []

    def all_present(self):
        slabs = ops.deferred.items if ops.deferred.armed is not None else {}
        return all(p.grad is not None or id(p) in slabs for p in self.params)

    def _runs(self, want_missing, slabs=None):
        """Maximal runs [i, j) of not-yet-packed parameters that have (want_missing = False) / lack a gradient -- a ``.grad``
        tensor or, with ``slabs`` (ops.deferred.items), a registered slab of partial rows."""
        def missing(k):
            p = self.params[k]
            return p.grad is None and (slabs is None or id(p) not in slabs)

        def ready(k):
            # a slab is the parameter's gradient only once EVERY use of the parameter in this backward has reported
            # (ops.DeferredGrads.final: a Linear shared across two embedding lookups registers its first slab early)
            return want_missing or slabs is None or ops.deferred.final(self.params[k])
        runs, i, n = [], 0, len(self.params)
        while i < n:
            ok = (not self.packed[i]) and (missing(i) == want_missing) and ready(i)
            if ok:
                j = i
                while j < n and (not self.packed[j]) and (missing(j) == want_missing) and ready(j):
                    j += 1
                runs.append((i, j))
                i = j
            else:
                i += 1
        return runs

    def _reduce(self, runs):
        if self.group_size() == 1 or not runs:
            return
        if self.defer:
            self._deferred_runs += runs
            return
        self._launch(runs)

    def group_size(self):
        return self.world if not self.force else max(self.world, 2)

    force = False  # DataParallelContext(force=True): exercise the collectives even at world size 1 (1-GPU validation)

    def reduce_deferred(self):
        runs, self._deferred_runs = self._deferred_runs, []
        self._launch(runs)

    def join(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def _launch(self, runs):
        if not runs:
            return
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream())
        for i, j in runs:
            chunk = self.flat[self.offsets[i]:self.offsets[j]]
            if self.side is not None:
                with torch.cuda.stream(self.side):
                    self.pending.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:
                self.pending.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def flush(self):
        """Pack (+ start reducing) every gradient that exists and has not been packed yet.

        With ``ops.deferred`` armed for these parameters (the trainers' data-parallel step since round 5) a gradient may exist
        only as the per-split / per-block partial SLAB its backward kernel left behind: the runs are then packed by ONE
        rh_pack_grads launch that sums the slabs in split order straight into the flat bucket -- instead of one reduction
        launch per slab (wgrad_reduce x layers, colsum x 2) and a torch.cat per run -- and the all-reduce starts on that."""
        slabs = ops.deferred.items if (ops.deferred.armed is not None and self.use_cuda) else None
        runs = self._runs(want_missing=False, slabs=slabs)
        if slabs is not None:
            if runs:
                self._pack_runs(runs, slabs)
            self._reduce(runs)
            return
        for i, j in runs:
            torch.cat([self.params[k].grad.reshape(-1) for k in range(i, j)],
                      out=self.flat[self.offsets[i]:self.offsets[j]])
            for k in range(i, j):
                self.packed[k] = True
        self._reduce(runs)

    def _pack_runs(self, runs, slabs):
        import ctypes

        from . import _lib
        ops.flush_wgrad_rider()
        idx = [k for i, j in runs for k in range(i, j)]
        items = (_lib.PackItem * len(idx))()
        keep = []
        for n, k in enumerate(idx):
            p = self.params[k]
            it, rec, g = items[n], slabs.get(id(p)), p.grad
            it.numel, it.dst_offset = self.sizes[k], self.offsets[k]
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
                g = g.float().contiguous()
            if g is not None:
                keep.append(g)
            if rec is not None:
                it.src, it.nparts, it.stride = rec["src"], rec["nparts"], rec["stride"]
                it.add = g.data_ptr() if g is not None else 0
                keep.append(rec["keep"])
            else:
                it.src, it.nparts, it.stride, it.add = g.data_ptr(), 1, self.sizes[k], 0
            self.packed[k] = True
        self._pack_keep_dp.extend(keep)  # alive until the step's launches are enqueued (under capture: pooled by the graph)
        _lib.call("rh_pack_grads", ctypes.cast(items, ctypes.c_void_p), len(idx), ops._p(self.flat), ops._stream())

    def finish(self, assign_views=False):
        """Pack / reduce what is left (parameters without a gradient contribute zeros, like DDP) and join."""
        ops.deferred.backward_done()  # (called behind loss.backward(): every gradient that exists is final now)
        self.flush()
        slabs = ops.deferred.items if (ops.deferred.armed is not None and self.use_cuda) else None
        missing = self._runs(want_missing=True, slabs=slabs)
        for i, j in missing:
            self.flat[self.offsets[i]:self.offsets[j]].zero_()
            for k in range(i, j):
                self.packed[k] = True
        self._reduce(missing)
        if not self.defer:
            self.join()
        if assign_views:  # a stock torch optimizer reads p.grad: point it at the (reduced) bucket
            for i, p in enumerate(self.params):
                p.grad = self.view(i)

    def pack(self, deferred_items, adam=None, gate=None):
        """Single-GPU fast path: ONE launch (rh_pack_grads) fills the flat bucket from whatever each parameter has -- a
        slab of partial rows registered in ``ops.deferred`` (summed in fixed order), a plain ``.grad`` tensor (copied; added
        on top of a slab when both exist), or nothing (zeros) -- instead of torch.cat + one reduction launch per slab."""
        import ctypes

        from . import _lib
        ops.flush_wgrad_rider()  # (a packing launch in FRONT of optimizer.step(): slabs still riding are launched now)
        n = len(self.params)
        items = (_lib.PackItem * n)()
        keep = []
        for i, p in enumerate(self.params):
            it, rec, g = items[i], deferred_items.get(id(p)), p.grad
            it.numel, it.dst_offset = self.sizes[i], self.offsets[i]
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
                g = g.float().contiguous()
            if g is not None:
                keep.append(g)
            if rec is not None:
                it.src, it.nparts, it.stride = rec["src"], rec["nparts"], rec["stride"]
                it.add = g.data_ptr() if g is not None else 0
                keep.append(rec["keep"])
            elif g is not None:
                it.src, it.nparts, it.stride, it.add = g.data_ptr(), 1, self.sizes[i], 0
            else:
                it.src, it.nparts, it.stride, it.add = 0, 0, 0, 0
            self.packed[i] = True
        self._pack_keep = keep  # alive until the launch is enqueued (and, under capture, pooled by the graph)
        if adam is not None and gate is not None:  # ... and the launch opens the deferred sweep's gate when it starts (optim.py)
            _lib.call("rh_pack_grads_adam_gate", ctypes.cast(items, ctypes.c_void_p), n, ops._p(self.flat), ops._p(adam[0]),
                      ops._p(adam[1]), ops._p(gate), ops._stream())
        elif adam is not None:  # (sdesc, hyper) of optim.TableAdam.small_adam_args(): the dense parameters' Adam step rides along
            _lib.call("rh_pack_grads_adam", ctypes.cast(items, ctypes.c_void_p), n, ops._p(self.flat), ops._p(adam[0]),
                      ops._p(adam[1]), ops._stream())
        else:
            _lib.call("rh_pack_grads", ctypes.cast(items, ctypes.c_void_p), n, ops._p(self.flat), ops._stream())

    def close(self):
        self.pending = []


class DataParallelContext(object):
    """Replica synchronisation for one model on this rank."""

    def __init__(self, model, group=None, broadcast=True, force=False, shard_tables=False, shard_min_rows=0):
        """``shard_tables``: keep ONE row-shard of every table (>= shard_min_rows rows) on this rank instead of a
        replica (sharding.py): their rows travel inside the forward / backward (all-gather of indices, reduce-scatter
        of rows), so they take no part in the gradient-row exchange below."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch with torchrun)")
        self.deferred_mode = False  # True: the backward only records (call, rows); exchange_deferred() runs later
        self.deferred = []
        self._gather_bufs = {}
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.model = model
        if broadcast:
            with torch.no_grad():
                for t in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(t, src=0, group=group)
        self.sharded = []
        if shard_tables:
            from . import sharding
            self.sharded = sharding.shard_tables(model, group, min_rows=shard_min_rows)
        tables = {id(p) for p in table_parameters(model)}
        dense = [p for p in model.parameters() if id(p) not in tables]
        self.bucket = DenseGradBucket(dense, group)
        self.bucket.force = force
        self._hook = ops.add_pre_embed_backward_hook(self.bucket.flush)
        ops.set_sparse_exchange(self.sparse_exchange, self.gather_rows)

    def sparse_exchange(self, call, rows):
        """all-gather (indices, gradient rows) of the local batch from every rank."""
        if self.deferred_mode:
            self.deferred.append((call, rows))
            return None
        idx_all = all_gather_cat(pack_indices(call.idx), self.group)
        rows_all = all_gather_cat(rows, self.group)
        return idx_all, rows_all

    def gather_rows(self, t):
        """Rows of every rank, rank order (sequence-feature backward: its indices and gradient rows)."""
        return all_gather_cat(t, self.group)

    def exchange_deferred(self, deferred):
        """Run the recorded exchanges into STATIC gather buffers (addresses must not change between graph replays)."""
        out = []
        for i, (call, rows) in enumerate(deferred):
            packed = pack_indices(call.idx)
            bufs = self._gather_bufs.get(i)
            w = max(self.world, _EMULATE_WORLD)
            if bufs is None or bufs[1].shape[1:] != rows.shape[1:] or bufs[1].shape[0] != w * rows.shape[0]:
                bufs = (torch.empty((w * packed.shape[0], packed.shape[1]), dtype=packed.dtype, device=packed.device),
                        torch.empty((w * rows.shape[0],) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device))
                self._gather_bufs[i] = bufs
            all_gather_cat(packed, self.group, out=bufs[0])
            all_gather_cat(rows, self.group, out=bufs[1])
            out.append((call, bufs[0], bufs[1]))
        return out

    def close(self):
        ops.remove_pre_embed_backward_hook(self._hook)
        ops.set_sparse_exchange(None)
        self.bucket.close()

"""Row-sharded embedding tables: one shard per rank instead of one replica per rank.

The reference trains replicated tables: ``nn.DataParallel`` broadcasts every table to every GPU in every forward and
reduces vocab-sized dense gradients (trainers/ctr_trainer.py:53-55), and its default optimizer touches every row of
every table every step (ctr_trainer.py:59-61).  ``distributed.DataParallelContext`` keeps the replicas but exchanges only
the looked-up rows; this module removes the replicas: rank r keeps the rows g with g % world == r of every table
(local row g // world) plus one all-zero *sink* row, so table memory, optimizer state and the optimizer's sweep
traffic are 1 / world per GPU -- what lets the 100 M-item table of the DSSM configuration grow past one GPU.

One lookup of the global batch (B rows per rank, F fields):

    idx (B, F)  --all-gather-->  idx_all (W*B, F)                      4-8 B per lookup and rank
    rh_shard_localize: owned -> local row, not owned / padding -> sink  one launch
    fused gather over the local shards -> (W*B, F*D), zeros where this rank does not own the row
    reduce-scatter (SUM) -> (B, F*D)                                    exactly one non-zero term per element: the sum
                                                                        IS the reference's row, in any order
    backward: all-gather of the (B, F*D) gradient; the local scatter-add / lazy Adam skip the sink row as padding_idx.

The result equals the reference's lookup on replicated tables; the gradient a row receives is the sum over the lookups
of ALL ranks (the trainer scales the loss by 1 / world), i.e. what DataParallel's reduce computes for the global batch.
Everything runs on the caller's stream, so a hipGraph capture of the step contains the collectives.

``gather_rows`` (all-gather with a reduce-scatter backward) also carries the cross-rank in-batch negatives of the
two-tower trainer (match_trainer.py:118-138 is single-device; here every rank scores its users against the items of
all ranks).

Batch sizes and batch counts must be equal on all ranks: lookups are collective calls.  The trainers check it on
every rank before an epoch / a sharded evaluation (``CTRTrainer._check_equal_batches``) and
``DeviceDataLoader.from_parquet`` truncates every rank to the minimum row count.
"""
import os

import torch
import torch.distributed as dist
from torch import nn

from . import ops
from .distributed import _EMULATE_WORLD, all_gather_cat, pack_indices


def local_row_count(vocab, world):
    """Rows a shard holds (without the sink row); the same on every rank, trailing rows of high ranks are unused."""
    return -(-int(vocab) // int(world))


class RowShard(object):
    """Placement of one table, attached to its ``nn.Embedding`` as ``_rh_shard``."""

    def __init__(self, vocab, dim, world, rank, pad, group):
        self.vocab, self.dim, self.world, self.rank, self.group = int(vocab), int(dim), int(world), int(rank), group
        self.pad = -1 if pad is None else int(pad)
        self.sink = local_row_count(vocab, world)  # local index of the all-zero row
        if self.sink + 1 >= 2**31:
            raise ValueError(f"shard of {vocab} rows over {world} ranks does not fit int32 local indices")

    def owned(self):
        """Global row ids of the local rows 0 .. n-1 (n <= sink)."""
        return range(self.rank, self.vocab, self.world)


def _refuse_direct_call(*a, **kw):
    raise RuntimeError("torch_rechub_amd: this nn.Embedding holds one shard of a row-sharded table; look rows up through "
                       "EmbeddingLayer (sharding.lookup), which exchanges indices and rows between the ranks")


def is_sharded(emb):
    return getattr(emb, "_rh_shard", None) is not None


def _world_rank(group):
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if _EMULATE_WORLD > 1 and world == 1:  # single-GPU cost study of an N-rank job: rank 0's shard of N
        world = _EMULATE_WORLD
    return world, rank


def shard_table(emb, group=None):
    """Replace the FULL table held by ``emb`` (identical on every rank) by this rank's shard, in place: the Parameter
    object survives (optimizers / Feature caches keep their reference), its storage becomes (rows / world + 1, D)."""
    if is_sharded(emb):
        return emb._rh_shard
    world, rank = _world_rank(group)
    full = emb.weight.data
    sh = RowShard(full.shape[0], full.shape[1], world, rank, emb.padding_idx, group)
    local = torch.zeros((sh.sink + 1, sh.dim), dtype=full.dtype, device=full.device)
    mine = full[rank::world]
    local[:mine.shape[0]].copy_(mine)
    if sh.pad >= 0 and sh.pad % world == rank:
        local[sh.pad // world].zero_()  # the padding row is zero in the reference too (initializers.py:17-20)
    emb.weight.data = local
    emb.weight.grad = None
    emb._rh_shard = sh
    emb.forward = _refuse_direct_call
    return sh


def shard_tables(model, group=None, min_rows=0):
    """Shard every ``nn.Embedding`` of ``model`` with at least ``min_rows`` rows; returns the sharded modules.
    Call it after the replicas were made equal (DataParallelContext broadcasts rank 0's parameters) and BEFORE the
    optimizer is built (its state is sized from the parameters)."""
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("row-sharded tables need an initialised process group (launch with torchrun)")
    out, seen = [], set()
    for m in model.modules():
        if hasattr(m, "logical_dim") and isinstance(m, nn.Embedding) and m.weight.shape[0] >= min_rows:
            raise NotImplementedError("row-sharded placement of a padded-width table (embed_dim not in 4, 8, 16, 32, 64, "
                                      "128) is not implemented: use tables='replicate' or a kernel width")
        if (isinstance(m, nn.Embedding) and id(m) not in seen and m.weight.shape[0] >= min_rows and
                not getattr(m, "_rh_dense", False)):
            seen.add(id(m))
            shard_table(m, group)
            out.append(m)
    return out


def full_table(emb):
    """The complete (vocab, D) table of a sharded module, assembled on every rank (checkpoints, export)."""
    sh = emb._rh_shard
    local = emb.weight.data[:sh.sink]
    if sh.world == 1 or dist.get_world_size(sh.group) == 1:
        shards = [local] + [torch.zeros_like(local)] * (sh.world - 1)
    else:
        shards = list(all_gather_cat(local.contiguous(), sh.group).chunk(sh.world, dim=0))
    # rank r, local row q  <->  global row q * world + r
    woven = torch.stack(shards, dim=1).reshape(sh.sink * sh.world, sh.dim)
    return woven[:sh.vocab].contiguous()


def full_state_dict(model):
    """``model.state_dict()`` with every sharded table replaced by the complete one: the reference's checkpoint ABI
    (``embedding.embed_dict.<feature>.weight`` of shape (vocab, D)).  Collective when world > 1."""
    sd = model.state_dict()
    for name, m in model.named_modules():
        if isinstance(m, nn.Embedding) and is_sharded(m):
            key = (name + "." if name else "") + "weight"
            if key in sd:
                sd[key] = full_table(m)
    return sd


def load_full_state_dict(model, sd):
    """Load a reference-layout state dict into a model whose tables are sharded (each rank keeps its rows)."""
    sd = dict(sd)
    for name, m in model.named_modules():
        if isinstance(m, nn.Embedding) and is_sharded(m):
            key = (name + "." if name else "") + "weight"
            if key in sd:
                sh = m._rh_shard
                full = sd[key]
                if tuple(full.shape) != (sh.vocab, sh.dim):
                    raise ValueError(f"{key}: expected the full table {(sh.vocab, sh.dim)}, got {tuple(full.shape)}")
                local = torch.zeros_like(m.weight.data)
                mine = full[sh.rank::sh.world]
                local[:mine.shape[0]].copy_(mine)
                sd[key] = local
    return model.load_state_dict(sd)


# -- differentiable collectives over the rows of a batch ------------------------------------------------------------
def _reduce_scatter_sum(x_all, group, out=None):
    world = dist.get_world_size(group)
    B = x_all.shape[0] // max(world, _EMULATE_WORLD if world == 1 else 1)
    if out is None:
        out = torch.empty((B,) + tuple(x_all.shape[1:]), dtype=x_all.dtype, device=x_all.device)
    elif out.shape != (B,) + tuple(x_all.shape[1:]) or not out.is_contiguous():
        raise ValueError("reduce-scatter: bad output buffer")
    x_all = x_all.contiguous()
    if world == 1 and x_all.shape[0] != B:  # RECHUB_EMULATE_WORLD (timing only): the first block stands for the sum
        out.copy_(x_all[:B])
        return out
    try:
        dist.reduce_scatter_tensor(out, x_all, op=dist.ReduceOp.SUM, group=group)
    except (RuntimeError, NotImplementedError):  # gloo has no reduce-scatter: all-reduce, keep the own slice
        full = x_all.clone()
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
        r = dist.get_rank(group)
        out.copy_(full[r * B:(r + 1) * B])
    return out


class _ScatterRows(torch.autograd.Function):
    """(W*B, C) per-rank partial rows of the GLOBAL batch -> (B, C): the sum over ranks of this rank's row block."""

    @staticmethod
    def forward(ctx, x_all, group, box):
        ctx.group = group
        # a persistent output buffer travels in a box, not as a tensor argument: the tensor object handed back to
        # autograd is a fresh alias of it every step (the cached object never acquires a grad_fn)
        return _reduce_scatter_sum(x_all, group, None if box is None else box[0].detach())

    @staticmethod
    def backward(ctx, g):
        return all_gather_cat(g, ctx.group), None, None


class _GatherRows(torch.autograd.Function):
    """(B, C) -> (W*B, C) rows of every rank in rank order; the backward sums what every rank sent back for my rows."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return all_gather_cat(x, group)

    @staticmethod
    def backward(ctx, g_all):
        return _reduce_scatter_sum(g_all, ctx.group), None


def scatter_rows_sum(x_all, group=None, out=None):
    """``out``: write into this (B, C) buffer (a fixed address from step to step; it is returned)."""
    return _ScatterRows.apply(x_all, group, None if out is None else (out,))


def gather_rows(x, group=None):
    return _GatherRows.apply(x, group)


# -- lookups -------------------------------------------------------------------------------------------------------
_desc_cache = ops._DescCache()


class _StaticBuffers(object):
    """(gathered indices, localised indices) per lookup site, keyed by (tables, index buffer address, shape).

    The fused gather finds its index columns through a descriptor table keyed by ADDRESS (ops.EmbedCall.idesc); with
    the batch in the loader's static buffers, reusing one localised-index buffer per site keeps that address -- and so
    the descriptor -- the same from step to step: no per-step descriptor upload, and a hipGraph capture finds every
    descriptor already on the device.  Entries used while capturing are pinned (the graph replays into them)."""

    def __init__(self, cap=64):
        from collections import OrderedDict
        self.cap, self.d, self.pinned = cap, OrderedDict(), {}

    def get(self, key, make):
        b = self.pinned.get(key)
        if b is not None:
            return b
        b = self.d.get(key)
        if b is None:
            b = make()
            self.d[key] = b
            if len(self.d) > self.cap:
                self.d.popitem(last=False)
        else:
            self.d.move_to_end(key)
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            self.pinned[key] = b
        return b


_buffers = _StaticBuffers()


def _is_static(t):
    """True when ``t`` lives in a buffer that keeps its address AND its role from step to step: a DeviceDataLoader
    batch buffer (tagged ``_rh_static``, or a view of one), or anything seen while a hipGraph is being captured (the
    replays read the same addresses by construction)."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return True
    base = t._base if t._base is not None else t
    return bool(getattr(t, "_rh_static", False) or getattr(base, "_rh_static", False))


def _localize(embs, idx, group, static=False):
    """all-gather the (B, F) index matrix and rewrite it for the local shards -> int32 (W*B, F).

    ``static``: the index matrix is a loader-owned buffer, so the per-site buffers (gathered ids, localised ids,
    received rows) are cached by its address (descriptor tables upload once, the step is hipGraph-capturable).  For an
    ordinary temporary (a host DataLoader batch moved to the device, ``.long()`` casts) the address can be recycled
    by the allocator within ONE forward, so two lookups would share -- and overwrite -- each other's buffers: those
    get fresh buffers per call."""
    sh0 = embs[0]._rh_shard
    for e in embs:
        s = e._rh_shard
        if s.world != sh0.world or s.rank != sh0.rank or s.group is not sh0.group:
            raise ValueError("one lookup spans tables sharded over different process groups")
    key = tuple([e._rh_shard.vocab for e in embs] + [e._rh_shard.pad for e in embs] + [e._rh_shard.sink for e in embs])
    desc = _desc_cache.get(key, idx.device)
    n_all = idx.shape[0] * max(dist.get_world_size(group), sh0.world)
    width = sum(e._rh_shard.dim for e in embs)
    # ids travel as int32 when every vocabulary allows it (the loader holds int64, as the reference's encoders emit):
    # half the bytes of the one exchange that grows with the number of ranks on the forward's critical path
    narrow = idx.dtype == torch.int64 and max(e._rh_shard.vocab for e in embs) < 2**31
    wire = torch.int32 if narrow else idx.dtype

    def make():
        return (torch.empty((n_all, idx.shape[1]), dtype=wire, device=idx.device),
                torch.empty((n_all, idx.shape[1]), dtype=torch.int32, device=idx.device),
                torch.empty((idx.shape[0], width), dtype=torch.float32, device=idx.device))

    # (object ids can be recycled after a model is freed: the placement tuple `key` and the width make a stale entry
    # with other shapes impossible)
    site = (tuple(id(e) for e in embs), key, width, n_all, idx.data_ptr(), tuple(idx.shape), tuple(idx.stride()),
            idx.dtype, str(idx.device), sh0.rank)
    idx_all, loc, rows = _buffers.get(site, make) if static else make()
    # saturating: an id beyond int32 must stay out of range (-> RH_FLAG_INDEX_OOB), not wrap onto a valid row
    if narrow and idx.is_cuda and idx.stride(1) == 1:
        # (round 6) narrowed straight into this rank's slice of the gather buffer: the collective then runs IN PLACE -- one
        # launch where clamp, cast and the copy into the buffer were three in the sharded step's head
        world = dist.get_world_size(group)
        r = dist.get_rank(group) if world > 1 else 0
        send = ops.shard_narrow(idx, idx_all[r * idx.shape[0]:(r + 1) * idx.shape[0]])
    else:
        send = idx.clamp(min=-1, max=2**31 - 1).to(torch.int32) if narrow else idx
    all_gather_cat(send, group, out=idx_all)
    return ops.shard_localize(idx_all, desc, sh0.world, sh0.rank, out=loc), rows


def lookup(embs, idx_cols):
    """Rows of F sharded tables for the local batch: ``idx_cols`` = F index tensors (B,) -> (B, F*D) float32.

    ``embs[f]`` is the (sharded) ``nn.Embedding`` of field f; entries may repeat (shared tables, history positions)."""
    group = embs[0]._rh_shard.group
    idx_cols = list(idx_cols)
    loc, rows = _localize(embs, pack_indices(idx_cols), group, static=all(_is_static(c) for c in idx_cols))
    F = len(embs)
    call = ops.EmbedCall([e.weight for e in embs], [e._rh_shard.sink for e in embs], [loc[:, f] for f in range(F)],
                         local_grads=True)
    out_all, _, _ = ops.fused_embedding(call)
    # the received rows land at a fixed address per lookup site: ops.fused_rows (FM / LR / dense append over them)
    # caches its descriptor tables by address
    return scatter_rows_sum(out_all, group, out=rows)


def pooled_lookup(emb, idx, pooling):
    """Sequence feature on a sharded table: idx (B, L) -> (B, D) for "sum" / "mean" (reference SumPooling /
    AveragePooling over InputMask, layers.py:148-161,208-251).  The partial sums of the ranks add up to the masked sum;
    the mean divides by the count of non-sentinel positions + 1e-16, taken from the local indices."""
    sh = emb._rh_shard
    B, L = idx.shape
    loc = _localize([emb], idx.reshape(B * L, 1).contiguous(), sh.group, static=_is_static(idx))[0].view(-1, L)
    part = ops.seq_pool(emb.weight, loc, "sum", sh.sink, local_grads=True)
    total = scatter_rows_sum(part, sh.group)
    if pooling == "sum":
        return total
    count = (idx != (sh.pad if sh.pad >= 0 else -1)).sum(dim=1, keepdim=True).float()
    return total / (count + 1e-16)


TABLES_ENV = "RECHUB_TABLES"


def placement_from_env():
    v = os.environ.get(TABLES_ENV, "replicate")
    if v not in ("replicate", "shard"):
        raise ValueError(f"{TABLES_ENV} must be 'replicate' or 'shard', got {v!r}")
    return v

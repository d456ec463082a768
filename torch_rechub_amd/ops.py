"""Autograd bindings of the gfx950 kernels (C ABI in ``include/rechub_hip.h``).

Each op hands raw device pointers + the current HIP stream to ``librechub_hip.so`` through ctypes.
PyTorch is only the allocator / stream / autograd plumbing here.  There is no CPU path: tensors that
are not on a HIP device raise ``RuntimeError``.

Gradient convention for embedding tables (replaces ``embedding_dense_backward`` + ``model.zero_grad``,
reference trainers/ctr_trainer.py:97-98): every table owns ONE persistent dense gradient buffer
(``grad_buffer(weight)``), zero outside the rows touched since the last optimizer step.  The fused
backward scatter-adds into it and publishes it as ``weight.grad``; ``FusedDenseAdam`` re-zeroes the
touched rows inside its own pass.  A stock ``torch.optim`` optimizer sees an ordinary dense ``.grad``.
"""
import ctypes
import os
from collections import OrderedDict

import torch

from . import _lib

_NULL = ctypes.c_void_p(0)


_empty_stub = {}


def _p(t):
    """Device pointer of a tensor for the C ABI.  An EMPTY tensor has a null data_ptr(); the entry points check their
    pointers before they look at the sizes (and then return at once for B = 0), so empties are passed as a pointer to
    a small per-device stub allocation."""
    if t is None:
        return _NULL
    if t.numel() == 0:
        stub = _empty_stub.get(t.device)
        if stub is None:
            stub = torch.zeros(64, dtype=torch.float32, device=t.device)
            _empty_stub[t.device] = stub
        return ctypes.c_void_p(stub.data_ptr())
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_BRANCH_STREAMS = {}


def run_beside(fn_main, fn_side, side_inputs=()):
    """(fn_main(), fn_side()) with fn_side on a second HIP stream: fork after everything enqueued so far, join before
    returning.  Inside a hipGraph capture the two become concurrent branches of the graph (tools/probe/branch_probe.cpp:
    captured branches do run side by side on this runtime); autograd runs each op's backward on the stream of its forward,
    so the backward forks and joins the same way.  For two independent chains of small kernels (the towers of a two-tower
    model: each launch fills a fraction of the 256 CUs)."""
    cur = torch.cuda.current_stream()
    key = cur.device.index
    side = _BRANCH_STREAMS.get(key)
    if side is None:
        from . import graphs
        side = _BRANCH_STREAMS[key] = graphs.role_stream("branch", cur.device)
    side.wait_stream(cur)
    for t in side_inputs:
        t.record_stream(side)
    with torch.cuda.stream(side):
        b = fn_side()
    a = fn_main()
    cur.wait_stream(side)
    for t in (b if isinstance(b, (tuple, list)) else (b,)):
        if torch.is_tensor(t):
            t.record_stream(cur)
    return a, b


def require_hip(*tensors):
    """Fail loudly for anything that is not a HIP tensor (there is no CPU fallback)."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("torch_rechub_amd: this op runs only on a HIP device (MI355X); got a "
                               f"{t.device} tensor. There is no CPU fallback by design.")
    _lib.load()


# --------------------------------------------------------------------------------------------
# persistent per-table state
# --------------------------------------------------------------------------------------------
def grad_buffer(weight):
    """The persistent dense gradient buffer of a table (allocated zeroed on first use)."""
    buf = getattr(weight, "_rh_grad", None)
    if buf is None or buf.device != weight.device or buf.shape != weight.shape:
        buf = torch.zeros_like(weight, memory_format=torch.contiguous_format)
        weight._rh_grad = buf
        weight._rh_dirty = False
    return buf


def _publish_grad(weight):
    """Make ``weight.grad`` the persistent buffer (folding in a foreign dense grad if one exists)."""
    buf = weight._rh_grad
    g = weight.grad
    if g is None:
        weight.grad = buf
    elif g.data_ptr() != buf.data_ptr():
        buf.add_(g)
        weight.grad = buf
    weight._rh_dirty = True


def _prepare_grad(weight):
    """Called before a scatter: drop stale contents if the user reset ``.grad`` since the last backward."""
    buf = grad_buffer(weight)
    if weight._rh_dirty:
        g = weight.grad
        if g is None or g.data_ptr() != buf.data_ptr():
            buf.zero_()  # zero_grad(set_to_none=True) happened: the buffer holds last step's rows
            weight._rh_dirty = False
    return buf


_err_flags = {}


def err_flag(device):
    """Per-device int32 word the kernels OR error bits into (index out of range ...)."""
    f = _err_flags.get(device)
    if f is None:
        f = torch.zeros(1, dtype=torch.int32, device=device)
        _err_flags[device] = f
    return f


def check_errors(device=None):
    """Synchronising check of the kernel error word; raises IndexError like the reference's CPU path."""
    for dev, f in list(_err_flags.items()):
        if device is not None and dev != device:
            continue
        v = int(f.item())
        if v:
            f.zero_()
            if v & 1:
                raise IndexError("torch_rechub_amd: an embedding index was out of range (index < 0 or >= vocab_size)")
            if v & 64:  # RH_ERR_GATE_TIMEOUT
                raise RuntimeError("torch_rechub_amd: a deferred table sweep waited 2 s for a training step that never started "
                                   "(rh_adam_sweep_gate); the tables may be inconsistent")
            raise RuntimeError(f"torch_rechub_amd: kernel error flag {v}")


class _DescCache(object):
    """LRU of host tuples -> device int64 descriptor tensors (no upload when pointers repeat).

    An entry that was looked up while a hipGraph was being captured is PINNED: the graph keeps reading that device
    tensor on every replay, so it must never be evicted (freed) afterwards."""

    def __init__(self, cap=1024):
        self.cap = cap
        self.d = OrderedDict()
        self.pinned = {}

    def get(self, key, device):
        k = (key, device)
        t = self.pinned.get(k)
        if t is not None:
            return t
        capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        t = self.d.get(k)
        if t is None:
            if capturing:
                raise RuntimeError("torch_rechub_amd: a descriptor table would have to be uploaded during hipGraph "
                                   "capture; run the step eagerly once with the same (static) buffers first")
            t = torch.tensor(key, dtype=torch.int64).to(device)
            self.d[k] = t
            if len(self.d) > self.cap:
                self.d.popitem(last=False)
        else:
            self.d.move_to_end(k)
        if capturing:
            self.pinned[k] = t
        return t


class EmbedCall(object):
    """Everything one fused gather needs besides the autograd inputs: descriptors + shapes.

    weights : table weights per field (a weight may repeat: shared tables)
    pads    : padding_idx per field (-1 = none)
    idx     : index tensors per field, shape (B,), int64 or int32, on the device
    dense   : dense value tensors (B,), float32, appended after the sparse block
    """
    _fcache = _DescCache()
    _icache = _DescCache()
    _dcache = _DescCache()

    def __init__(self, weights, pads, idx, dense=(), want_fm=False, want_lr=False, slots=None, width=None,
                 field_split=0, samples_per_block=0, local_grads=False):
        self.local_grads = bool(local_grads)  # True: never hand the gradient rows to the data-parallel exchange
        self.weights = list(weights)
        self.idx = list(idx)
        self.dense = list(dense)
        F = len(self.weights)
        if F == 0 or len(self.idx) != F or len(pads) != F:
            raise ValueError("EmbedCall: need one weight, padding_idx and index tensor per field")
        require_hip(*self.weights, *self.idx, *self.dense)
        self.F = F
        self.D = int(self.weights[0].shape[1])
        for w in self.weights:
            if w.dim() != 2 or w.shape[1] != self.D or w.dtype != torch.float32 or not w.is_contiguous():
                raise ValueError("EmbedCall: tables must be contiguous float32 (vocab, D) with one common D")
        self.B = int(self.idx[0].shape[0])
        idt = self.idx[0].dtype
        if idt not in (torch.int64, torch.int32):
            raise ValueError(f"EmbedCall: index dtype {idt} unsupported (int64 / int32)")
        for t in self.idx:
            if t.dim() != 1 or t.shape[0] != self.B or t.dtype != idt:
                raise ValueError("EmbedCall: every index tensor must be (B,) with one common integer dtype")
        for t in self.dense:
            if t.dim() != 1 or t.shape[0] != self.B or t.dtype != torch.float32:
                raise ValueError("EmbedCall: dense values must be float32 (B,)")
        self.idx_is_i64 = 1 if idt == torch.int64 else 0
        self.pads = [(-1 if p is None else int(p)) for p in pads]
        self.slots = list(range(F)) if slots is None else list(slots)
        self.dense_col = (max(self.slots) + 1) * self.D
        self.width = self.dense_col + len(self.dense) if width is None else int(width)
        self.want_fm = bool(want_fm)
        self.want_lr = bool(want_lr)
        if self.want_lr and self.slots != list(range(F)):
            raise ValueError("EmbedCall: fused LR needs slot f == field f")
        self.field_split = field_split
        self.samples_per_block = samples_per_block
        self.device = self.weights[0].device

    # descriptor tables ------------------------------------------------------------------
    def fdesc(self, with_grads):
        ptrs = [w.data_ptr() for w in self.weights]
        if with_grads:
            gp = [(_prepare_grad(w).data_ptr() if w.requires_grad else 0) for w in self.weights]
        else:
            gp = [0] * self.F
        key = tuple(ptrs + gp + [int(w.shape[0]) for w in self.weights] + self.pads)
        return EmbedCall._fcache.get(key, self.device)

    def idesc(self):
        key = tuple([t.data_ptr() for t in self.idx] + [t.stride(0) for t in self.idx] + self.slots)
        return EmbedCall._icache.get(key, self.device)

    def ddesc(self):
        if not self.dense:
            return None
        key = tuple([t.data_ptr() for t in self.dense] + [t.stride(0) for t in self.dense])
        return EmbedCall._dcache.get(key, self.device)


# Fusions that have an unfused twin kept for A/B parity tests (tests flip these module attributes; they are not
# environment switches): the head's backward forming the BatchNorm sums of the layer below, the Dice + output-layer pass of
# DIN's attention MLP, BatchNorm + PReLU + Dropout as one launch pair.
FUSE_HEAD_BN = True
FUSE_DICE_HEAD = True
FUSE_BN_PRELU = True


# Lazy optimizers (optim.TableAdam(lazy_k > 1)) listen to two events, through weak references:
#   on_gather(record): rows are about to be READ -> bring them up to date first (a row that was not in recent batches
#                      lags behind the dense semantics until someone looks at it);
#   on_touch(record):  (table, index column) pairs received gradient -> the next step must claim those rows.
# A record keeps the index tensors alive until the listener consumed it.
_lazy_listeners = []


def add_lazy_listener(obj):
    import weakref
    _lazy_listeners.append(weakref.ref(obj))


def _listeners():
    live = [r for r in _lazy_listeners if r() is not None]
    if len(live) != len(_lazy_listeners):
        _lazy_listeners[:] = live
    return [r() for r in live]


def _log_touch(weights, pads, idesc, idx_is_i64, B, F, D, keep):
    for lst in _listeners():
        lst.on_touch(dict(weights=list(weights), pads=list(pads), idesc=idesc, idx_is_i64=idx_is_i64, B=B, F=F, D=D,
                          keep=keep))


def _pre_gather(weights, pads, idesc, idx_is_i64, B, F, D, training=None, keep=None):
    """``training``: the lookup is part of a differentiated forward (a backward / optimizer step follows).  Inside an
    autograd.Function.forward grad mode is off, so the callers pass ctx.needs_input_grad; None = ask grad mode.
    ``keep``: the index tensor(s) behind ``idesc`` -- a listener that keeps the record (TableAdam's refresh-ahead replays the
    previous step's records) keeps the memory the descriptor points at alive with it."""
    if training is None:
        training = torch.is_grad_enabled()
    for lst in _listeners():
        lst.on_gather(dict(weights=list(weights), pads=list(pads), idesc=idesc, idx_is_i64=idx_is_i64, B=B, F=F, D=D,
                           training=bool(training), keep=keep))


# data-parallel exchange hook: set by torch_rechub_amd.distributed when world_size > 1
_sparse_exchange = None
_row_gather = None


def set_sparse_exchange(fn, row_gather=None):
    """fn(call, rows_local (B,F,D)) -> (idx_all (W*B,F) int, rows_all (W*B,F,D)), or None when fn keeps the rows and
    runs the exchange + scatter itself after the backward.  row_gather(t) -> the rows of ``t`` from every rank in rank
    order (used by the sequence-feature backward, which scatters the gathered batch in line).
    set_sparse_exchange(None) disables both."""
    global _sparse_exchange, _row_gather
    _sparse_exchange = fn
    _row_gather = row_gather if fn is not None else None


_pre_backward_hooks = []


def add_pre_embed_backward_hook(fn):
    """fn() runs at the start of every fused-embedding backward (used to kick the dense all-reduce early)."""
    _pre_backward_hooks.append(fn)
    return fn


def remove_pre_embed_backward_hook(fn):
    if fn in _pre_backward_hooks:
        _pre_backward_hooks.remove(fn)


class _EmbedFused(torch.autograd.Function):
    """out (B,width), fm (B,1)|None, lr (B,1)|None = fused_embedding(call, lr_w, lr_b, *table weights)."""

    @staticmethod
    def forward(ctx, call, lr_w, lr_b, *weights):
        B, F, D = call.B, call.F, call.D
        dev = call.device
        # row pitch rounded up to 64 bytes: with 13 dense columns a (B, 429) row is 1716 B, every 64-byte field chunk
        # straddles two cache lines and every store is a partial-line write; at pitch 432 the chunks ARE lines.  The
        # result is the (B, width) view of the padded buffer (row stride 432): GEMMs take a leading dimension.
        pitch = (call.width + 15) // 16 * 16
        out = torch.empty((B, pitch), dtype=torch.float32, device=dev)[:, :call.width]
        fm = torch.empty((B, 1), dtype=torch.float32, device=dev) if call.want_fm else None
        lr = torch.empty((B, 1), dtype=torch.float32, device=dev) if call.want_lr else None
        s_sum = torch.empty((B, D), dtype=torch.float32, device=dev) if call.want_fm else None
        if call.want_lr:
            if lr_w is None or lr_w.numel() != F * D or not lr_w.is_contiguous() or lr_w.dtype != torch.float32:
                raise ValueError("fused LR weight must be contiguous float32 with F*D elements")
            require_hip(lr_w, lr_b)
        ddesc = call.ddesc()
        _pre_gather(call.weights, call.pads, call.idesc(), call.idx_is_i64, B, F, D, training=any(ctx.needs_input_grad),
                    keep=call.idx)
        _lib.call("rh_embed_fwd", _p(call.fdesc(False)), _p(call.idesc()), call.idx_is_i64, B, F, D, _p(ddesc),
                  len(call.dense), call.dense_col, _p(out), out.stride(0), _p(lr_w if call.want_lr else None),
                  _p(lr_b if call.want_lr else None), _p(lr), _p(fm), _p(s_sum), call.field_split,
                  _p(err_flag(dev)), _stream())
        ctx.call = call
        ctx.has_lr_b = lr_b is not None
        ctx.lr_params = (lr_w, lr_b)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(out, s_sum, lr_w if call.want_lr else None)
        return out, fm, lr

    @staticmethod
    def backward(ctx, g_out, g_fm, g_lr):
        call = ctx.call
        out, s_sum, lr_w = ctx.saved_tensors
        B, F, D = call.B, call.F, call.D
        dev = call.device
        for hook in list(_pre_backward_hooks):
            hook()
        if g_out is not None and (g_out.stride(1) != 1 or g_out.stride(0) < call.width):
            g_out = g_out.contiguous()
        g_fm = None if g_fm is None else g_fm.reshape(-1).contiguous()
        g_lr = None if g_lr is None else g_lr.reshape(-1).contiguous()
        any_table = any(w.requires_grad for w in call.weights)
        want_wgrad = g_lr is not None and lr_w is not None and ctx.needs_input_grad[1]
        nchunks = _lib.call("rh_embed_bwd_nchunks", B, call.samples_per_block)
        partial = torch.empty((nchunks, F * D), dtype=torch.float32, device=dev) if want_wgrad else None
        # local_grads: a lookup over row-sharded tables already holds the gradient rows of the global batch
        exchange = _sparse_exchange if (any_table and not call.local_grads) else None
        if any_table or want_wgrad:
            if exchange is None:
                fdesc = call.fdesc(True)
                rows = None
                sink = 0
            else:
                fdesc = call.fdesc(False)
                rows = torch.empty((B, F, D), dtype=torch.float32, device=dev)
                sink = 1
            _lib.call("rh_embed_bwd", _p(fdesc), _p(call.idesc()), call.idx_is_i64, B, F, D, _p(g_out),
                      0 if g_out is None else g_out.stride(0), _p(out), out.stride(0), _p(s_sum), _p(g_fm), _p(g_lr),
                      _p(lr_w), _p(partial), 1.0, sink, _p(rows), call.samples_per_block, _p(err_flag(dev)),
                      _stream())
            if exchange is not None:
                gathered = exchange(call, rows)
                if gathered is not None:  # None: the exchange is deferred to after the backward (split-graph step)
                    scatter_rows(call, gathered[0], gathered[1])
            elif any_table:
                _log_touch(call.weights, call.pads, call.idesc(), call.idx_is_i64, B, F, D, call.idx)
            if any_table:
                for w in {id(w): w for w in call.weights if w.requires_grad}.values():
                    _publish_grad(w)
        want_b = g_lr is not None and ctx.has_lr_b and ctx.needs_input_grad[2]
        wp, bp = ctx.lr_params
        armed = deferred.armed
        if B > 0 and want_wgrad and armed is not None and id(wp) in armed and (not want_b or id(bp) in armed):
            # the per-block partial rows (and, for the bias, the per-sample g_lr) go to the step's packing launch
            g_w = deferred.offer(wp, partial.data_ptr(), nchunks, F * D, F * D, lambda: partial.sum(0).view_as(lr_w), partial)
            g_b = None
            if want_b:
                glc = g_lr.contiguous()
                g_b = deferred.offer(bp, glc.data_ptr(), glc.numel(), 1, 1, lambda: glc.sum().view(1), glc)
            return (None, g_w, g_b) + (None,) * len(call.weights)
        g_w = torch.empty_like(lr_w) if want_wgrad else None
        g_b = torch.empty((1,), dtype=torch.float32, device=dev) if want_b else None
        if B == 0:  # empty batch: nothing was launched; the parameter gradients are exact zeros
            g_w = None if g_w is None else g_w.zero_()
            g_b = None if g_b is None else g_b.zero_()
        elif want_wgrad or want_b:  # one launch: column sums of the per-block LR partials + sum of g_lr
            glc = g_lr.contiguous() if want_b else None
            _lib.call("rh_colsum", _p(partial), nchunks if want_wgrad else 0, F * D, _p(g_w), _p(glc),
                      glc.numel() if want_b else 0, _p(g_b), _stream())
        return (None, g_w, g_b) + (None,) * len(call.weights)


def fused_embedding(call, lr_w=None, lr_b=None):
    """Run the fused gather (+FM, +LR); returns (out, fm, lr) with fm / lr None when not requested."""
    return _EmbedFused.apply(call, lr_w, lr_b, *call.weights)


_row_ids = {}


def _row_index(B, F, device):
    """int32 (B, F) with entry b * F + f: the lookup (b, f) of a (B, F*D) row block read as a (B*F, D) table."""
    key = (B, F, str(device))
    t = _row_ids.get(key)
    if t is None:
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("torch_rechub_amd: run the step eagerly once before capturing a hipGraph")
        t = torch.arange(B * F, dtype=torch.int32, device=device).view(B, F)
        _row_ids[key] = t
    return t


class _RowsFused(torch.autograd.Function):
    """The fused gather's second half on rows that are already in HBM: ``emb`` (B, F*D) -- the result of a row-sharded
    lookup (sharding.lookup) -- goes through the SAME kernels as a (B*F, D) table indexed by b*F + f, so that one launch
    emits the flattened MLP input (sparse block + dense values, Q1), the FM scalar and the LR scalar, and one launch
    computes the gradient of all three with respect to every row (``sink = 1``: per-lookup gradient rows, the form the
    row exchange carries) plus the per-block LR weight partials.  Keep ``emb`` at a fixed address from step to step
    (sharding.lookup does): the descriptor tables are cached by address."""

    @staticmethod
    def forward(ctx, emb, F, dense, lr_w, lr_b, want_fm):
        require_hip(emb, *dense)
        if emb.dim() != 2 or emb.dtype != torch.float32 or not emb.is_contiguous() or emb.shape[1] % F:
            raise ValueError("fused_rows: rows must be a contiguous float32 (B, F*D) matrix")
        B, D = int(emb.shape[0]), int(emb.shape[1]) // F
        dev = emb.device
        want_lr = lr_w is not None
        if want_lr:
            if lr_w.numel() != F * D or not lr_w.is_contiguous() or lr_w.dtype != torch.float32:
                raise ValueError("fused LR weight must be contiguous float32 with F*D elements")
            require_hip(lr_w, lr_b)
        for t in dense:
            if t.dim() != 1 or t.shape[0] != B or t.dtype != torch.float32:
                raise ValueError("fused_rows: dense values must be float32 (B,)")
        ids = _row_index(B, F, dev)
        fdesc = EmbedCall._fcache.get(tuple([emb.data_ptr()] * F + [0] * F + [B * F] * F + [-1] * F), dev)
        idesc = EmbedCall._icache.get(tuple([ids.data_ptr() + 4 * f for f in range(F)] + [F] * F + list(range(F))), dev)
        ddesc = None
        if dense:
            ddesc = EmbedCall._dcache.get(tuple([t.data_ptr() for t in dense] + [t.stride(0) for t in dense]), dev)
        out = torch.empty((B, F * D + len(dense)), dtype=torch.float32, device=dev)
        fm = torch.empty((B, 1), dtype=torch.float32, device=dev) if want_fm else None
        lr = torch.empty((B, 1), dtype=torch.float32, device=dev) if want_lr else None
        s_sum = torch.empty((B, D), dtype=torch.float32, device=dev) if want_fm else None
        _lib.call("rh_embed_fwd", _p(fdesc), _p(idesc), 0, B, F, D, _p(ddesc), len(dense), F * D, _p(out), out.stride(0),
                  _p(lr_w), _p(lr_b if want_lr else None), _p(lr), _p(fm), _p(s_sum), 0, _p(err_flag(dev)), _stream())
        ctx.meta = (B, F, D, fdesc, idesc, lr_b is not None)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(out, s_sum, lr_w)
        return out, fm, lr

    @staticmethod
    def backward(ctx, g_out, g_fm, g_lr):
        out, s_sum, lr_w = ctx.saved_tensors
        B, F, D, fdesc, idesc, has_b = ctx.meta
        dev = out.device
        for hook in list(_pre_backward_hooks):
            hook()
        if g_out is not None and (g_out.stride(1) != 1 or g_out.stride(0) < out.shape[1]):
            g_out = g_out.contiguous()
        g_fm = None if g_fm is None else g_fm.reshape(-1).contiguous()
        g_lr = None if g_lr is None else g_lr.reshape(-1).contiguous()
        want_wgrad = g_lr is not None and lr_w is not None and ctx.needs_input_grad[3]
        want_b = g_lr is not None and has_b and ctx.needs_input_grad[4]
        nchunks = _lib.call("rh_embed_bwd_nchunks", B, 0)
        partial = torch.empty((nchunks, F * D), dtype=torch.float32, device=dev) if want_wgrad else None
        rows = torch.empty((B, F * D), dtype=torch.float32, device=dev)
        g_w = torch.empty_like(lr_w) if want_wgrad else None
        g_b = torch.empty((1,), dtype=torch.float32, device=dev) if want_b else None
        if B == 0:
            g_w = None if g_w is None else g_w.zero_()
            g_b = None if g_b is None else g_b.zero_()
            return rows, None, None, g_w, g_b, None
        _lib.call("rh_embed_bwd", _p(fdesc), _p(idesc), 0, B, F, D, _p(g_out), 0 if g_out is None else g_out.stride(0),
                  _p(out), out.stride(0), _p(s_sum), _p(g_fm), _p(g_lr), _p(lr_w), _p(partial), 1.0, 1, _p(rows), 0,
                  _p(err_flag(dev)), _stream())
        if want_wgrad or want_b:
            glc = g_lr.contiguous() if want_b else None
            _lib.call("rh_colsum", _p(partial), nchunks if want_wgrad else 0, F * D, _p(g_w), _p(glc),
                      glc.numel() if want_b else 0, _p(g_b), _stream())
        return rows, None, None, g_w, g_b, None


def fused_rows(emb, n_fields, dense=(), lr_w=None, lr_b=None, want_fm=False):
    """(out (B, F*D + n_dense), fm (B,1)|None, lr (B,1)|None) from rows already gathered into ``emb`` (B, F*D)."""
    return _RowsFused.apply(emb, int(n_fields), tuple(dense), lr_w, lr_b, bool(want_fm))


def scatter_rows(call, idx_all, rows_all):
    """Scatter-add gradient rows (N,F,D) for packed indices idx_all (N,F) into the tables' grad buffers."""
    require_hip(idx_all, rows_all)
    N = int(idx_all.shape[0])
    F, D = call.F, call.D
    ptrs = [idx_all.data_ptr() + f * idx_all.element_size() for f in range(F)]
    key = tuple(ptrs + [idx_all.stride(0)] * F + list(range(F)))
    idesc = EmbedCall._icache.get(key, call.device)
    _lib.call("rh_embed_scatter_rows", _p(call.fdesc(True)), _p(idesc), 1 if idx_all.dtype == torch.int64 else 0,
              N, F, D, _p(rows_all), 1.0, call.samples_per_block, _p(err_flag(call.device)), _stream())
    _log_touch(call.weights, call.pads, idesc, 1 if idx_all.dtype == torch.int64 else 0, N, F, D, [idx_all])


# --------------------------------------------------------------------------------------------
class _FMFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, reduce_sum):
        require_hip(x)
        if x.dim() != 3 or x.dtype != torch.float32:
            raise ValueError("FM expects a float32 (B, num_features, embed_dim) tensor")
        if x.stride(2) != 1 or x.stride(1) != x.shape[2]:
            x = x.contiguous()
        B, F, D = x.shape
        out = torch.empty((B, 1) if reduce_sum else (B, D), dtype=torch.float32, device=x.device)
        _lib.call("rh_fm_fwd", _p(x), x.stride(0), B, F, D, 1 if reduce_sum else 0, _p(out), _stream())
        ctx.reduce_sum = reduce_sum
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        B, F, D = x.shape
        g = g.contiguous()
        gx = torch.empty((B, F, D), dtype=torch.float32, device=x.device)
        _lib.call("rh_fm_bwd", _p(x), x.stride(0), B, F, D, 1 if ctx.reduce_sum else 0, _p(g), _p(gx), gx.stride(0),
                  _stream())
        return gx, None


def fm(x, reduce_sum=True):
    return _FMFn.apply(x, reduce_sum)


# --------------------------------------------------------------------------------------------
_POOL_MODES = {"sum": 0, "mean": 1, "concat": 2}


class _SeqPoolFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, weight, idx, mode, sentinel, padding_idx, local_grads=False):
        ctx.local_grads = local_grads
        require_hip(weight, idx)
        if idx.dim() != 2 or idx.dtype not in (torch.int64, torch.int32):
            raise ValueError("sequence feature values must be an integer (B, L) tensor")
        B, L = idx.shape
        V, D = weight.shape
        out = torch.empty((B, L, D) if mode == 2 else (B, D), dtype=torch.float32, device=weight.device)
        if _lazy_listeners:
            flat = idx.reshape(-1) if idx.is_contiguous() else idx.contiguous().view(-1)
            # (the padding positions of a post-padded history all carry padding_idx: the refresh skips those lookups and
            # keeps the padding row itself current once per launch -- csrc/optim.hip)
            _pre_gather([weight], [padding_idx], EmbedCall._icache.get((flat.data_ptr(), 1, 0), weight.device),
                        1 if idx.dtype == torch.int64 else 0, B * L, 1, D, training=any(ctx.needs_input_grad), keep=[flat])
        _lib.call("rh_seq_pool_fwd", _p(weight), V, _p(idx), 1 if idx.dtype == torch.int64 else 0, idx.stride(0),
                  idx.stride(1), B, L, D, mode, sentinel, _p(out), out.stride(0), _p(err_flag(weight.device)),
                  _stream())
        ctx.meta = (mode, sentinel, -1 if padding_idx is None else int(padding_idx))
        ctx.weight = weight
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        weight = ctx.weight
        mode, sentinel, pad = ctx.meta
        if weight.requires_grad:
            g = g.contiguous()
            if _row_gather is not None and not ctx.local_grads:
                # replicated table under data parallelism: every replica applies the gradient rows of the global batch
                idx, g = _row_gather(idx.contiguous()), _row_gather(g)
            B, L = idx.shape
            V, D = weight.shape
            buf = _prepare_grad(weight)
            _lib.call("rh_seq_pool_bwd", _p(buf), V, _p(idx), 1 if idx.dtype == torch.int64 else 0, idx.stride(0),
                      idx.stride(1), B, L, D, mode, sentinel, pad, _p(g), g.stride(0), 1.0,
                      _p(err_flag(weight.device)), _stream())
            _publish_grad(weight)
            if _lazy_listeners:  # the (B, L) positions are B*L lookups of one field
                flat = idx.reshape(-1) if idx.is_contiguous() else idx.contiguous().view(-1)
                idesc = EmbedCall._icache.get((flat.data_ptr(), 1, 0), weight.device)
                _log_touch([weight], [pad], idesc, 1 if idx.dtype == torch.int64 else 0, B * L, 1, D, [flat])
        return None, None, None, None, None, None


def seq_pool(weight, idx, pooling, padding_idx, local_grads=False):
    """Gather + masked pooling of one sequence feature (mask sentinel = padding_idx, or -1 when unset).
    ``local_grads``: the caller already feeds the gradient of the global batch (row-sharded tables)."""
    if pooling not in _POOL_MODES:
        raise ValueError("Sequence pooling method supports only pooling in %s, got %s." % (["sum", "mean"], pooling))
    sentinel = -1 if padding_idx is None else int(padding_idx)
    return _SeqPoolFn.apply(weight, idx, _POOL_MODES[pooling], sentinel, padding_idx, local_grads)


# --------------------------------------------------------------------------------------------
class _CrossFn(torch.autograd.Function):
    """CrossNetwork: x_{l+1} = x0 * (w_l . x_l) + b_l + x_l, W and Bv stacked (L, d)."""

    @staticmethod
    def forward(ctx, x, W, Bv):
        require_hip(x, W, Bv)
        if x.dim() != 2 or x.dtype != torch.float32:
            raise ValueError("CrossNetwork expects a float32 (B, d) tensor")
        if x.stride(1) != 1:
            x = x.contiguous()
        W = W.contiguous()
        Bv = Bv.contiguous()
        B, d = x.shape
        L = W.shape[0]
        seg = _lib.call("rh_cross_max_layers", d)
        if seg <= 0:
            raise ValueError(f"CrossNetwork width {d} unsupported by the HIP kernel (1..2048)")
        xs = [x]
        cur = x
        for l0 in range(0, L, seg):
            n = min(seg, L - l0)
            out = torch.empty((B, d), dtype=torch.float32, device=x.device)
            _lib.call("rh_cross_fwd", _p(x), x.stride(0), _p(cur), cur.stride(0), _p(W[l0:l0 + n]), _p(Bv[l0:l0 + n]),
                      B, d, n, _p(out), out.stride(0), _stream())
            cur = out
            xs.append(cur)
        ctx.seg = seg
        ctx.save_for_backward(W, Bv, *xs[:-1])
        return cur

    @staticmethod
    def backward(ctx, g):
        W, Bv, *xs = ctx.saved_tensors
        x = xs[0]
        B, d = x.shape
        L = W.shape[0]
        seg = ctx.seg
        g = g.contiguous()
        if B == 0:
            return g, torch.zeros_like(W), torch.zeros_like(Bv)
        gW = torch.empty_like(W)
        gB = torch.empty_like(Bv)
        nblocks = _lib.call("rh_cross_bwd_nblocks", B)
        starts = list(range(0, L, seg))
        gx0_total = None
        for si in reversed(range(len(starts))):
            l0 = starts[si]
            n = min(seg, L - l0)
            cur = xs[si]
            first = si == 0
            partial = torch.empty((nblocks, 2, n, d), dtype=torch.float32, device=x.device)
            gx = torch.empty((B, d), dtype=torch.float32, device=x.device)
            gx0 = None if first else torch.empty((B, d), dtype=torch.float32, device=x.device)
            _lib.call("rh_cross_bwd", _p(x), x.stride(0), _p(cur), cur.stride(0), _p(W[l0:l0 + n]), _p(Bv[l0:l0 + n]),
                      B, d, n, _p(g), g.stride(0), _p(gx0), _p(gx), gx.stride(0), 1 if first else 0, _p(partial),
                      _stream())
            red = partial.sum(0)
            gW[l0:l0 + n] = red[0]
            gB[l0:l0 + n] = red[1]
            if not first:
                gx0_total = gx0 if gx0_total is None else gx0_total + gx0
            g = gx
        if gx0_total is not None:
            g = g + gx0_total
        return g, gW, gB


def cross_network(x, W, Bv):
    return _CrossFn.apply(x, W, Bv)


# --------------------------------------------------------------------------------------------
_rng_state = {}


def _dropout_rng(device):
    """Device-resident (seed, call counter) of the fused dropout; seeded from torch's generator on first use."""
    st = _rng_state.get(device)
    if st is None:
        st = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0, 0, 0], dtype=torch.int64).to(device)
        _rng_state[device] = st
    return st


class _BnReluDropoutFn(torch.autograd.Function):
    """y = dropout(relu(batch_norm(h))) for one MLP hidden layer (training mode); see csrc/mlp.hip.
    ``stats`` / ``stats_ctr``: per-slab (sum, M2) of h and the dropout counter drawn by the GEMM that produced h
    (ops.linear_stats), or None."""

    @staticmethod
    def forward(ctx, h, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, p_drop, stats,
                stats_ctr, relu):
        require_hip(h, gamma, beta)
        h = h.contiguous()
        B, C = h.shape
        dev = h.device
        out = torch.empty_like(h)
        if stats is not None:
            partial, prow = stats, _lib.call("rh_gemm_stats_rows", B, C)  # the slab height rh_linear_fwd used
            if tuple(stats.shape) != (-(-B // prow), 2, C):
                raise ValueError("BatchNorm statistics slabs do not match the activations")
        else:
            partial = torch.empty((_lib.call("rh_bn_act_nchunks", B), 2, C), dtype=torch.float32, device=dev)
            prow = 0
        stat = torch.empty((4, C), dtype=torch.float32, device=dev)
        # with GEMM-provided statistics the GEMM already drew this call's dropout counter (and counted the batch)
        saved_ctr = stats_ctr if stats is not None else torch.empty(1, dtype=torch.int64, device=dev)
        rng = _dropout_rng(dev)
        _lib.call("rh_bn_relu_dropout_fwd", _p(h), B, C, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                  _p(num_batches_tracked), float(momentum), float(eps), float(p_drop), 1, _p(rng), _p(saved_ctr),
                  _p(partial), prow, _p(stat), _p(out), 1 if relu else 0, _stream())
        ctx.p_drop = float(p_drop)
        ctx.relu = bool(relu)
        ctx.save_for_backward(h, gamma, beta, stat, saved_ctr)
        return out

    @staticmethod
    def backward(ctx, dy):
        h, gamma, beta, stat, saved_ctr = ctx.saved_tensors
        B, C = h.shape
        dev = h.device
        dy = dy.contiguous()
        dx = torch.empty_like(h)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(beta)
        pre = getattr(ctx, "rh_pre", None)
        ctx.rh_pre = None
        if pre is not None and pre[0].data_ptr() == dy.data_ptr() and pre[0].shape == dy.shape and \
                dy._version == pre[1]:
            # dy is exactly the tensor the output head's backward produced (the head was the only consumer of this
            # layer): its launch already summed (g1, g1 * xhat) per column -> finalize + apply only.  ``pre`` HOLDS that
            # tensor: autograd can then neither sum a second consumer's gradient into it in place (InputBuffer only
            # reuses a buffer it owns alone -> the sum is a new tensor with another address) nor recycle its memory
            # for another gradient; the version counter covers any other in-place writer.
            _lib.call("rh_bn_relu_dropout_bwd_pre", _p(h), _p(dy), B, C, _p(gamma), _p(beta), ctx.p_drop,
                      _p(_dropout_rng(dev)), _p(saved_ctr), _p(pre[2]), pre[3], _p(stat), _p(dx), _p(dgamma), _p(dbeta),
                      1 if ctx.relu else 0, _stream())
            return dx, dgamma, dbeta, None, None, None, None, None, None, None, None, None
        partial = torch.empty((_lib.call("rh_bn_act_nchunks", B), 2, C), dtype=torch.float32, device=dev)
        _lib.call("rh_bn_relu_dropout_bwd", _p(h), _p(dy), B, C, _p(gamma), _p(beta), ctx.p_drop, _p(_dropout_rng(dev)),
                  _p(saved_ctr), _p(partial), _p(stat), _p(dx), _p(dgamma), _p(dbeta), 1 if ctx.relu else 0, _stream())
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None, None


class _BnPreluDropoutFn(torch.autograd.Function):
    """y = dropout(prelu(batch_norm(h))) for one hidden layer of a two-tower MLP (training mode; one slope, nn.PReLU())."""

    @staticmethod
    def forward(ctx, h, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, p_drop, slope):
        require_hip(h, gamma, beta, slope)
        h = h.contiguous()
        B, C = h.shape
        dev = h.device
        out = torch.empty_like(h)
        partial = torch.empty((_lib.call("rh_bn_act_nchunks", B), 2, C), dtype=torch.float32, device=dev)
        stat = torch.empty((4, C), dtype=torch.float32, device=dev)
        saved_ctr = torch.empty(1, dtype=torch.int64, device=dev)
        _lib.call("rh_bn_prelu_dropout_fwd", _p(h), B, C, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                  _p(num_batches_tracked), float(momentum), float(eps), float(p_drop), 1, _p(_dropout_rng(dev)), _p(saved_ctr),
                  _p(partial), 0, _p(stat), _p(out), _p(slope), _stream())
        ctx.p_drop = float(p_drop)
        ctx.param = slope
        ctx.save_for_backward(h, gamma, beta, stat, saved_ctr, slope)
        return out

    @staticmethod
    def backward(ctx, dy):
        h, gamma, beta, stat, saved_ctr, slope = ctx.saved_tensors
        B, C = h.shape
        dev = h.device
        dy = dy.contiguous()
        dx = torch.empty_like(h)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(beta)
        partial = torch.empty((_lib.call("rh_bn_act_nchunks", B), 2, C), dtype=torch.float32, device=dev)
        nb = _lib.call("rh_bn_prelu_nblocks", B, C)
        sp = torch.empty((nb,), dtype=torch.float32, device=dev)
        _lib.call("rh_bn_prelu_dropout_bwd", _p(h), _p(dy), B, C, _p(gamma), _p(beta), ctx.p_drop, _p(_dropout_rng(dev)),
                  _p(saved_ctr), _p(partial), _p(stat), _p(dx), _p(dgamma), _p(dbeta), _p(slope), _p(sp), _stream())
        g_slope = deferred.offer(ctx.param, sp.data_ptr(), nb, 1, 1, lambda: sp.sum().reshape(1), sp)
        return dx, dgamma, dbeta, None, None, None, None, None, None, g_slope


def bn_prelu_dropout_ok(h, bn, act):
    return (bn.training and type(act) is torch.nn.PReLU and act.weight.numel() == 1 and h.is_cuda and
            h.dtype == torch.float32 and h.dim() == 2 and h.shape[0] > 1 and FUSE_BN_PRELU)


def bn_prelu_dropout(h, bn, act, p_drop):
    """Fused BatchNorm1d + nn.PReLU() + Dropout of one hidden layer (training mode), driven by the modules."""
    return _BnPreluDropoutFn.apply(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                   bn.momentum, bn.eps, p_drop, act.weight)


def bn_relu_dropout(h, bn, p_drop, stats=None, relu=True):
    """Fused BatchNorm1d + ReLU + Dropout of one MLP hidden layer, driven by the nn.BatchNorm1d module ``bn``
    (``relu=False``: BatchNorm1d + Dropout only, for layers whose activation is Dice / PReLU / ...).
    ``stats``: the (statistics, counter) pair ops.linear_stats returned for this very ``bn``, or None."""
    if bn.training:
        st, ctr = stats if stats is not None else (None, None)
        return _BnReluDropoutFn.apply(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                      bn.momentum, bn.eps, p_drop, st, ctr, relu)
    require_hip(h)
    h = h.contiguous()
    out = torch.empty_like(h)
    _lib.call("rh_bn_relu_dropout_fwd", _p(h), h.shape[0], h.shape[1], _p(bn.weight), _p(bn.bias), _p(bn.running_mean),
              _p(bn.running_var), _p(None), 0.0, float(bn.eps), 0.0, 0, _p(None), _p(None), _p(None), 0, _p(None), _p(out),
              1 if relu else 0, _stream())
    return out


# --------------------------------------------------------------------------------------------
# MLP GEMM backward (weight gradient), output head and loss (csrc/linear.hip)
_MAX_WGRAD_TILES = 4096


def _offer_wgrad_slabs(weight, bias, partial, B):
    """Register the per-split slabs rh_linear_wgrad_partial(_group) left in ``partial`` -- S x (N, K), then S x (N,) -- as
    the gradient sources of ``weight`` / ``bias`` (None: no bias) with the armed ops.deferred."""
    N, K = weight.shape
    S = _lib.call("rh_linear_wgrad_splits", B, N, K)

    def reduce_w():
        flush_wgrad_rider()  # (the slabs may still be waiting for the optimizer's end-of-step launch)
        return partial[:S * N * K].view(S, N, K).sum(0)

    def reduce_b():
        flush_wgrad_rider()
        return partial[S * N * K:S * N * K + S * N].view(S, N).sum(0)

    dW = deferred.offer(weight, partial.data_ptr(), S, N * K, N * K, reduce_w, partial)
    db = None
    if bias is not None:
        db = deferred.offer(bias, partial.data_ptr() + 4 * S * N * K, S, N, N, reduce_b, partial)
    return dW, db


def linear_wgrad(g, x, want_bias=True, weight=None, bias=None):
    """(dW (N, K), db (N,) | None) = (g^T x, colsum(g)) for g (B, N), x (B, K): split-batch f32 MFMA kernel.

    ``weight`` / ``bias``: the parameters these are the gradients of.  While the trainer's fast path has armed
    ``ops.deferred`` for them, the per-split slabs are NOT reduced here: they are registered and the function returns
    None for that gradient (the step's packing launch sums them into the flat bucket)."""
    require_hip(g, x)
    if g.stride(1) != 1:
        g = g.contiguous()
    if x.stride(1) != 1:
        x = x.contiguous()
    B, N = g.shape
    K = x.shape[1]
    dev = g.device
    if B == 0:
        return (torch.zeros((N, K), dtype=torch.float32, device=dev),
                torch.zeros((N,), dtype=torch.float32, device=dev) if want_bias else None)
    partial = torch.empty(_lib.call("rh_linear_wgrad_workspace", B, N, K), dtype=torch.float32, device=dev)
    armed = deferred.armed
    if armed is not None and weight is not None and id(weight) in armed and (not want_bias or
                                                                              (bias is not None and id(bias) in armed)):
        # (step-ahead graph being captured: the slabs feed nothing but the packing launch at the end of the step, so the launch
        # rides in the optimizer's end-of-step launch with the other weight gradients of the step -- wgrad_rider below)
        rider = wgrad_rider
        if not (rider is not None and B < _WGRAD_GROUP_MAX_B and rider([(g, g.stride(0), x, x.stride(0), N, K, partial)], B)):
            _lib.call("rh_linear_wgrad_partial", _p(g), g.stride(0), _p(x), x.stride(0), B, N, K, _p(partial), _stream())
        return _offer_wgrad_slabs(weight, bias if want_bias else None, partial, B)
    dW = torch.empty((N, K), dtype=torch.float32, device=dev)
    db = torch.empty((N,), dtype=torch.float32, device=dev) if want_bias else None
    _lib.call("rh_linear_wgrad", _p(g), g.stride(0), _p(x), x.stride(0), B, N, K, _p(dW), _p(db), _p(partial),
              _stream())
    return dW, db


def linear_ok(x, weight):
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32 and
            _lib.call("rh_linear_wgrad_tiles", weight.shape[0], weight.shape[1]) <= _MAX_WGRAD_TILES)


# Forward / input-gradient GEMMs: hipBLASLt by default.  RECHUB_OWN_GEMM=1 (or "fwd": forward only) routes batch-sized
# problems (M <= 16384, N, K <= 1024) through the f32-MFMA tile kernel of csrc/gemm.hip, whose epilogue also emits the
# BatchNorm statistics (one BN launch less per layer).  Measured on MI355X at B = 4096 (round 1, same box A/B): the
# whole step is within +-0.5 % of the library path (0.4331 vs 0.4325 ms) -- parity, not yet a win, so it stays opt-in
# (DESIGN.md 3.8).
_GEMM_MAX_M = 16384 if os.environ.get("RECHUB_OWN_GEMM", "0") in ("1", "fwd") else 0
_GEMM_MAX_NK = 1024
_GEMM_DGRAD = os.environ.get("RECHUB_OWN_GEMM", "0") != "fwd"  # "fwd": own forward GEMMs only


def _own_gemm(M, N, K):
    return 0 < M <= _GEMM_MAX_M and N <= _GEMM_MAX_NK and K <= _GEMM_MAX_NK


# Round 6: the FORWARD of an nn.Linear over (B L) rows (DIN's attention MLP: 409 600 x 256 -> 128) on the tile kernel too -- on
# those shapes it is faster than the library GEMM it replaced (configs[3] 5.02-5.17 -> 4.91-4.92 ms per step, same box; the input
# gradient is not: 5.18-5.44 with both), and its epilogue hands the BatchNorm that follows the per-slab statistics of the output
# (no statistics pass over the (B L, C) tensor).  RECHUB_AB=tallgemm=0: the library GEMM + the statistics pass.
TALL_GEMM_FWD = _lib.ab("tallgemm")
_TALL_MIN_M = 65536
_STATS_ONLY = object()  # _LinearFn: per-slab statistics of the output WITHOUT the BatchNorm bookkeeping of rh_linear_fwd


def _tall_fwd(M, N, K):
    # (the measured territory: the attention MLP's widths; wide layers stay with the library's larger tiles)
    return TALL_GEMM_FWD and M >= _TALL_MIN_M and N <= 256 and K <= 512


class _LinearFn(torch.autograd.Function):
    """nn.Linear: forward and input gradient on the f32-MFMA tile kernel at CTR batch sizes (library GEMM otherwise),
    weight + bias gradient on the split-batch MFMA kernel.  With ``want_stats`` the forward also returns the per-slab
    (sum, M2) of its output for the BatchNorm that follows."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn_batches):
        want_stats = bn_batches is not None
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)  # the parameter objects themselves (ops.deferred keys gradients by identity)
        M, K = x.shape
        N = weight.shape[0]
        stats = ctr = None
        stats_only = bn_batches is _STATS_ONLY
        if (_own_gemm(M, N, K) or _tall_fwd(M, N, K)) and weight.is_contiguous() and x.stride(1) == 1:
            y = torch.empty((M, N), dtype=torch.float32, device=x.device)
            rng = None
            if want_stats:
                rows = _lib.call("rh_gemm_stats_rows", M, N)
                stats = torch.empty((-(-M // rows), 2, N), dtype=torch.float32, device=x.device)
                if not stats_only:
                    ctr = torch.empty(1, dtype=torch.int64, device=x.device)
                    rng = _dropout_rng(x.device)
            _lib.call("rh_linear_fwd", _p(x), x.stride(0), _p(weight), K, _p(bias), M, N, K, _p(y), N, _p(stats), _p(rng),
                      _p(ctr), _p(bn_batches if want_stats and not stats_only else None), _stream())
        else:
            y = torch.nn.functional.linear(x, weight, bias)
        if want_stats:
            if stats is None:
                return y, None, None
            if stats_only:
                ctx.mark_non_differentiable(stats)
                return y, stats, None
            ctx.mark_non_differentiable(stats, ctr)
            return y, stats, ctr
        return y

    @staticmethod
    def backward(ctx, g, *unused):
        x, weight = ctx.saved_tensors
        gx = None
        if ctx.needs_input_grad[0]:
            M, N = g.shape
            K = weight.shape[1]
            if _GEMM_DGRAD and _own_gemm(M, K, N) and weight.is_contiguous() and g.stride(1) == 1:
                gx = torch.empty((M, K), dtype=torch.float32, device=g.device)
                _lib.call("rh_linear_dgrad", _p(g), g.stride(0), _p(weight), K, M, N, K, _p(gx), K, _stream())
            else:
                gx = g.mm(weight)
        dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW, db = linear_wgrad(g, x, want_bias=ctx.has_bias, weight=ctx.params[0], bias=ctx.params[1])
        return gx, dW, db, None


def linear_chunk_stats(x, weight, bias=None):
    """(h, chunk_stats, chunk_rows) for a Linear in front of a training-mode BatchNorm1d whose statistics the consumer finalises
    itself (``bn_dice`` / ``bn_dice_head`` with ``chunk_stats``: rh_bn_stats_from_partial does the BatchNorm bookkeeping): on
    (B L)-row inputs the tile GEMM's epilogue emits the per-slab (sum, M2) of h; otherwise (h, None, 0)."""
    if linear_ok(x, weight) and x.dim() == 2 and _tall_fwd(x.shape[0], weight.shape[0], weight.shape[1]) and \
            weight.is_contiguous() and x.stride(1) == 1:
        h, st, _ = _LinearFn.apply(x, weight, bias, _STATS_ONLY)
        if st is not None:
            return h, st, _lib.call("rh_gemm_stats_rows", int(x.shape[0]), int(weight.shape[0]))
        return h, None, 0
    return linear(x, weight, bias), None, 0


def linear(x, weight, bias=None):
    """F.linear for 2-D fp32 HIP activations; see _LinearFn."""
    if linear_ok(x, weight) and (_own_gemm(x.shape[0], weight.shape[0], weight.shape[1]) or _tall_fwd(
            x.shape[0], weight.shape[0], weight.shape[1]) or (
            torch.is_grad_enabled() and (weight.requires_grad or (bias is not None and bias.requires_grad)))):
        return _LinearFn.apply(x, weight, bias, None)
    return torch.nn.functional.linear(x, weight, bias)


def linear_stats(x, weight, bias, bn):
    """The Linear in front of the training-mode BatchNorm1d ``bn``: returns (h, stats) where stats is None or the pair
    (per-slab (sum, M2) of h, dropout counter) to hand to ``bn_relu_dropout(h, bn, p, stats=stats)`` -- the GEMM then
    has already counted the batch for ``bn`` and drawn the dropout counter of that call."""
    if linear_ok(x, weight):
        h, st, ctr = _LinearFn.apply(x, weight, bias, bn.num_batches_tracked)
        return h, (None if st is None else (st, ctr))
    return torch.nn.functional.linear(x, weight, bias), None


def _col(t, B):
    """(B,) contiguous view of a (B,) / (B, 1) tensor."""
    if t is None:
        return None
    if t.numel() != B:
        raise ValueError(f"head term of {tuple(t.shape)} does not match batch {B}")
    return t.reshape(B).contiguous()


class DeferredGrads(object):
    """Parameter gradients that exist only as per-block / per-split partial slabs until the step's ONE packing launch
    (rh_pack_grads) sums them straight into the flat gradient bucket.

    Armed by the single-GPU fast path of the trainers for the parameters of their bucket.  While armed, the backward of
    the weight-gradient, head and fused-embedding kernels skips its trailing reduction launch, registers
    ``(slab, rows, row stride)`` for the parameter and hands autograd ``None`` for it.  A parameter that is used twice in
    one step (second registration) falls back to the immediate reduction for BOTH uses; a gradient produced by any
    other op (``p.grad`` set by autograd) is added on top by the packing launch."""

    def __init__(self):
        self.armed = None  # {id(param): param} of the parameters whose gradients may be deferred
        self.items = {}    # id(param) -> dict(param, slab, nparts, stride, numel, reduce_now)

    def arm(self, params, root=None):
        """``root``: the tensor ``backward()`` is about to be called on.  Given (the data-parallel step: its bucket packs slabs
        and starts their all-reduce IN THE MIDDLE of the backward, DenseGradBucket.flush from the pre-embedding-backward hook),
        the uses of every armed parameter in the autograd graph are counted, and ``final(p)`` says whether all of them have
        reported -- a slab registered by the first of two uses (a Linear shared by two towers, with an embedding backward in
        between) is not the parameter's gradient yet (round-5 advisor finding: the second contribution was lost)."""
        self.armed = {id(p): p for p in params}
        self.items = {}
        self.uses = None
        if root is not None and root.grad_fn is not None:
            self.uses = self._count_uses(root.grad_fn, self.armed)

    @staticmethod
    def _count_uses(root_fn, want):
        """{id(param): edges into its AccumulateGrad node} over the graph below ``root_fn`` (one edge per use of the leaf)."""
        counts, seen, stack = {}, set(), [root_fn]
        while stack:
            fn = stack.pop()
            if fn in seen:
                continue
            seen.add(fn)
            for nxt, _ in fn.next_functions:
                if nxt is None:
                    continue
                v = getattr(nxt, "variable", None)  # AccumulateGrad
                if v is not None:
                    if id(v) in want:
                        counts[id(v)] = counts.get(id(v), 0) + 1
                else:
                    stack.append(nxt)
        return counts

    uses = None

    def final(self, param):
        """Have all uses of ``param`` in this backward produced their gradient?  (True when uses are not tracked: the
        single-GPU step packs once, after the backward.)"""
        return self.uses is None or self.uses.get(id(param), 0) <= 0

    def backward_done(self):
        self.uses = None  # the backward has returned: whatever exists now is final

    def disarm(self):
        items, self.items, self.armed = self.items, {}, None
        self.uses = None
        return items

    def offer(self, param, slab_ptr, nparts, stride, numel, reduce_now, keep):
        """Called from a backward: returns the gradient to hand to autograd (None = deferred)."""
        if self.uses is not None and param is not None and id(param) in self.uses:
            self.uses[id(param)] -= 1
        if self.armed is None or param is None or id(param) not in self.armed:
            return reduce_now()
        first = self.items.pop(id(param), None)
        if first is not None:  # second use of this parameter in the step: both reduce now, autograd accumulates
            g1 = first["reduce_now"]()
            param.grad = g1 if param.grad is None else param.grad + g1
            self.armed.pop(id(param))  # no more deferral for it in this step
            return reduce_now()
        self.items[id(param)] = dict(param=param, src=slab_ptr, nparts=int(nparts), stride=int(stride), numel=int(numel),
                                     reduce_now=reduce_now, keep=keep)
        return None


deferred = DeferredGrads()


class StepFusion(object):
    """Cross-op state of ONE fused training step, armed by the trainers around ``model(x)`` + criterion.

    The reference's step ends with  y = sigmoid(...);  loss = BCELoss()(y, t);  loss.backward();  optimizer.step()
    (models/ranking/deepfm.py:43, trainers/ctr_trainer.py:88-99): five tiny launches here (head, BCE forward, BCE
    backward, Adam bias corrections, loader position) that are pure launch latency at B = 4096.  While a trainer has
    armed ``target`` (the labels of the batch), the output head also emits the per-block BCE terms, ``bce_mean`` on that
    very output finishes the mean inside ``rh_step_scalars`` -- the one single-block launch that also does the Adam
    bias corrections of the coming optimizer step and advances the registered device counters -- and the head's
    backward forms the BCE gradient inline.  Nothing changes numerically (same per-row arithmetic); any other use of
    the head's output (another criterion, no trainer) takes the unfused kernels."""

    def __init__(self):
        self.clear()

    def clear(self):
        self.target = None     # (B,) float32 labels the head may score against
        self.optimizer = None  # optim.TableAdam: its prepare() rides in rh_step_scalars
        self.counters = []     # [(int64 device tensor of one element, increment, modulus)]
        self.head = None       # dict(y_ptr, partial, t) of the head launch that computed BCE terms
        self.g_loss = None     # (1,) gradient of the loss, handed from the fused BCE backward to the head backward
        self.defer_scalars = False  # armed by a trainer whose backward follows the forward at once (train_step)
        self.pending = None    # scalar work a fused-BCE forward left to the chain's head backward (see _FusedBceFn)


fusion = StepFusion()


def _flush_pending_scalars():
    """A fused-BCE forward left its scalar work to a head backward that never ran (an abandoned step): run it now, so that
    the loader's position and the loss buffer are what they would have been."""
    pend, fusion.pending = fusion.pending, None
    if pend is not None:
        _lib.call("rh_step_scalars", _p(pend["partial"]), pend["partial"].numel(), pend["B"], _p(pend["loss"]), _p(None),
                  _p(None), _p(None), 0, *pend["flat"], _stream())


def fusion_begin(target=None, optimizer=None, counters=(), defer_scalars=False):
    _flush_pending_scalars()
    fusion.clear()
    fusion.defer_scalars = bool(defer_scalars)
    if target is not None and target.is_cuda and target.dtype == torch.float32 and target.dim() == 1 and \
            target.is_contiguous():
        fusion.target = target
    fusion.optimizer = optimizer
    fusion.counters = list(counters)


def fusion_end():
    """Disarm; returns the device counters nobody advanced (the caller launches rh_batch_advance for them)."""
    left = fusion.counters
    g, pend = fusion.g_loss, fusion.pending
    fusion.clear()
    fusion.g_loss = g  # the backward of this step has not run yet
    fusion.pending = pend  # ... and may carry the step's scalar work (_FusedBceFn, defer_scalars)
    return left


def advance_counters(counters):
    for t, inc, mod in counters:
        _lib.call("rh_batch_advance", _p(t), int(inc), int(mod), _stream())


class _HeadFn(torch.autograd.Function):
    """y = sigmoid(h w^T + b + e0 + e1): the MLP's Linear(K, 1), the wide / FM terms and the sigmoid in one launch each way."""

    @staticmethod
    def forward(ctx, h, weight, bias, e0, e1):
        require_hip(h, weight)
        if h.stride(1) != 1:
            h = h.contiguous()
        B, K = h.shape
        c0, c1 = _col(e0, B), _col(e1, B)
        y = torch.empty((B,), dtype=torch.float32, device=h.device)
        t = fusion.target
        ctx.fused_t = None
        if t is not None and t.numel() == B and t.device == h.device and B > 0 and any(ctx.needs_input_grad):
            partial = torch.empty((_lib.call("rh_head_loss_nblocks", B),), dtype=torch.float32, device=h.device)
            _lib.call("rh_head_loss_fwd", _p(h), h.stride(0), _p(weight), _p(bias), _p(c0), _p(c1), B, K, _p(y), _p(t),
                      _p(partial), _stream())
            fusion.head = dict(y_ptr=y.data_ptr(), partial=partial, t=t)
            ctx.fused_t = t
        else:
            _lib.call("rh_head_fwd", _p(h), h.stride(0), _p(weight), _p(bias), _p(c0), _p(c1), B, K, _p(y), _stream())
        ctx.save_for_backward(h, weight, y)
        ctx.shapes = (None if e0 is None else e0.shape, None if e1 is None else e1.shape, bias is not None)
        ctx.params = (weight, bias)
        # h straight out of a BatchNorm1d + ReLU + Dropout layer (the MLP's last hidden layer): the backward below then
        # also forms that layer's BatchNorm-backward column sums (one statistics launch less per step)
        node = h.grad_fn
        ctx.bn_node = node if (node is not None and type(node).__name__ == "_BnReluDropoutFnBackward" and
                               FUSE_HEAD_BN) else None
        return y

    @staticmethod
    def backward(ctx, g_y):
        h, weight, y = ctx.saved_tensors
        s0, s1, has_bias = ctx.shapes
        B, K = h.shape
        dev = h.device
        g_h = torch.empty((B, K), dtype=torch.float32, device=dev)
        g_z = torch.empty((B,), dtype=torch.float32, device=dev)
        nblk = _lib.call("rh_head_nblocks", B)
        partial = torch.empty((nblk, K + 1), dtype=torch.float32, device=dev)
        wp, bp = ctx.params
        armed = deferred.armed
        defer = armed is not None and id(wp) in armed and (not has_bias or id(bp) in armed)
        g_w = None if defer else torch.empty_like(weight)
        g_b = torch.empty((1,), dtype=torch.float32, device=dev) if (has_bias and not defer) else None
        g_loss = fusion.g_loss
        t = gl = None
        if ctx.fused_t is not None and g_loss is not None and g_loss[1] == y.data_ptr():
            # y went into the fused BCE: that gradient is formed per row inside this launch.  The loss's backward
            # returned a persistent all-zero placeholder (which it keeps a reference to, so autograd cannot sum into it
            # in place): if g_y still IS that tensor the loss was the only consumer of y, otherwise g_y = 0 + the other
            # consumers' gradient and rides along
            fusion.g_loss = None
            t, gl = ctx.fused_t, g_loss[0]
            zero = g_loss[2]
            if g_y.data_ptr() == zero.data_ptr() and g_y._version == g_loss[3]:
                g_y = None
            else:
                g_y = g_y.contiguous()
        else:
            g_y = g_y.contiguous()
        bn = ctx.bn_node
        if bn is not None and K % 4 == 0 and K <= 256 and nblk <= 128 and h.stride(0) == K:
            z, gamma, beta, stat, saved_ctr = bn.saved_tensors
            bn_partial = torch.empty((nblk, 2, K), dtype=torch.float32, device=dev)
            _lib.call("rh_head_bwd_bn", _p(h), h.stride(0), _p(weight), _p(y), _p(g_y), _p(t), _p(gl), B, K, _p(g_h),
                      _p(g_z), _p(g_w), _p(g_b), _p(partial), 0 if defer else 1, _p(z), _p(stat), _p(gamma), _p(beta),
                      float(bn.p_drop), _p(_dropout_rng(dev)), _p(saved_ctr), 1 if bn.relu else 0, _p(bn_partial), _stream())
            # valid only if the layer's upstream gradient IS this g_h, unmodified: the tensor itself + its version
            bn.rh_pre = (g_h, g_h._version, bn_partial, nblk)
        else:
            _lib.call("rh_head_bwd_ex", _p(h), h.stride(0), _p(weight), _p(y), _p(g_y), _p(t), _p(gl), B, K, _p(g_h),
                      _p(g_z), _p(g_w), _p(g_b), _p(partial), 0 if defer else 1, _stream())
        if defer:
            g_w = deferred.offer(wp, partial.data_ptr(), nblk, K + 1, K, lambda: partial[:, :K].sum(0).view_as(weight),
                                 partial)
            if has_bias:
                g_b = deferred.offer(bp, partial.data_ptr() + 4 * K, nblk, K + 1, 1, lambda: partial[:, K].sum().view(1),
                                     partial)
        return (g_h, g_w, g_b, None if s0 is None else g_z.view(s0), None if s1 is None else g_z.view(s1))


# Fused MLP chain (round 4): [Linear -> BatchNorm1d -> ReLU -> Dropout] x L -> Linear(., 1) (+ wide / FM terms)
# -> sigmoid of torch_rechub/basic/layers.py:276-292 + models/ranking/deepfm.py:39-43 as ONE autograd node over L + 1 forward
# launches and 3 L + 1 backward launches.  A/B switch for benchmarks: RECHUB_AB=chain=0 (tests flip the attribute).
FUSE_MLP_CHAIN = _lib.ab("chain")
# The chain's L weight gradients as ONE grouped launch behind its last input-gradient GEMM (A/B: RECHUB_AB=chainwgroup=0).
# Bit-equal to the per-layer launches (same body, same split plan).  Measured on the headline step, same box, 200 steps, two
# rounds: 0.2422 / 0.2424 -> 0.2411 / 0.2411 ms (profiles/r05_ab_chain_wgroup_same_box.txt): one launch less on a chain that is
# co-critical with the deferred sweep, so most of the 5 us it saves on the chain is not a shorter step.
CHAIN_WGRAD_GROUP = _lib.ab("chainwgroup")
_WGRAD_GROUP_MAX_B = 32768  # kLongRows of csrc/linear.hip: the grouped launch takes batch-sized reductions only
chain_gate = None      # the gate words of the optimizer whose step-ahead graph is being captured (optim.TableAdam), or None
chain_gate_used = []   # ... and a mark per rh_linear_fwd_gate launch captured for it
# Round 6: while optim.TableAdam captures a step-ahead graph, the chain's grouped weight gradients are not launched where the
# backward produces them but handed to the optimizer, whose end-of-step launch carries them as its first workgroups
# (rh_adam_lazy_step_ahead_wgrad): nothing on the step's critical chain reads their slabs.  wgrad_rider(problems, B) -> True
# when the optimizer took them (it launches them itself if its step then ends differently: TableAdam._flush_rider).
wgrad_rider = None
wgrad_rider_flush = None  # the rider's owner launches what it holds NOW (whoever reads weight-gradient slabs calls flush_wgrad_rider)


def flush_wgrad_rider():
    """Weight gradients handed to the optimizer's end-of-step launch that has not run yet are launched on their own, now: called
    by everything that reads their slabs (the packing launches, an immediate reduction) -- a step whose packing launch comes
    BEFORE optimizer.step() (no dense-parameter Adam riding in it) must not read slabs that are still to be written."""
    f = wgrad_rider_flush
    if f is not None:
        f()


def _reset_capture_state():
    """chain_gate / chain_gate_used / wgrad_rider belong to ONE capture (optim.TableAdam arms them at the head of a step-ahead
    capture and disarms them at its last launch).  A capture that is abandoned in between must not leave them armed for
    whatever is captured next -- another trainer's MLP chain would bake rh_linear_fwd_gate with THIS optimizer's gate words
    into its graph, or hand its weight gradients to a launch that never comes (round-5 advisor finding)."""
    global chain_gate, wgrad_rider, wgrad_rider_flush
    chain_gate = None
    wgrad_rider = None
    wgrad_rider_flush = None
    del chain_gate_used[:]


from . import graphs as _graphs  # noqa: E402  (graphs imports torch only)

_graphs.capture_end_hooks.append(_reset_capture_state)
_CHAIN_MAX_B = 4096  # rh_head_bwd_bn hands over rh_head_nblocks(B) <= 128 partial rows up to here


class _MlpChainFn(torch.autograd.Function):
    """y (B,) = sigmoid(head(hidden_L(... hidden_1(x))) + e0 + e1), hidden_l = Dropout(ReLU(BatchNorm1d(Linear_l(.)))) in
    training mode.  No hidden layer's BatchNorm / ReLU / Dropout runs as a pass of its own: the statistics are a by-product
    of the GEMM that produced the pre-activations (csrc/gemm.hip, STATS), the normalisation + activation + mask are applied
    where the NEXT layer (rh_linear_bnact_fwd) or the head (rh_head_bnact_fwd) reads them; in the backward the BatchNorm sums
    come from the head's backward (rh_head_bwd_bn, h = NULL) / the input-gradient GEMM's epilogue (rh_linear_dgrad_bnbwd).
    Saved for the backward: x, the pre-activations h_l, the activations a_l of all but the last hidden layer (written by the
    consuming GEMM for the weight gradient), per-layer (mean, rstd) and dropout counters, y.
    params = [W_l, b_l, gamma_l, beta_l] * L + [head_w, head_b]."""

    @staticmethod
    def forward(ctx, x, e0, e1, cfg, *params):
        bns, ps = cfg["bns"], cfg["p"]
        L = len(bns)
        require_hip(x, *[t for t in params if t is not None])
        B = x.shape[0]
        dev = x.device
        rng = _dropout_rng(dev)
        hs, stats, rows, ctrs, stat, acts = [], [], [], [], [], []
        inp = x
        for l in range(L):
            W, b, gamma, beta = params[4 * l:4 * l + 4]
            N, K = W.shape
            h = torch.empty((B, N), dtype=torch.float32, device=dev)
            r = _lib.call("rh_gemm_stats_rows", B, N) if l == 0 else _lib.call("rh_gemm_chain_stats_rows", B)
            st = torch.empty((-(-B // r), 2, N), dtype=torch.float32, device=dev)
            ctr = torch.empty(1, dtype=torch.int64, device=dev)
            if l == 0 and chain_gate is not None and torch.cuda.is_current_stream_capturing():
                # the first own GEMM of a step-ahead graph counts the chain start that releases the optimizer's deferred sweep
                # (optim.TableAdam sets ops.chain_gate while it captures such a step; csrc/optim.hip::stream_gate_kernel)
                _lib.call("rh_linear_fwd_gate", _p(inp), inp.stride(0), _p(W), K, _p(b), B, N, K, _p(h), N, _p(st), _p(rng),
                          _p(ctr), _p(bns[0].num_batches_tracked), _p(chain_gate), _stream())
                chain_gate_used.append(1)
            elif l == 0:
                _lib.call("rh_linear_fwd", _p(inp), inp.stride(0), _p(W), K, _p(b), B, N, K, _p(h), N, _p(st), _p(rng),
                          _p(ctr), _p(bns[0].num_batches_tracked), _stream())
            else:
                pb, pg, pbt = bns[l - 1], params[4 * (l - 1) + 2], params[4 * (l - 1) + 3]
                so = torch.empty((4, K), dtype=torch.float32, device=dev)
                act = torch.empty((B, K), dtype=torch.float32, device=dev)
                _lib.call("rh_linear_bnact_fwd", _p(hs[-1]), K, B, K, _p(stats[-1]), rows[-1], _p(pg), _p(pbt),
                          _p(pb.running_mean), _p(pb.running_var), float(pb.momentum), float(pb.eps), float(ps[l - 1]),
                          _p(rng), _p(ctrs[-1]), _p(so), _p(act), _p(W), K, _p(b), N, _p(h), N, _p(st), _p(rng), _p(ctr),
                          _p(bns[l].num_batches_tracked), _stream())
                stat.append(so)
                acts.append(act)
            hs.append(h)
            stats.append(st)
            rows.append(r)
            ctrs.append(ctr)
        hw, hb = params[4 * L], params[4 * L + 1]
        K = hs[-1].shape[1]
        lb, lg, lbt = bns[-1], params[4 * (L - 1) + 2], params[4 * (L - 1) + 3]
        so = torch.empty((4, K), dtype=torch.float32, device=dev)
        c0, c1 = _col(e0, B), _col(e1, B)
        y = torch.empty((B,), dtype=torch.float32, device=dev)
        t = fusion.target
        partial = None
        ctx.fused_t = None
        if t is not None and t.numel() == B and t.device == dev and any(ctx.needs_input_grad):
            partial = torch.empty((_lib.call("rh_head_loss_nblocks", B),), dtype=torch.float32, device=dev)
            ctx.fused_t = t
        _lib.call("rh_head_bnact_fwd", _p(hs[-1]), K, _p(stats[-1]), rows[-1], _p(lg), _p(lbt), _p(lb.running_mean),
                  _p(lb.running_var), float(lb.momentum), float(lb.eps), float(ps[-1]), _p(rng), _p(ctrs[-1]), _p(so), _p(hw),
                  _p(hb), _p(c0), _p(c1), B, K, _p(y), _p(ctx.fused_t), _p(partial), _stream())
        if ctx.fused_t is not None:
            fusion.head = dict(y_ptr=y.data_ptr(), partial=partial, t=t)
        stat.append(so)
        ctx.L, ctx.ps = L, [float(v) for v in ps]
        ctx.params = params  # the parameter objects themselves (ops.deferred keys gradients by identity)
        ctx.shapes = (None if e0 is None else e0.shape, None if e1 is None else e1.shape)
        ctx.save_for_backward(x, y, *hs, *acts, *stat, *ctrs)
        return y

    @staticmethod
    def backward(ctx, g_y):
        L, ps, params = ctx.L, ctx.ps, ctx.params
        saved = ctx.saved_tensors
        x, y = saved[0], saved[1]
        hs = saved[2:2 + L]
        acts = saved[2 + L:2 + L + (L - 1)]
        stat = saved[2 + L + (L - 1):2 + L + (L - 1) + L]
        ctrs = saved[2 + L + (L - 1) + L:]
        B = x.shape[0]
        dev = x.device
        rng = _dropout_rng(dev)
        hw, hb = params[4 * L], params[4 * L + 1]
        K = hs[-1].shape[1]
        # ---- head: d loss / d y -> g of the last hidden layer's OUTPUT + that layer's BatchNorm-backward sums
        g_a = torch.empty((B, K), dtype=torch.float32, device=dev)
        g_z = torch.empty((B,), dtype=torch.float32, device=dev)
        nblk = _lib.call("rh_head_nblocks", B)
        partial = torch.empty((nblk, K + 1), dtype=torch.float32, device=dev)
        armed = deferred.armed
        has_bias = hb is not None
        defer = armed is not None and id(hw) in armed and (not has_bias or id(hb) in armed)
        g_w = None if defer else torch.empty_like(hw)
        g_b = torch.empty((1,), dtype=torch.float32, device=dev) if (has_bias and not defer) else None
        g_loss = fusion.g_loss
        t = gl = None
        if ctx.fused_t is not None and g_loss is not None and g_loss[1] == y.data_ptr():
            fusion.g_loss = None  # (see _HeadFn.backward: the fused BCE's gradient is formed per row inside the launch)
            t, gl = ctx.fused_t, g_loss[0]
            zero = g_loss[2]
            if g_y.data_ptr() == zero.data_ptr() and g_y._version == g_loss[3]:
                g_y = None
            else:
                g_y = g_y.contiguous()
        else:
            g_y = g_y.contiguous()
        bn_partial = torch.empty((nblk, 2, K), dtype=torch.float32, device=dev)
        pend = fusion.pending
        if pend is not None and pend["y_ptr"] == y.data_ptr():
            fusion.pending = None
            opt = pend["opt"]
            hyper = step = ring = None
            ring_size = 0
            if opt is not None and opt.can_fuse_prepare():
                hyper, step, ring, ring_size = opt.fuse_prepare()
            _lib.call("rh_head_bwd_bn_scalars", _p(None), K, _p(hw), _p(y), _p(g_y), _p(t), _p(gl), B, K, _p(g_a), _p(g_z),
                      _p(g_w), _p(g_b), _p(partial), 0 if defer else 1, _p(hs[-1]), _p(stat[-1]), _p(params[4 * (L - 1) + 2]),
                      _p(params[4 * (L - 1) + 3]), ps[-1], _p(rng), _p(ctrs[-1]), 1, _p(bn_partial), _p(pend["partial"]),
                      pend["partial"].numel(), _p(pend["loss"]), _p(hyper), _p(step), _p(ring), ring_size, *pend["flat"],
                      _stream())
        else:
            _lib.call("rh_head_bwd_bn", _p(None), K, _p(hw), _p(y), _p(g_y), _p(t), _p(gl), B, K, _p(g_a), _p(g_z), _p(g_w),
                      _p(g_b), _p(partial), 0 if defer else 1, _p(hs[-1]), _p(stat[-1]), _p(params[4 * (L - 1) + 2]),
                      _p(params[4 * (L - 1) + 3]), ps[-1], _p(rng), _p(ctrs[-1]), 1, _p(bn_partial), _stream())
        if defer:
            g_w = deferred.offer(hw, partial.data_ptr(), nblk, K + 1, K, lambda: partial[:, :K].sum(0).view_as(hw), partial)
            if has_bias:
                g_b = deferred.offer(hb, partial.data_ptr() + 4 * K, nblk, K + 1, 1, lambda: partial[:, K].sum().view(1),
                                     partial)
        grads = [None] * (4 * L + 2)
        grads[4 * L], grads[4 * L + 1] = g_w, g_b
        nchunks = nblk
        g_x = None
        # The L weight gradients depend on nothing later in this backward and nothing later depends on them (their slabs are
        # summed by the step's packing launch): with ops.deferred armed for every Linear of the chain they run as ONE
        # rh_linear_wgrad_partial_group launch behind the last input-gradient GEMM instead of one launch per layer between
        # the GEMMs -- same kernel body, same split plan per problem, hence the same slabs bit for bit.
        group = (CHAIN_WGRAD_GROUP and 2 <= L <= 8 and armed is not None and B < _WGRAD_GROUP_MAX_B and
                 all(id(params[4 * l]) in armed and (params[4 * l + 1] is None or id(params[4 * l + 1]) in armed)
                     for l in range(L)))
        problems = []
        for l in range(L - 1, -1, -1):
            W, b, gamma, beta = params[4 * l:4 * l + 4]
            N, Kin = W.shape
            # BatchNorm + ReLU + Dropout backward of layer l from the sums its consumer formed: finalize + apply
            g_h = torch.empty((B, N), dtype=torch.float32, device=dev)
            dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
            _lib.call("rh_bn_relu_dropout_bwd_pre", _p(hs[l]), _p(g_a), B, N, _p(gamma), _p(beta), ps[l], _p(rng),
                      _p(ctrs[l]), _p(bn_partial), nchunks, _p(stat[l]), _p(g_h), _p(dgamma), _p(dbeta), 1, _stream())
            grads[4 * l + 2], grads[4 * l + 3] = dgamma, dbeta
            inp = acts[l - 1] if l > 0 else x
            if l > 0:
                r = _lib.call("rh_gemm_stats_rows", B, Kin)
                nchunks = -(-B // r)
                bn_partial = torch.empty((nchunks, 2, Kin), dtype=torch.float32, device=dev)
                g_a = torch.empty((B, Kin), dtype=torch.float32, device=dev)
                _lib.call("rh_linear_dgrad_bnbwd", _p(g_h), N, _p(W), Kin, B, N, Kin, _p(g_a), Kin, _p(hs[l - 1]), Kin,
                          _p(stat[l - 1]), _p(params[4 * (l - 1) + 2]), _p(params[4 * (l - 1) + 3]), ps[l - 1], _p(rng),
                          _p(ctrs[l - 1]), _p(bn_partial), _stream())
            elif ctx.needs_input_grad[0]:
                g_x = torch.empty((B, Kin), dtype=torch.float32, device=dev)
                _lib.call("rh_linear_dgrad", _p(g_h), N, _p(W), Kin, B, N, Kin, _p(g_x), Kin, _stream())
            if group:
                problems.append((l, g_h, inp, W, b))
            else:
                grads[4 * l], grads[4 * l + 1] = linear_wgrad(g_h, inp, want_bias=b is not None, weight=W, bias=b)
        if group:
            problems.reverse()  # layer 0 first: the widest problem's workgroups are dispatched first
            recs = []
            for l, g_h, inp, W, b in problems:
                N, Kin = W.shape
                part = torch.empty(_lib.call("rh_linear_wgrad_workspace", B, N, Kin), dtype=torch.float32, device=dev)
                recs.append((g_h, g_h.stride(0), inp, inp.stride(0), N, Kin, part))
            rider = wgrad_rider
            if not (rider is not None and rider(recs, B)):
                linear_wgrad_partial_group(recs, B)
            for (l, g_h, inp, W, b), rec in zip(problems, recs):
                grads[4 * l], grads[4 * l + 1] = _offer_wgrad_slabs(W, b, rec[6], B)
        s0, s1 = ctx.shapes
        return (g_x, None if s0 is None else g_z.view(s0), None if s1 is None else g_z.view(s1), None) + tuple(grads)


def mlp_chain_ok(x, blocks, head, extras):
    """blocks = [(nn.Linear, nn.BatchNorm1d, p_drop)] of consecutive Linear -> BatchNorm1d -> ReLU -> Dropout layers in
    training mode, head = the closing nn.Linear(., 1): can _MlpChainFn run them?"""
    if not (FUSE_MLP_CHAIN and blocks and torch.is_grad_enabled() and x.is_cuda and x.dim() == 2 and
            x.dtype == torch.float32 and x.stride(1) == 1 and 2 <= x.shape[0] <= _CHAIN_MAX_B and len(extras) <= 2 and
            type(head) is torch.nn.Linear and head.out_features == 1 and head.weight.dtype == torch.float32 and
            all(e.numel() == x.shape[0] for e in extras)):
        return False
    width = x.shape[1]
    for lin, bn, p in blocks:
        if not (type(lin) is torch.nn.Linear and type(bn) is torch.nn.BatchNorm1d and bn.training and bn.affine and
                bn.track_running_stats and bn.momentum is not None and lin.in_features == width and
                lin.weight.is_contiguous() and lin.weight.dtype == torch.float32 and lin.out_features % 4 == 0 and
                lin.out_features <= _GEMM_MAX_NK and width <= _GEMM_MAX_NK and 0.0 <= p < 1.0 and linear_ok(x, lin.weight)):
            return False
        width = lin.out_features
    return width <= 256 and head.in_features == width and _lib.call("rh_head_nblocks", x.shape[0]) <= 128


def mlp_chain_sigmoid(x, blocks, head, *extras):
    """sigmoid((head(hidden(x)) + sum(extras)).squeeze(1)) through _MlpChainFn; see mlp_chain_ok."""
    e = list(extras) + [None, None]
    cfg = {"bns": [bn for _, bn, _ in blocks], "p": [p for _, _, p in blocks]}
    params = []
    for lin, bn, _ in blocks:
        params += [lin.weight, lin.bias, bn.weight, bn.bias]
    return _MlpChainFn.apply(x, e[0], e[1], cfg, *params, head.weight, head.bias)


def head_ok(h, lin, extras):
    K = lin.in_features
    return (h.is_cuda and h.dim() == 2 and h.dtype == torch.float32 and lin.out_features == 1 and 1 <= K <= 1024 and
            len(extras) <= 2 and h.shape[0] > 0 and all(e.numel() == h.shape[0] for e in extras))


def head_sigmoid(h, weight, bias, *extras):
    """sigmoid((h @ weight.T + bias + sum(extras)).squeeze(1)) -> (B,)"""
    e = list(extras) + [None, None]
    return _HeadFn.apply(h, weight, bias, e[0], e[1])


class _BceFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, y, t):
        require_hip(y, t)
        y, t = y.contiguous(), t.contiguous()
        loss = torch.empty((1,), dtype=torch.float32, device=y.device)
        _lib.call("rh_bce_fwd", _p(y), _p(t), y.numel(), _p(loss), _stream())
        ctx.save_for_backward(y, t)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        y, t = ctx.saved_tensors
        g = g.contiguous().view(1)
        g_y = torch.empty_like(y)
        _lib.call("rh_bce_bwd", _p(y), _p(t), _p(g), y.numel(), _p(g_y), _stream())
        return g_y, None


def bce_ok(criterion, y, t):
    return (type(criterion) is torch.nn.BCELoss and criterion.reduction == "mean" and criterion.weight is None and
            y.is_cuda and y.dtype == torch.float32 and t.dtype == torch.float32 and y.dim() == 1 and
            y.shape == t.shape and 0 < y.numel() <= (1 << 22) and not t.requires_grad)


_zero_cache = {}


def _zero_placeholder(shape, dev):
    """A persistent, never-written all-zero float32 tensor per (shape, device) (no memset launch per step)."""
    key = (tuple(shape), str(dev))
    z = _zero_cache.get(key)
    if z is None:
        z = torch.zeros(shape, dtype=torch.float32, device=dev)
        if torch.cuda.is_current_stream_capturing():
            return z  # lives in the graph's pool, re-zeroed by the captured fill on every replay: not cached
        _zero_cache[key] = z
    return z


class _FusedBceFn(torch.autograd.Function):
    """Mean BCE of a head output whose per-block terms were computed by rh_head_loss_fwd: the forward is the step's
    ONE scalar launch (rh_step_scalars: mean of the terms + Adam bias corrections + device counters), the backward only
    hands the loss gradient to the head's backward (which forms dL/dy inline, rh_head_loss_bwd)."""

    @staticmethod
    def forward(ctx, y, t, partial):
        loss = torch.empty((1,), dtype=torch.float32, device=y.device)
        opt = fusion.optimizer
        cs = fusion.counters[:2]
        fusion.counters = fusion.counters[2:]
        flat = []
        for c, inc, mod in cs:
            flat += [_p(c), int(inc), int(mod)]
        while len(flat) < 6:
            flat += [_NULL, 0, 0]
        if fusion.defer_scalars and y.grad_fn is not None and y.grad_fn.__class__.__name__ == "_MlpChainFnBackward":
            # the chain's head backward carries this scalar work as one extra workgroup of its own launch (the loss value
            # is then written during the backward; the trainer that armed defer_scalars reads it only afterwards)
            fusion.pending = dict(partial=partial, B=y.numel(), loss=loss, flat=flat, y_ptr=y.data_ptr(), opt=opt,
                                  keep=[c for c, _, _ in cs])
        else:
            hyper = step = ring = None
            ring_size = 0
            if opt is not None and opt.can_fuse_prepare():
                hyper, step, ring, ring_size = opt.fuse_prepare()
            _lib.call("rh_step_scalars", _p(partial), partial.numel(), y.numel(), _p(loss), _p(hyper), _p(step), _p(ring),
                      ring_size, *flat, _stream())
        ctx.y_ptr = y.data_ptr()
        ctx.shape = y.shape
        ctx.dev = y.device
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        # dL/dy itself is formed inside the head's backward (from y, t and g); what goes back through autograd is a
        # persistent all-zero tensor, so a second consumer of y adds its gradient to 0 (and not to uninitialised memory)
        zero = _zero_placeholder(ctx.shape, ctx.dev)
        fusion.g_loss = (g.contiguous().view(1), ctx.y_ptr, zero, zero._version)
        return zero, None, None


def bce_mean(y, t):
    """torch.nn.BCELoss()(y, t) (mean reduction, log clamped at -100) in one launch each way; when ``y`` is the output
    of a head that already scored against ``t`` (StepFusion), the mean is finished by the step's scalar launch."""
    head = fusion.head
    if head is not None and head["y_ptr"] == y.data_ptr() and head["t"].data_ptr() == t.data_ptr() and \
            y.grad_fn is not None and y.grad_fn.__class__.__name__ in ("_HeadFnBackward", "_MlpChainFnBackward"):
        fusion.head = None
        return _FusedBceFn.apply(y, t, head["partial"])
    return _BceFn.apply(y, t)


# --------------------------------------------------------------------------------------------
class _DiceFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, alpha, eps):
        require_hip(x, alpha)
        if x.dim() != 2 or x.dtype != torch.float32:
            raise ValueError("Dice expects a float32 (N, num_neurons) tensor")
        x = x.contiguous()
        N, C = x.shape
        out = torch.empty_like(x)
        _lib.call("rh_dice_fwd", _p(x), _p(alpha), float(eps), N, C, _p(None), _p(None), _p(out), _stream())
        ctx.eps = float(eps)
        ctx.save_for_backward(x, alpha)
        return out

    @staticmethod
    def backward(ctx, g):
        x, alpha = ctx.saved_tensors
        N, C = x.shape
        g = g.contiguous()
        gx = torch.empty_like(x)
        partial = torch.empty(_lib.call("rh_dice_nblocks", N), dtype=torch.float32, device=x.device)
        _lib.call("rh_dice_bwd", _p(x), _p(g), _p(alpha), ctx.eps, N, C, _p(gx), _p(partial), _stream())
        return gx, partial.sum().reshape(1), None


def dice(x, alpha, eps):
    return _DiceFn.apply(x, alpha, eps)


class _PReluFn(torch.autograd.Function):
    """nn.PReLU() with one slope: one pass each way (csrc/din.hip); the slope gradient leaves as per-block partials
    (summed by the step's packing launch when the trainer armed ops.deferred for the parameter)."""

    @staticmethod
    def forward(ctx, x, weight):
        require_hip(x, weight)
        x = x.contiguous()
        out = torch.empty_like(x)
        _lib.call("rh_prelu_fwd", _p(x), _p(weight), x.numel(), _p(out), _stream())
        ctx.save_for_backward(x, weight)
        ctx.param = weight
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = torch.empty_like(x)
        nb = _lib.call("rh_prelu_nblocks", x.numel())
        partial = torch.empty((nb,), dtype=torch.float32, device=x.device)
        _lib.call("rh_prelu_bwd", _p(x), _p(g), _p(weight), x.numel(), _p(gx), _p(partial), _stream())
        return gx, deferred.offer(ctx.param, partial.data_ptr(), nb, 1, 1, lambda: partial.sum().reshape(1), partial)


class _L2NormFn(torch.autograd.Function):
    """F.normalize(x, p=2, dim=1, eps): one launch each way (csrc/match.hip)."""

    @staticmethod
    def forward(ctx, x, eps):
        require_hip(x)
        if x.stride(1) != 1 or x.stride(0) % 4:
            x = x.contiguous()
        B, d = x.shape
        y = torch.empty((B, d), dtype=torch.float32, device=x.device)
        nrm = torch.empty((B,), dtype=torch.float32, device=x.device)
        _lib.call("rh_l2norm_fwd", _p(x), x.stride(0), B, d, float(eps), _p(y), _p(nrm), _stream())
        ctx.eps = float(eps)
        ctx.save_for_backward(y, nrm)
        return y

    @staticmethod
    def backward(ctx, g):
        y, nrm = ctx.saved_tensors
        B, d = y.shape
        if g.stride(1) != 1 or g.stride(0) % 4:
            g = g.contiguous()
        gx = torch.empty_like(y)
        _lib.call("rh_l2norm_bwd", _p(y), _p(nrm), _p(g), g.stride(0), B, d, ctx.eps, _p(gx), _stream())
        return gx, None


def l2_normalize_ok(x):
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0 and x.shape[1] % 4 == 0 and \
        4 <= x.shape[1] <= 4096


def l2_normalize(x, eps=1e-12):
    """torch.nn.functional.normalize(x, p=2, dim=1, eps=eps) for a float32 (B, d) HIP tensor, d % 4 == 0."""
    return _L2NormFn.apply(x, eps)


class _CrossEntropyFn(torch.autograd.Function):
    """torch.nn.CrossEntropyLoss() (mean) over (B, C) logits, int64 targets (None = class 0 everywhere)."""

    @staticmethod
    def forward(ctx, logits, target):
        require_hip(logits)
        logits = logits.contiguous()
        B, C = logits.shape
        dev = logits.device
        lse = torch.empty((B,), dtype=torch.float32, device=dev)
        nb = _lib.call("rh_ce_nblocks", B)
        partial = torch.empty((nb,), dtype=torch.float32, device=dev)
        _lib.call("rh_ce_fwd", _p(logits), _p(target), B, C, _p(lse), _p(partial), _p(err_flag(dev)), _stream())
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        _lib.call("rh_colsum", _p(partial), nb, 1, _p(loss), _NULL, 0, _NULL, _stream())
        ctx.save_for_backward(logits, lse) if target is None else ctx.save_for_backward(logits, lse, target)
        ctx.has_target = target is not None
        return (loss / B).view(())

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        logits, lse = saved[0], saved[1]
        target = saved[2] if ctx.has_target else None
        B, C = logits.shape
        gx = torch.empty_like(logits)
        _lib.call("rh_ce_bwd", _p(logits), _p(target), _p(lse), _p(g.contiguous().view(1)), B, C, _p(gx), _stream())
        return gx, None


def cross_entropy_ok(criterion, logits, target):
    return (type(criterion) is torch.nn.CrossEntropyLoss and criterion.reduction == "mean" and criterion.weight is None and
            criterion.label_smoothing == 0.0 and criterion.ignore_index == -100 and logits.is_cuda and
            logits.dtype == torch.float32 and logits.dim() == 2 and 1 <= logits.shape[1] <= 1024 and logits.shape[0] > 0 and
            (target is None or (target.dtype == torch.int64 and target.shape == (logits.shape[0],) and target.is_cuda)))


def cross_entropy_mean(logits, target=None):
    return _CrossEntropyFn.apply(logits, None if target is None else target.contiguous())


def prelu_ok(mod, x):
    return (type(mod) is torch.nn.PReLU and mod.weight.numel() == 1 and x.is_cuda and x.dtype == torch.float32 and
            x.numel() >= 1)


def prelu(x, weight):
    return _PReluFn.apply(x, weight)


class _BnDiceFn(torch.autograd.Function):
    """Dice(BatchNorm1d(h)) with the normalisation folded into the Dice passes (csrc/din.hip, csrc/mlp.hip): the
    normalised tensor is never written.  Training mode; running statistics and num_batches_tracked updated in place."""

    @staticmethod
    def forward(ctx, h, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, bn_eps, alpha, eps,
                chunk_stats=None, chunk_rows=0):
        require_hip(h, gamma, beta, alpha)
        h = h.contiguous()
        N, C = h.shape
        dev = h.device
        stat = torch.empty((6, C), dtype=torch.float32, device=dev)
        if chunk_stats is not None:  # the producer of h already emitted per-chunk (sum, M2): no pass over h
            _lib.call("rh_bn_stats_from_partial", _p(chunk_stats), int(chunk_rows), N, C, _p(gamma), _p(beta),
                      _p(running_mean), _p(running_var), _p(num_batches_tracked), float(momentum), float(bn_eps), _p(stat),
                      _stream())
        else:
            partial = torch.empty((_lib.call("rh_bn_act_nchunks", N), 2, C), dtype=torch.float32, device=dev)
            _lib.call("rh_bn_stats_fwd", _p(h), N, C, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                      _p(num_batches_tracked), float(momentum), float(bn_eps), 1, _p(partial), _p(stat), _stream())
        out = torch.empty_like(h)
        _lib.call("rh_dice_fwd", _p(h), _p(alpha), float(eps), N, C, _p(stat[4]), _p(stat[5]), _p(out), _stream())
        ctx.eps = float(eps)
        ctx.save_for_backward(h, gamma, alpha, stat)
        return out

    @staticmethod
    def backward(ctx, g):
        h, gamma, alpha, stat = ctx.saved_tensors
        N, C = h.shape
        dev = h.device
        g = g.contiguous()
        nb = _lib.call("rh_bn_dice_stats_blocks", N)
        col_partial = torch.empty((nb, 2, C), dtype=torch.float32, device=dev)
        alpha_partial = torch.empty((nb,), dtype=torch.float32, device=dev)
        _lib.call("rh_bn_dice_bwd_stats", _p(h), _p(g), _p(alpha), ctx.eps, N, C, _p(stat), _p(gamma), _p(col_partial),
                  _p(alpha_partial), _stream())
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        d_alpha = torch.empty((1,), dtype=torch.float32, device=dev)
        # (the finalize launch also sums Dice's alpha partials: no framework reduction launch behind the apply pass)
        _lib.call("rh_bn_finalize_bwd_tail", _p(col_partial), nb, C, _p(stat), _p(dgamma), _p(dbeta), _p(None), _p(None),
                  _p(alpha_partial), 1, _p(d_alpha), _stream())
        dh = torch.empty_like(h)
        _lib.call("rh_bn_dice_bwd_apply", _p(h), _p(g), _p(alpha), ctx.eps, N, C, _p(stat), _p(gamma), _p(dh), _stream())
        return dh, dgamma, dbeta, None, None, None, None, None, d_alpha, None, None, None


def bn_dice_ok(h, bn, dice_mod):
    return (h.is_cuda and h.dim() == 2 and h.dtype == torch.float32 and h.shape[1] <= 512 and h.shape[0] > 1 and
            bn.affine and bn.track_running_stats and bn.momentum is not None)


def bn_dice(h, bn, alpha, eps, chunk_stats=None, chunk_rows=0):
    """Dice(bn(h)) for a Linear -> BatchNorm1d -> Dice block: folded into two passes over h (training) or one (eval).
    ``chunk_stats`` / ``chunk_rows``: per-chunk (sum, M2) of h from its producer (din_att_l1), replacing the statistics pass."""
    if bn.training:
        return _BnDiceFn.apply(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.momentum,
                               bn.eps, alpha, eps, chunk_stats, chunk_rows)
    require_hip(h)
    h = h.contiguous()
    N, C = h.shape
    stat = torch.empty((6, C), dtype=torch.float32, device=h.device)
    _lib.call("rh_bn_stats_fwd", _p(h), N, C, _p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var), _p(None),
              0.0, float(bn.eps), 0, _p(None), _p(stat), _stream())
    out = torch.empty_like(h)
    _lib.call("rh_dice_fwd", _p(h), _p(alpha), float(eps), N, C, _p(stat[4]), _p(stat[5]), _p(out), _stream())
    return out


class _BnDiceHeadFn(torch.autograd.Function):
    """Linear(C, 1)(Dice(BatchNorm1d(h))) -- the tail of the ActivationUnit's MLP (layers.py:281-288) -- without the Dice
    output: forward one pass over h writing (N, 1); backward two passes over h with the rank-1 gradient g[r] * w[c] formed in
    registers, the head's weight / bias gradients as per-block partials of the statistics pass (csrc/din.hip, HEAD)."""

    @staticmethod
    def forward(ctx, h, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, bn_eps, alpha, eps, head_w,
                head_b, chunk_stats=None, chunk_rows=0):
        require_hip(h, gamma, beta, alpha, head_w)
        h = h.contiguous()
        N, C = h.shape
        dev = h.device
        stat = torch.empty((6, C), dtype=torch.float32, device=dev)
        if chunk_stats is not None:
            _lib.call("rh_bn_stats_from_partial", _p(chunk_stats), int(chunk_rows), N, C, _p(gamma), _p(beta),
                      _p(running_mean), _p(running_var), _p(num_batches_tracked), float(momentum), float(bn_eps), _p(stat),
                      _stream())
        else:
            partial = torch.empty((_lib.call("rh_bn_act_nchunks", N), 2, C), dtype=torch.float32, device=dev)
            _lib.call("rh_bn_stats_fwd", _p(h), N, C, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                      _p(num_batches_tracked), float(momentum), float(bn_eps), 1, _p(partial), _p(stat), _stream())
        head_w = head_w.contiguous()
        out = torch.empty((N, 1), dtype=torch.float32, device=dev)
        _lib.call("rh_bn_dice_head_fwd", _p(h), _p(alpha), float(eps), N, C, _p(stat[4]), _p(stat[5]), _p(head_w),
                  _p(head_b), _p(out), _stream())
        ctx.eps = float(eps)
        ctx.has_bias = head_b is not None
        ctx.save_for_backward(h, gamma, alpha, stat, head_w)
        return out

    @staticmethod
    def backward(ctx, g):
        h, gamma, alpha, stat, head_w = ctx.saved_tensors
        N, C = h.shape
        dev = h.device
        g = g.contiguous()
        nb = _lib.call("rh_bn_dice_stats_blocks", N)
        col_partial = torch.empty((nb, 2, C), dtype=torch.float32, device=dev)
        scalar_partial = torch.empty((2, nb), dtype=torch.float32, device=dev)
        head_partial = torch.empty((nb, C), dtype=torch.float32, device=dev)
        _lib.call("rh_bn_dice_head_bwd_stats", _p(h), _p(g), _p(alpha), ctx.eps, N, C, _p(stat), _p(gamma), _p(head_w),
                  _p(col_partial), _p(scalar_partial), _p(head_partial), _stream())
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        scalars = torch.empty((2,), dtype=torch.float32, device=dev)
        d_head_w = torch.empty((1, C), dtype=torch.float32, device=dev)
        # (the finalize launch also sums the head's weight-gradient partials and the two scalar rows -- Dice's alpha, the head's
        # bias: three framework reduction launches + a fill per attention unit less on the backward's critical path)
        _lib.call("rh_bn_finalize_bwd_tail", _p(col_partial), nb, C, _p(stat), _p(dgamma), _p(dbeta), _p(head_partial),
                  _p(d_head_w), _p(scalar_partial), 2, _p(scalars), _stream())
        dh = torch.empty_like(h)
        _lib.call("rh_bn_dice_head_bwd_apply", _p(h), _p(g), _p(alpha), ctx.eps, N, C, _p(stat), _p(gamma), _p(head_w),
                  _p(dh), _stream())
        return (dh, dgamma, dbeta, None, None, None, None, None, scalars[0:1], None, d_head_w,
                scalars[1:2] if ctx.has_bias else None, None, None)


def bn_dice_head_ok(h_cols, bn, dice_mod, lin):
    return (type(lin) is torch.nn.Linear and lin.out_features == 1 and lin.in_features == h_cols and h_cols <= 512 and
            lin.weight.dtype == torch.float32 and FUSE_DICE_HEAD)


def bn_dice_head(h, bn, alpha, eps, lin, chunk_stats=None, chunk_rows=0):
    """lin(Dice(bn(h))) for Linear -> BatchNorm1d -> Dice -> Linear(C, 1); (N, 1)."""
    if bn.training:
        return _BnDiceHeadFn.apply(h, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                   bn.momentum, bn.eps, alpha, eps, lin.weight, lin.bias, chunk_stats, chunk_rows)
    require_hip(h)
    h = h.contiguous()
    N, C = h.shape
    stat = torch.empty((6, C), dtype=torch.float32, device=h.device)
    _lib.call("rh_bn_stats_fwd", _p(h), N, C, _p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var), _p(None),
              0.0, float(bn.eps), 0, _p(None), _p(stat), _stream())
    out = torch.empty((N, 1), dtype=torch.float32, device=h.device)
    _lib.call("rh_bn_dice_head_fwd", _p(h), _p(alpha), float(eps), N, C, _p(stat[4]), _p(stat[5]),
              _p(lin.weight.detach().contiguous()), _p(None if lin.bias is None else lin.bias.detach()), _p(out), _stream())
    return out


def _hist_layout(history):
    """(B, L, D) view whose rows are contiguous over (L, D); returns (tensor, batch stride in floats)."""
    if history.stride(2) != 1 or history.stride(1) != history.shape[2]:
        history = history.contiguous()
    return history, history.stride(0)


class _AttInputFn(torch.autograd.Function):
    """cat[t, h, t-h, t*h] over (B*L, 4D) without the expand / sub / mul / cat temporaries."""

    @staticmethod
    def forward(ctx, history, target):
        require_hip(history, target)
        history, hs = _hist_layout(history)
        if target.stride(1) != 1:
            target = target.contiguous()
        B, L, D = history.shape
        out = torch.empty((B * L, 4 * D), dtype=torch.float32, device=history.device)
        _lib.call("rh_din_att_input_fwd", _p(history), hs, _p(target), target.stride(0), B, L, D, _p(out), _stream())
        ctx.save_for_backward(history, target)
        return out

    @staticmethod
    def backward(ctx, g):
        history, target = ctx.saved_tensors
        B, L, D = history.shape
        g = g.contiguous()
        g_hist = torch.empty((B, L, D), dtype=torch.float32, device=g.device)
        g_tgt = torch.empty((B, D), dtype=torch.float32, device=g.device)
        _lib.call("rh_din_att_input_bwd", _p(history), history.stride(0), _p(target), target.stride(0), _p(g), B, L, D,
                  _p(g_hist), _p(g_tgt), _stream())
        return g_hist, g_tgt


class _AttPoolFn(torch.autograd.Function):
    """(att_weight.unsqueeze(-1) * history).sum(dim=1)."""

    @staticmethod
    def forward(ctx, weight, history):
        require_hip(weight, history)
        history, hs = _hist_layout(history)
        weight = weight.contiguous()
        B, L, D = history.shape
        out = torch.empty((B, D), dtype=torch.float32, device=history.device)
        _lib.call("rh_din_pool_fwd", _p(history), hs, _p(weight), B, L, D, _p(out), _stream())
        ctx.save_for_backward(weight, history)
        return out

    @staticmethod
    def backward(ctx, g):
        weight, history = ctx.saved_tensors
        B, L, D = history.shape
        g = g.contiguous()
        g_hist = torch.empty((B, L, D), dtype=torch.float32, device=g.device)
        g_w = torch.empty((B, L), dtype=torch.float32, device=g.device)
        _lib.call("rh_din_pool_bwd", _p(history), history.stride(0), _p(weight), _p(g), B, L, D, _p(g_hist), _p(g_w),
                  _stream())
        return g_w, g_hist


class _AttL1Fn(torch.autograd.Function):
    """z (B*L, N) = [t, h, t-h, t*h] W^T + b with the operand built in registers (csrc/dinmlp.hip) and, when asked, the
    per-chunk BatchNorm statistics of z as an epilogue.  Backward: the operand is rebuilt once (rh_din_att_input_fwd) for
    the split-batch weight gradient; input gradient = library GEMM + rh_din_att_input_bwd."""

    @staticmethod
    def forward(ctx, history, target, weight, bias, want_stats):
        require_hip(history, target, weight)
        history, hs = _hist_layout(history)
        if target.stride(1) != 1:
            target = target.contiguous()
        weight_c = weight.contiguous()
        B, L, D = history.shape
        N = weight.shape[0]
        dev = history.device
        z = torch.empty((B * L, N), dtype=torch.float32, device=dev)
        partial = None
        if want_stats:
            rows = _lib.call("rh_din_att_l1_chunk_rows", B * L)
            partial = torch.empty((-(-(B * L) // rows), 2, N), dtype=torch.float32, device=dev)
        _lib.call("rh_din_att_l1_fwd", _p(history), hs, _p(target), target.stride(0), _p(weight_c),
                  _p(None if bias is None else bias.contiguous()), B, L, D, N, _p(z), _p(partial), _stream())
        ctx.save_for_backward(history, target, weight_c)
        ctx.params = (weight, bias)
        if partial is None:
            return z, None
        ctx.mark_non_differentiable(partial)
        return z, partial

    @staticmethod
    def backward(ctx, g, _unused):
        history, target, weight = ctx.saved_tensors
        B, L, D = history.shape
        wp, bp = ctx.params
        g = g.contiguous()
        dev = g.device
        dW = db = None
        if ctx.needs_input_grad[2] or (bp is not None and ctx.needs_input_grad[3]):
            att = torch.empty((B * L, 4 * D), dtype=torch.float32, device=dev)
            _lib.call("rh_din_att_input_fwd", _p(history), history.stride(0), _p(target), target.stride(0), B, L, D, _p(att),
                      _stream())
            dW, db = linear_wgrad(g, att, want_bias=bp is not None, weight=wp, bias=bp)
        g_hist = g_tgt = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            # (B*L, N) x (N, 4D): the tile kernel of csrc/gemm.hip runs this tall, narrow-output product at 184 us against
            # the library's 260 us at configs[3] (tools/gemm_tall_probe.py)
            g_att = torch.empty((B * L, 4 * D), dtype=torch.float32, device=dev)
            _lib.call("rh_linear_dgrad", _p(g), g.stride(0), _p(weight), 4 * D, B * L, weight.shape[0], 4 * D, _p(g_att),
                      4 * D, _stream())
            g_hist = torch.empty((B, L, D), dtype=torch.float32, device=dev)
            g_tgt = torch.empty((B, D), dtype=torch.float32, device=dev)
            _lib.call("rh_din_att_input_bwd", _p(history), history.stride(0), _p(target), target.stride(0), _p(g_att), B, L,
                      D, _p(g_hist), _p(g_tgt), _stream())
        return g_hist, g_tgt, dW, db, None


def din_att_l1_ok(history, target, lin):
    return (history.is_cuda and history.dtype == torch.float32 and history.dim() == 3 and history.shape[0] >= 1 and
            type(lin) is torch.nn.Linear and lin.in_features == 4 * history.shape[2] and
            _lib.call("rh_din_att_l1_supported", history.shape[2], lin.out_features) == 1)


def din_att_l1(history, target, weight, bias, want_stats):
    """(z, chunk_stats | None): the ActivationUnit's first Linear on the never-materialised [t, h, t-h, t*h] operand."""
    return _AttL1Fn.apply(history, target, weight, bias, want_stats)


def din_att_input(history, target):
    return _AttInputFn.apply(history, target)


def din_att_pool(weight, history):
    return _AttPoolFn.apply(weight, history)


def din_dim_ok(d):
    q = d // 4
    return d % 4 == 0 and 1 <= q <= 32 and (q & (q - 1)) == 0


# --------------------------------------------------------------------------------------------
class _CrossV2Epilogue(torch.autograd.Function):
    """out = x0 * y + b + x  (y = W_l x): Hadamard + bias + residual of one CrossNetV2 layer in one pass."""

    @staticmethod
    def forward(ctx, x0, y, b, x):
        require_hip(x0, y, b, x)
        x0, y, b, x = x0.contiguous(), y.contiguous(), b.contiguous(), x.contiguous()
        B, d = x.shape
        out = torch.empty_like(x)
        _lib.call("rh_cross_v2_epilogue_fwd", _p(x0), _p(y), _p(b), _p(x), B, d, _p(out), _stream())
        ctx.save_for_backward(x0, y)
        return out

    @staticmethod
    def backward(ctx, g):
        x0, y = ctx.saved_tensors
        B, d = x0.shape
        g = g.contiguous()
        g_x0 = torch.empty_like(x0)
        g_y = torch.empty_like(y)
        _lib.call("rh_cross_v2_epilogue_bwd", _p(x0), _p(y), _p(g), B, d, _p(g_x0), _p(g_y), _stream())
        return g_x0, g_y, g.sum(0), g


def cross_v2_epilogue(x0, y, b, x):
    return _CrossV2Epilogue.apply(x0, y, b, x)


class _CrossV2LayerFn(torch.autograd.Function):
    """out = x0 * (x W^T) + b + x: ONE CrossNetV2 layer (torch_rechub/basic/layers.py:440-444) as one launch each for the
    forward (rh_cross_v2_fwd: tile GEMM whose epilogue is the Hadamard + bias + residual) and the input gradient
    (rh_cross_v2_dgrad: g_y W + g), plus rh_cross_v2_epilogue_bwd (g_x0 = g * y, g_y = g * x0) and the split-batch weight
    gradient.  Round 5; before: library GEMM + epilogue pass forward, epilogue pass + two library products + autograd's residual
    add backward."""

    @staticmethod
    def forward(ctx, x0, x, weight, bias):
        require_hip(x0, x, weight, bias)
        x0, x = x0.contiguous(), x.contiguous()
        B, d = x.shape
        y = torch.empty_like(x)
        out = torch.empty_like(x)
        _lib.call("rh_cross_v2_fwd", _p(x0), _p(x), _p(weight), _p(bias), B, d, _p(y), _p(out), _stream())
        ctx.save_for_backward(x0, x, y, weight)
        ctx.params = (weight, bias)
        return out

    @staticmethod
    def backward(ctx, g):
        x0, x, y, weight = ctx.saved_tensors
        B, d = x.shape
        g = g.contiguous()
        g_x0 = torch.empty_like(x0)
        g_y = torch.empty_like(y)
        _lib.call("rh_cross_v2_epilogue_bwd", _p(x0), _p(y), _p(g), B, d, _p(g_x0), _p(g_y), _stream())
        g_x = None
        if ctx.needs_input_grad[1]:
            g_x = torch.empty_like(x)
            _lib.call("rh_cross_v2_dgrad", _p(g_y), _p(weight), _p(g), B, d, _p(g_x), _stream())
        g_w = None
        if ctx.needs_input_grad[2]:
            g_w, _ = linear_wgrad(g_y, x, want_bias=False, weight=ctx.params[0])
        g_b = g.sum(0) if ctx.needs_input_grad[3] else None
        return (g_x0 if ctx.needs_input_grad[0] else None), g_x, g_w, g_b


CROSS_V2_MAX_B, CROSS_V2_MAX_D = 16384, 1024  # limits of rh_cross_v2_fwd / rh_cross_v2_dgrad (enforced there: csrc/gemm.hip)


def cross_v2_shape_ok(x, weight):
    """THE predicate of the fused CrossNetV2 layer (ops.cross_v2_layer_ok and torch.ops.rechub_hip.cross_net_v2 share it):
    contiguous f32 x (B, d) and W (d, d) within the entry points' limits."""
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32 and
            1 <= x.shape[0] <= CROSS_V2_MAX_B and 1 <= x.shape[1] <= CROSS_V2_MAX_D and
            tuple(weight.shape[-2:]) == (x.shape[1], x.shape[1]))


def cross_v2_layer_ok(x, lin):
    """Can one CrossNetV2 layer run on the tile GEMM (f32, square weight without bias up to the GEMM's widths)?"""
    return (type(lin) is torch.nn.Linear and lin.bias is None and x.dim() == 2 and cross_v2_shape_ok(x, lin.weight) and
            x.shape[1] <= _GEMM_MAX_NK and lin.weight.is_contiguous() and linear_ok(x, lin.weight))


def cross_v2_layer(x0, x, weight, bias):
    return _CrossV2LayerFn.apply(x0, x, weight, bias)


class _CrossMixEpilogue(torch.autograd.Function):
    """out = sum_e gate_e * x0 * (uv_e + bias) + xl: bias + Hadamard + gated expert mix + residual in one pass."""

    @staticmethod
    def forward(ctx, x0, xl, uv, gate, bias):
        require_hip(x0, xl, uv, gate, bias)
        ctx.bias_param, ctx.bias_shape = bias, bias.shape  # the (d, 1) / (d,) parameter itself (ops.deferred keys by identity)
        x0, xl, uv, gate = x0.contiguous(), xl.contiguous(), uv.contiguous(), gate.contiguous()
        bias = bias.reshape(-1).contiguous()
        E, B, d = uv.shape
        out = torch.empty_like(x0)
        _lib.call("rh_cross_mix_epilogue_fwd", _p(x0), _p(xl), _p(uv), _p(gate), _p(bias), B, d, E, _p(out), _stream())
        ctx.save_for_backward(x0, uv, gate, bias)
        return out

    @staticmethod
    def backward(ctx, g):
        x0, uv, gate, bias = ctx.saved_tensors
        E, B, d = uv.shape
        g = g.contiguous()
        g_x0 = torch.empty_like(x0)
        g_uv = torch.empty_like(uv)
        g_gate = torch.empty_like(gate)
        nblk = _lib.call("rh_cross_mix_nblocks", B)
        partial = torch.empty((nblk, d), dtype=torch.float32, device=g.device)
        _lib.call("rh_cross_mix_epilogue_bwd_b", _p(x0), _p(uv), _p(gate), _p(bias), _p(g), B, d, E, _p(g_x0), _p(g_uv),
                  _p(g_gate), _p(partial), _stream())
        # d/d bias = sum_b g x0 sum_e gate_e: per-block partial rows from the same pass (was 2 multiplies + 2 reductions)
        shape = ctx.bias_shape
        g_bias = deferred.offer(ctx.bias_param, partial.data_ptr(), nblk, d, d, lambda: partial.sum(0).view(shape), partial)
        return g_x0, g, g_uv, g_gate, g_bias


def cross_mix_epilogue(x0, xl, uv, gate, bias):
    return _CrossMixEpilogue.apply(x0, xl, uv, gate, bias)


# --------------------------------------------------------------------------------------------
def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _int_array(values):
    import ctypes
    return (ctypes.c_int * len(values))(*values)


def cross_moe_ok(x, L, E, d, r):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 1 and x.stride(1) == 1 and
            _lib.call("rh_cross_moe_supported", L, E, d, r) == 1)


class _CrossMoeFn(torch.autograd.Function):
    """CrossNetMix.forward (reference basic/layers.py:470-506) for ALL its layers: per layer two library GEMMs
    (B, d) x (d, KP) and (B, KP) x (KP, d), the mid pass between them and one addcmul (csrc/moe.hip); the backward is two
    GEMMs + two split-batch weight-gradient launches + two fused passes per layer, and ONE launch that turns the
    weight-gradient slabs of all layers into the gradients of u_list / v_list / c_list / bias / gating.

    inputs: x, then L x U, L x V, L x C, L x bias, E x gating weight."""

    @staticmethod
    def forward(ctx, x, L, E, *params):
        import ctypes
        require_hip(x, *params)
        U, V, C, bias, Wg = (params[0:L], params[L:2 * L], params[2 * L:3 * L], params[3 * L:4 * L], params[4 * L:4 * L + E])
        B, d = x.shape
        r = U[0].shape[2]
        KP = _lib.call("rh_cross_moe_kp", E, r)
        dev = x.device
        cont = [t.contiguous() for t in params]
        Uc, Vc, Cc, bc, Wgc = cont[0:L], cont[L:2 * L], cont[2 * L:3 * L], cont[3 * L:4 * L], cont[4 * L:4 * L + E]
        VgT = torch.empty((L, KP, d), dtype=torch.float32, device=dev)
        UTb = torch.empty((L, d, KP), dtype=torch.float32, device=dev)
        _lib.call("rh_cross_moe_pack", ctypes.cast(_ptr_array(Uc), ctypes.c_void_p), ctypes.cast(_ptr_array(Vc), ctypes.c_void_p),
                  ctypes.cast(_ptr_array(bc), ctypes.c_void_p), ctypes.cast(_ptr_array(Wgc), ctypes.c_void_p), L, E, d, r,
                  _p(VgT), _p(UTb), _stream())
        xl, saved = x, []
        for l in range(L):
            PG = torch.mm(xl, VgT[l].t())
            v1 = torch.empty((B, E * r), dtype=torch.float32, device=dev)
            v2 = torch.empty_like(v1)
            gate = torch.empty((B, E), dtype=torch.float32, device=dev)
            wp = torch.empty((B, KP), dtype=torch.float32, device=dev)
            _lib.call("rh_cross_moe_mid_fwd", _p(PG), _p(Cc[l]), B, E, r, _p(v1), _p(v2), _p(gate), _p(wp), _stream())
            Y = torch.mm(wp, UTb[l].t())
            nxt = torch.addcmul(xl, x, Y)
            saved += [xl, v1, v2, gate, wp, Y]
            xl = nxt
        ctx.dims = (L, E, B, d, r, KP)
        ctx.shapes = [t.shape for t in params]
        ctx.save_for_backward(x, VgT, UTb, *Cc, *saved)
        return xl

    @staticmethod
    def backward(ctx, G):
        import ctypes
        L, E, B, d, r, KP = ctx.dims
        x, VgT, UTb = ctx.saved_tensors[:3]
        Cc = ctx.saved_tensors[3:3 + L]
        saved = ctx.saved_tensors[3 + L:]
        dev = x.device
        if G.stride(1) != 1:
            G = G.contiguous()
        acc = torch.empty((B, d), dtype=torch.float32, device=dev)  # running gradient of x0
        nb = _lib.call("rh_cross_moe_mid_blocks", B, E, r)
        s1 = _lib.call("rh_linear_wgrad_splits", B, KP, d)
        s2 = _lib.call("rh_linear_wgrad_splits", B, d, KP)
        ws1 = _lib.call("rh_linear_wgrad_workspace", B, KP, d)
        ws2 = _lib.call("rh_linear_wgrad_workspace", B, d, KP)
        slabV, slabU, gC = [], [], []
        # The two weight gradients of a layer (g_UTb = g_Y^T wp, g_VgT = g_PG^T x_l) feed nothing but the final unpack: they
        # are collected and run as grouped launches of <= 8 problems after the loop (round 4: 2 L launches of ~16.5 us -> 1)
        group = WGRAD_GROUP and B < 32768
        problems = []
        own_G = False
        for l in reversed(range(L)):
            xl, v1, v2, gate, wp, Y = saved[6 * l:6 * l + 6]
            g_Y = torch.empty((B, d), dtype=torch.float32, device=dev)
            # (l == 0: acc also takes G, the closing product below accumulates into it -- that IS the gradient of x)
            _lib.call("rh_cross_moe_res_bwd", _p(G), G.stride(0), _p(x), x.stride(0), _p(Y), B, d,
                      (1 if l == L - 1 else 0) | (2 if l == 0 else 0), _p(g_Y), _p(acc), _stream())
            pu = torch.empty((ws2,), dtype=torch.float32, device=dev)  # g_UTb (d, KP) = g_Y^T wp
            if group:
                problems.append((g_Y, d, wp, KP, d, KP, pu))
            else:
                _lib.call("rh_linear_wgrad_partial", _p(g_Y), d, _p(wp), KP, B, d, KP, _p(pu), _stream())
            g_wp = torch.mm(g_Y, UTb[l])
            g_PG = torch.empty((B, KP), dtype=torch.float32, device=dev)
            pc = torch.empty((nb, E * r * r), dtype=torch.float32, device=dev)
            _lib.call("rh_cross_moe_mid_bwd", _p(g_wp), _p(v1), _p(v2), _p(gate), _p(Cc[l]), B, E, r, _p(g_PG), _p(pc),
                      _stream())
            pv = torch.empty((ws1,), dtype=torch.float32, device=dev)  # g_VgT (KP, d) = g_PG^T x_l
            if group:
                problems.append((g_PG, KP, xl, xl.stride(0), KP, d, pv))
            else:
                _lib.call("rh_linear_wgrad_partial", _p(g_PG), KP, _p(xl), xl.stride(0), B, KP, d, _p(pv), _stream())
            # gradient of x_l: residual + through the first product.  Only the first of the backward copies (the incoming
            # gradient is not ours to overwrite); the others accumulate in place, the last one into acc
            if l == 0:
                g_x = acc.addmm_(g_PG, VgT[l])  # x is both x_0 (Hadamard factor of every layer) and the first x_l
            elif own_G:
                G.addmm_(g_PG, VgT[l])
            else:
                G, own_G = torch.addmm(G, g_PG, VgT[l]), True
            slabV.append(pv), slabU.append(pu), gC.append(pc)
        for i in range(0, len(problems), 8):
            linear_wgrad_partial_group(problems[i:i + 8], B)
        slabV.reverse(), slabU.reverse(), gC.reverse()
        g_U = [torch.empty(ctx.shapes[l], dtype=torch.float32, device=dev) for l in range(L)]
        g_V = [torch.empty(ctx.shapes[L + l], dtype=torch.float32, device=dev) for l in range(L)]
        g_C = [torch.empty(ctx.shapes[2 * L + l], dtype=torch.float32, device=dev) for l in range(L)]
        g_b = [torch.empty(ctx.shapes[3 * L + l], dtype=torch.float32, device=dev) for l in range(L)]
        g_W = [torch.empty(ctx.shapes[4 * L + e], dtype=torch.float32, device=dev) for e in range(E)]
        cast = lambda arr: ctypes.cast(arr, ctypes.c_void_p)
        _lib.call("rh_cross_moe_unpack", cast(_ptr_array(slabV)), cast(_int_array([s1] * L)), cast(_ptr_array(slabU)),
                  cast(_int_array([s2] * L)), cast(_ptr_array(gC)), cast(_int_array([nb] * L)), L, E, d, r,
                  cast(_ptr_array(g_U)), cast(_ptr_array(g_V)), cast(_ptr_array(g_b)), cast(_ptr_array(g_C)),
                  cast(_ptr_array(g_W)), _stream())
        return (g_x, None, None) + tuple(g_U) + tuple(g_V) + tuple(g_C) + tuple(g_b) + tuple(g_W)


WGRAD_GROUP = _lib.ab("wgroup")  # False (RECHUB_AB=wgroup=0, tests): one rh_linear_wgrad_partial launch per problem


def wgrad_group_args(problems, B):
    """The nine array arguments of a grouped weight-gradient launch (n, g, ldg, x, ldx, B, N, K, partial) + what must stay
    alive until the call returns.  problems = [(g (B, N), ldg, x (B, K), ldx, N, K, partial workspace)]."""
    import ctypes
    n = len(problems)
    g = (ctypes.c_void_p * n)(*[p_[0].data_ptr() for p_ in problems])
    ldg = (ctypes.c_int64 * n)(*[int(p_[1]) for p_ in problems])
    x = (ctypes.c_void_p * n)(*[p_[2].data_ptr() for p_ in problems])
    ldx = (ctypes.c_int64 * n)(*[int(p_[3]) for p_ in problems])
    Bs = (ctypes.c_int * n)(*[int(B)] * n)
    Ns = (ctypes.c_int * n)(*[int(p_[4]) for p_ in problems])
    Ks = (ctypes.c_int * n)(*[int(p_[5]) for p_ in problems])
    part = (ctypes.c_void_p * n)(*[p_[6].data_ptr() for p_ in problems])
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
    return (n, cast(g), cast(ldg), cast(x), cast(ldx), cast(Bs), cast(Ns), cast(Ks), cast(part)), (g, ldg, x, ldx, Bs, Ns, Ks, part)


def linear_wgrad_partial_group(problems, B):
    """ONE launch for <= 8 independent weight-gradient problems (g (B, N) ld, x (B, K) ld, N, K, partial workspace)."""
    args, keep = wgrad_group_args(problems, B)
    _lib.call("rh_linear_wgrad_partial_group", *args, _stream())
    del keep


def cross_moe(x, u_list, v_list, c_list, bias_list, gating_weights):
    """The whole CrossNetMix stack (all layers) through csrc/moe.hip; check ``cross_moe_ok`` first."""
    L, E = len(u_list), len(gating_weights)
    return _CrossMoeFn.apply(x, L, E, *u_list, *v_list, *c_list, *bias_list, *gating_weights)


# --------------------------------------------------------------------------------------------
class _AugruFn(torch.autograd.Function):
    """h_all (B, T, D) = gated recurrence over xw (B, T, 3D), attn (B, T) | None, U (D, 3D), state_bias (3D) | None
    (csrc/augru.hip)."""

    @staticmethod
    def forward(ctx, xw, attn, U, ub):
        require_hip(xw, U)
        B, T, D3 = xw.shape
        D = D3 // 3
        xw, U = xw.contiguous(), U.contiguous()
        attn = None if attn is None else attn.contiguous()
        ub = None if ub is None else ub.contiguous()
        h_all = torch.empty((B, T, D), dtype=torch.float32, device=xw.device)
        _lib.call("rh_augru_fwd", _p(xw), _p(attn), _p(U), _p(ub), B, T, D, _p(h_all), _stream())
        ctx.save_for_backward(xw, attn, U, ub, h_all)
        return h_all

    @staticmethod
    def backward(ctx, g):
        xw, attn, U, ub, h_all = ctx.saved_tensors
        B, T, D3 = xw.shape
        D = D3 // 3
        g = g.contiguous()
        d_xw = torch.empty_like(xw)
        d_huh = torch.empty((B, T, D), dtype=torch.float32, device=xw.device)
        d_attn = None if attn is None else torch.empty((B, T), dtype=torch.float32, device=xw.device)
        _lib.call("rh_augru_bwd", _p(xw), _p(attn), _p(U), _p(ub), _p(h_all), _p(g), B, T, D, _p(d_xw), _p(d_huh),
                  _p(d_attn), _stream())
        d_U = d_ub = None
        if ctx.needs_input_grad[2] or (ub is not None and ctx.needs_input_grad[3]):
            # gradient of s = h_{t-1} U + state_bias: [d pre_u | d pre_r | d s_h] for every (sample, step); dU^T =
            # d_s^T h_prev sums over B*T rows -- the split-batch MFMA weight-gradient kernel's shape (a library GEMM
            # with K = B*T = 409600 took 580 us here), which also returns the column sums = d state_bias
            d_s = torch.cat([d_xw[:, :, :2 * D], d_huh], dim=2).reshape(B * T, D3)
            h_prev = torch.cat([h_all.new_zeros(B, 1, D), h_all[:, :-1]], dim=1).reshape(B * T, D)
            d_Ut, d_ub = linear_wgrad(d_s, h_prev, want_bias=ub is not None)
            d_U = d_Ut.t()
        return d_xw, d_attn, d_U, d_ub


def augru_ok(xw, D):
    return (xw.is_cuda and xw.dtype == torch.float32 and xw.dim() == 3 and xw.shape[1] >= 1 and
            D in (4, 8, 16, 32) and xw.shape[2] == 3 * D)


def augru(xw, attn, U, state_bias=None):
    """States h_1 .. h_T (B, T, D) of the attentional-update GRU (reference dien.py:30-36, 60-66) in one launch each
    way; ``xw`` holds the input halves of the three gates for every step, ``U`` = [Uu | Ur | Uh]."""
    return _AugruFn.apply(xw, attn, U, state_bias)


def gru_ok(gru, x):
    return (isinstance(gru, torch.nn.GRU) and gru.num_layers == 1 and not gru.bidirectional and gru.batch_first and
            gru.bias and gru.input_size == x.shape[-1] and x.dim() == 3 and x.is_cuda and x.dtype == torch.float32 and
            x.shape[1] >= 1 and gru.hidden_size in (4, 8, 16, 32))


def gru(gru_mod, x):
    """Outputs (B, T, H) of a one-layer ``nn.GRU(batch_first=True)`` from the zero state, through the same recurrence
    kernel: PyTorch's cell is r, z, n with h' = (1 - z) n + z h, i.e. the kernel's update gate is 1 - z = sigmoid of
    the NEGATED z pre-activation, its weight a = 1, and bias_hh rides on the state product."""
    H = gru_mod.hidden_size
    w_ih, w_hh, b_ih, b_hh = gru_mod.weight_ih_l0, gru_mod.weight_hh_l0, gru_mod.bias_ih_l0, gru_mod.bias_hh_l0

    def gates(m):  # rows (r, z, n) of a (3H, ...) parameter -> kernel order (1 - z, r, n)
        return torch.cat([-m[H:2 * H], m[:H], m[2 * H:]], dim=0)

    B, T, _ = x.shape
    xw = linear(x.reshape(B * T, -1), gates(w_ih), gates(b_ih)).view(B, T, 3 * H)
    return _AugruFn.apply(xw, None, gates(w_hh).t(), gates(b_hh))


# --------------------------------------------------------------------------------------------
_sample_rng = {}


class _InbatchLogitsFn(torch.autograd.Function):
    """(B, 1 + K) in-batch logits straight from the tower outputs (csrc/match.hip): no (B, C) score matrix."""

    @staticmethod
    def forward(ctx, u, v, neg, row0):
        require_hip(u, v, neg)
        if u.stride(1) != 1:
            u = u.contiguous()
        if v.stride(1) != 1:
            v = v.contiguous()
        neg = neg.contiguous()
        B, D = u.shape
        C, K = v.shape[0], neg.shape[1]
        logits = torch.empty((B, 1 + K), dtype=torch.float32, device=u.device)
        _lib.call("rh_inbatch_logits_fwd", _p(u), u.stride(0), _p(v), v.stride(0), _p(neg), B, C, D, K, int(row0),
                  _p(logits), _p(err_flag(u.device)), _stream())
        ctx.save_for_backward(u, v, neg)
        ctx.row0 = int(row0)
        return logits

    @staticmethod
    def backward(ctx, g):
        u, v, neg = ctx.saved_tensors
        B, D = u.shape
        C, K = v.shape[0], neg.shape[1]
        g = g.contiguous()
        g_u = torch.empty((B, D), dtype=torch.float32, device=u.device)
        g_v = torch.zeros((C, D), dtype=torch.float32, device=u.device)
        _lib.call("rh_inbatch_logits_bwd", _p(u), u.stride(0), _p(v), v.stride(0), _p(neg), _p(g), B, C, D, K, ctx.row0,
                  _p(g_u), _p(g_v), _stream())
        return g_u, g_v, None, None


def inbatch_logits_ok(u, v):
    return (u.is_cuda and v.is_cuda and u.dtype == torch.float32 and v.dtype == torch.float32 and u.dim() == 2 and
            v.dim() == 2 and u.shape[1] == v.shape[1] and 1 <= u.shape[1] <= 1024 and u.shape[0] >= 1)


def inbatch_logits(u, v, neg, row0=0):
    """logits[i, 0] = u_i . v_(row0 + i), logits[i, 1 + k] = u_i . v_neg[i, k]  (== gather_inbatch_logits(u @ v.T, neg))."""
    return _InbatchLogitsFn.apply(u, v, neg, row0)


def inbatch_sample(batch_size, k, device, seed=None, cols=None, row0=0):
    """(B, K) int64: per row K distinct in-batch negatives (never the row itself), uniformly at random; hipGraph-safe.

    ``cols`` / ``row0``: the rows are rows [row0, row0 + B) of a (cols x cols) problem (cross-rank negatives): row r
    draws from {0..cols-1} minus {row0 + r}, with the random stream of global row row0 + r."""
    key = (str(device), seed)
    st = _sample_rng.get(key)
    if st is None:
        s = torch.initial_seed() if seed is None else int(seed)
        st = torch.tensor([s & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64).to(device)
        _sample_rng[key] = st
    out = torch.empty((batch_size, k), dtype=torch.int64, device=device)
    if cols is None:
        _lib.call("rh_inbatch_sample", _p(st), batch_size, k, _p(out), _stream())
    else:
        _lib.call("rh_inbatch_sample_rows", _p(st), batch_size, int(cols), int(row0), k, _p(out), _stream())
    _lib.call("rh_batch_advance", ctypes.c_void_p(st.data_ptr() + 8), 1, 0, _stream())
    return out


def shard_narrow(idx, out):
    """``out`` (N, F) int32, contiguous <- the int64 index matrix ``idx`` (N, F) (rows contiguous, any row stride), saturating
    (``rh_shard_narrow``: what ``idx.clamp(-1, 2**31 - 1).to(torch.int32)`` computes, written where the caller says)."""
    require_hip(idx, out)
    if idx.dim() != 2 or idx.dtype != torch.int64 or idx.stride(1) != 1 or out.shape != idx.shape or \
            out.dtype != torch.int32 or not out.is_contiguous():
        raise ValueError("shard_narrow: idx (N, F) int64 with contiguous rows, out (N, F) contiguous int32")
    _lib.call("rh_shard_narrow", _p(idx), idx.stride(0), int(idx.shape[0]), int(idx.shape[1]), _p(out), _stream())
    return out


def shard_localize(idx, desc, world, rank, out=None):
    """int32 (N, F): the index matrix ``idx`` (N, F) of a global batch rewritten for this rank's table shards
    (``rh_shard_localize``; desc = device int64 [vocab | pad | sink] per field, see sharding.RowShard)."""
    require_hip(idx, desc)
    if idx.dim() != 2 or not idx.is_contiguous() or idx.dtype not in (torch.int64, torch.int32):
        raise ValueError("shard_localize: indices must be a contiguous (N, F) int64 / int32 matrix")
    N, F = int(idx.shape[0]), int(idx.shape[1])
    if desc.numel() != 3 * F or desc.dtype != torch.int64:
        raise ValueError("shard_localize: descriptor must hold 3 * F int64 entries")
    if out is None:
        out = torch.empty((N, F), dtype=torch.int32, device=idx.device)
    elif out.shape != (N, F) or out.dtype != torch.int32 or not out.is_contiguous():
        raise ValueError("shard_localize: out must be a contiguous int32 (N, F) tensor")
    _lib.call("rh_shard_localize", _p(idx), 1 if idx.dtype == torch.int64 else 0, N, F, _p(desc), int(world),
              int(rank), _p(out), _p(err_flag(idx.device)), _stream())
    return out

"""``torch.library`` registration of the stateless interaction kernels (namespace ``rechub_hip``): schema, HIP
implementation, fake (meta) implementation and autograd formula -- what SURVEY 8(b) calls the op-library form of the
boundary.  With these, ``FakeTensorMode`` / ``torch.export`` / ``torch.compile`` can trace through

    torch.ops.rechub_hip.fm(x, reduce_sum)                     FM.forward            basic/layers.py:313-319
    torch.ops.rechub_hip.cross_network(x, W, b)                CrossNetwork.forward  basic/layers.py:412-420
    torch.ops.rechub_hip.dice(x, alpha, eps)                   Dice.forward          basic/activation.py:15-25

    torch.ops.rechub_hip.embedding_fm_lr(tables, idx, dense, lr_w, lr_b)
                                                               EmbeddingLayer.forward + FM + LR  layers.py:77-127, :313-319, :185-189
    torch.ops.rechub_hip.cross_net_v2(x, W, b)                 CrossNetV2.forward    basic/layers.py:440-444
    torch.ops.rechub_hip.cross_net_mix(x, U, V, C, bias, gating)  CrossNetMix.forward  basic/layers.py:470-506
    torch.ops.rechub_hip.din_attention_input(history, target) / din_attention_pool(att_weight, history)
                                                               ActivationUnit.forward  models/ranking/din.py:77-92
    torch.ops.rechub_hip.embedding_bag_masked(weight, idx, pad_sentinel, mode)
                                                               sequence gather + InputMask + pooling  layers.py:86-99, :148-161, :204-251
    torch.ops.rechub_hip.inbatch_negative_sample(scores, k, hard, seed, call)
                                                               inbatch_negative_sampling  utils/match.py:104-161
    torch.ops.rechub_hip.adam_step_(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay)
                                                               torch.optim.Adam on one tensor    trainers/ctr_trainer.py:59-61, :99

without running a kernel (shape / dtype propagation), and ``torch.library.opcheck`` validates schema, fake impl and
autograd registration on the device (tests/test_gpu_kernels.py).  The layers of ``torch_rechub_amd.basic`` call the SAME
C entry points through ``autograd.Function`` (no dispatcher hop per call in the eager / hipGraph step).

The training fast path of the gather, its backward and the optimizer owns persistent table-gradient buffers and claim
words -- hidden state a functional schema cannot express.  Their ``torch.library`` form is therefore the FUNCTIONAL
restatement of the same kernels: ``embedding_fm_lr`` returns (flattened embeddings + dense block, fm, lr, S) from
``rh_embed_fwd``; its backward op returns the per-lookup gradient rows (``rh_embed_bwd`` in row form, no buffer behind it)
and the autograd formula turns them into one dense gradient per table with ``index_add`` -- exact, traceable, exportable
(``export_onnx`` / ``torch.compile`` users of trainers/ctr_trainer.py:189-245), and O(vocab) per step, which is why the
trainers do not use it.  ``adam_step_`` declares its mutated arguments in the schema (``Tensor(a!)``).

The implementations refuse CPU tensors like every op of this package (no fallback).
"""
import torch

from . import _lib, ops

_p, _stream = ops._p, ops._stream


# ---- FM ------------------------------------------------------------------------------------------------------------
@torch.library.custom_op("rechub_hip::fm", mutates_args=())
def fm(x: torch.Tensor, reduce_sum: bool) -> torch.Tensor:
    ops.require_hip(x)
    x = x.contiguous()
    B, F, D = x.shape
    out = torch.empty((B, 1) if reduce_sum else (B, D), dtype=torch.float32, device=x.device)
    _lib.call("rh_fm_fwd", _p(x), x.stride(0), B, F, D, 1 if reduce_sum else 0, _p(out), _stream())
    return out


@fm.register_fake
def _(x, reduce_sum):
    B, F, D = x.shape
    return x.new_empty((B, 1) if reduce_sum else (B, D))


@torch.library.custom_op("rechub_hip::fm_backward", mutates_args=())
def fm_backward(x: torch.Tensor, g: torch.Tensor, reduce_sum: bool) -> torch.Tensor:
    ops.require_hip(x, g)
    x, g = x.contiguous(), g.contiguous()
    B, F, D = x.shape
    gx = torch.empty((B, F, D), dtype=torch.float32, device=x.device)
    _lib.call("rh_fm_bwd", _p(x), x.stride(0), B, F, D, 1 if reduce_sum else 0, _p(g), _p(gx), gx.stride(0), _stream())
    return gx


@fm_backward.register_fake
def _(x, g, reduce_sum):
    return torch.empty_like(x)


def _fm_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])
    ctx.reduce_sum = inputs[1]


def _fm_bwd(ctx, g):
    return torch.ops.rechub_hip.fm_backward(ctx.saved_tensors[0], g, ctx.reduce_sum), None


fm.register_autograd(_fm_bwd, setup_context=_fm_setup)


# ---- CrossNetwork (up to rh_cross_max_layers(d) layers per call, i.e. every reference configuration) ---------------
@torch.library.custom_op("rechub_hip::cross_network", mutates_args=())
def cross_network(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    ops.require_hip(x, W, b)
    x, W, b = x.contiguous(), W.contiguous(), b.contiguous()
    B, d = x.shape
    L = W.shape[0]
    if not 0 < L <= _lib.call("rh_cross_max_layers", d):
        raise ValueError(f"rechub_hip::cross_network: {L} layers of width {d} in one call unsupported")
    out = torch.empty((B, d), dtype=torch.float32, device=x.device)
    _lib.call("rh_cross_fwd", _p(x), d, _p(x), d, _p(W), _p(b), B, d, L, _p(out), d, _stream())
    return out


@cross_network.register_fake
def _(x, W, b):
    return torch.empty_like(x)


@torch.library.custom_op("rechub_hip::cross_network_backward", mutates_args=())
def cross_network_backward(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor,
                           g: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    ops.require_hip(x, W, b, g)
    x, W, b, g = x.contiguous(), W.contiguous(), b.contiguous(), g.contiguous()
    B, d = x.shape
    L = W.shape[0]
    nblocks = _lib.call("rh_cross_bwd_nblocks", B)
    partial = torch.empty((nblocks, 2, L, d), dtype=torch.float32, device=x.device)
    gx = torch.empty((B, d), dtype=torch.float32, device=x.device)
    _lib.call("rh_cross_bwd", _p(x), d, _p(x), d, _p(W), _p(b), B, d, L, _p(g), d, _p(None), _p(gx), d, 1, _p(partial),
              _stream())
    red = partial.sum(0)
    return gx, red[0].clone(), red[1].clone()


@cross_network_backward.register_fake
def _(x, W, b, g):
    return torch.empty_like(x), torch.empty_like(W), torch.empty_like(b)


def _cross_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _cross_bwd(ctx, g):
    x, W, b = ctx.saved_tensors
    return torch.ops.rechub_hip.cross_network_backward(x, W, b, g)


cross_network.register_autograd(_cross_bwd, setup_context=_cross_setup)


# ---- Dice ----------------------------------------------------------------------------------------------------------
@torch.library.custom_op("rechub_hip::dice", mutates_args=())
def dice(x: torch.Tensor, alpha: torch.Tensor, eps: float) -> torch.Tensor:
    ops.require_hip(x, alpha)
    x = x.contiguous()
    N, C = x.shape
    out = torch.empty_like(x)
    _lib.call("rh_dice_fwd", _p(x), _p(alpha), float(eps), N, C, _p(None), _p(None), _p(out), _stream())
    return out


@dice.register_fake
def _(x, alpha, eps):
    return torch.empty_like(x)


@torch.library.custom_op("rechub_hip::dice_backward", mutates_args=())
def dice_backward(x: torch.Tensor, alpha: torch.Tensor, g: torch.Tensor, eps: float) -> tuple[torch.Tensor, torch.Tensor]:
    ops.require_hip(x, alpha, g)
    x, g = x.contiguous(), g.contiguous()
    N, C = x.shape
    gx = torch.empty_like(x)
    partial = torch.empty(_lib.call("rh_dice_nblocks", N), dtype=torch.float32, device=x.device)
    _lib.call("rh_dice_bwd", _p(x), _p(g), _p(alpha), float(eps), N, C, _p(gx), _p(partial), _stream())
    return gx, partial.sum().reshape(1)


@dice_backward.register_fake
def _(x, alpha, g, eps):
    return torch.empty_like(x), alpha.new_empty((1,))


def _dice_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])
    ctx.eps = inputs[2]


def _dice_bwd(ctx, g):
    x, alpha = ctx.saved_tensors
    gx, ga = torch.ops.rechub_hip.dice_backward(x, alpha, g, ctx.eps)
    return gx, ga, None


dice.register_autograd(_dice_bwd, setup_context=_dice_setup)


# ---- fused multi-field gather + FM + LR (functional form) -------------------------------------------------------------
def _embed_call(tables, idx, dense):
    F = len(tables)
    if idx.dim() != 2 or idx.shape[1] != F or idx.dtype not in (torch.int64, torch.int32):
        raise ValueError("rechub_hip::embedding_fm_lr: idx must be an integer (B, F) matrix, F = len(tables)")
    idx = idx.contiguous()
    cols = [idx[:, f] for f in range(F)]
    dcols = [] if dense is None else [dense[:, j] for j in range(dense.shape[1])]
    return ops.EmbedCall(list(tables), [None] * F, cols, dense=dcols, want_fm=True, want_lr=True), idx


@torch.library.custom_op("rechub_hip::embedding_fm_lr", mutates_args=())
def embedding_fm_lr(tables: list[torch.Tensor], idx: torch.Tensor, dense: torch.Tensor | None, lr_w: torch.Tensor,
                    lr_b: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """(out (B, F*D + n_dense), fm (B, 1), lr (B, 1), S (B, D)): sparse block first, dense values last (layers.py:120)."""
    ops.require_hip(idx, lr_w, lr_b, *tables)
    if dense is not None:
        dense = dense.float().contiguous()
    call, idx = _embed_call(tables, idx, dense)
    B, F, D = call.B, call.F, call.D
    dev = idx.device
    out = torch.empty((B, call.width), dtype=torch.float32, device=dev)
    fm_ = torch.empty((B, 1), dtype=torch.float32, device=dev)
    lr = torch.empty((B, 1), dtype=torch.float32, device=dev)
    s_sum = torch.empty((B, D), dtype=torch.float32, device=dev)
    lw = lr_w.contiguous()
    _lib.call("rh_embed_fwd", _p(call.fdesc(False)), _p(call.idesc()), call.idx_is_i64, B, F, D, _p(call.ddesc()),
              len(call.dense), call.dense_col, _p(out), out.stride(0), _p(lw), _p(lr_b), _p(lr), _p(fm_), _p(s_sum),
              call.field_split, _p(ops.err_flag(dev)), _stream())
    return out, fm_, lr, s_sum


@embedding_fm_lr.register_fake
def _(tables, idx, dense, lr_w, lr_b):
    B, F = idx.shape
    D = tables[0].shape[1]
    nd = 0 if dense is None else dense.shape[1]
    t = tables[0]
    return t.new_empty((B, F * D + nd)), t.new_empty((B, 1)), t.new_empty((B, 1)), t.new_empty((B, D))


@torch.library.custom_op("rechub_hip::embedding_fm_lr_backward", mutates_args=())
def embedding_fm_lr_backward(tables: list[torch.Tensor], idx: torch.Tensor, out: torch.Tensor, s_sum: torch.Tensor,
                             lr_w: torch.Tensor, g_out: torch.Tensor | None, g_fm: torch.Tensor | None,
                             g_lr: torch.Tensor | None) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(rows (B, F, D) = the gradient of every looked-up row, g_lr_w, g_lr_b): rh_embed_bwd in its row form."""
    ops.require_hip(idx, out, *tables)
    call, idx = _embed_call(tables, idx, None)
    B, F, D = call.B, call.F, call.D
    dev = idx.device
    rows = torch.empty((B, F, D), dtype=torch.float32, device=dev)
    nch = _lib.call("rh_embed_bwd_nchunks", B, call.samples_per_block)
    partial = torch.zeros((nch, F * D), dtype=torch.float32, device=dev)
    g_out = None if g_out is None else g_out.contiguous()
    g_fm = None if g_fm is None else g_fm.reshape(-1).contiguous()
    g_lr = None if g_lr is None else g_lr.reshape(-1).contiguous()
    lw = lr_w.contiguous()
    out = out.contiguous()
    _lib.call("rh_embed_bwd", _p(call.fdesc(False)), _p(call.idesc()), call.idx_is_i64, B, F, D, _p(g_out),
              0 if g_out is None else g_out.stride(0), _p(out), out.stride(0), _p(s_sum), _p(g_fm), _p(g_lr),
              _p(lw), _p(partial if g_lr is not None else None), 1.0, 1, _p(rows), call.samples_per_block,
              _p(ops.err_flag(dev)), _stream())
    g_w = partial.sum(0).view_as(lr_w)
    g_b = (g_lr.sum() if g_lr is not None else torch.zeros((), device=dev)).reshape(1)
    return rows, g_w, g_b


@embedding_fm_lr_backward.register_fake
def _(tables, idx, out, s_sum, lr_w, g_out, g_fm, g_lr):
    B, F = idx.shape
    return out.new_empty((B, F, tables[0].shape[1])), torch.empty_like(lr_w), lr_w.new_empty((1,))


def _embed_setup(ctx, inputs, output):
    tables, idx, dense, lr_w, lr_b = inputs
    out, fm_, lr, s_sum = output
    ctx.tables = list(tables)
    ctx.n_dense = 0 if dense is None else dense.shape[1]
    ctx.has_dense = dense is not None
    ctx.save_for_backward(idx, out, s_sum, lr_w)
    ctx.set_materialize_grads(False)


def _embed_bwd(ctx, g_out, g_fm, g_lr, g_s):
    idx, out, s_sum, lr_w = ctx.saved_tensors
    if g_s is not None:
        raise RuntimeError("rechub_hip::embedding_fm_lr: S (the per-sample field sum) is an auxiliary output")
    rows, g_w, g_b = torch.ops.rechub_hip.embedding_fm_lr_backward(ctx.tables, idx, out, s_sum, lr_w, g_out, g_fm, g_lr)
    by_table = {}
    for f, t in enumerate(ctx.tables):  # shared tables (shared_with): their fields' rows add up
        g = by_table.get(id(t))
        if g is None:
            g = by_table[id(t)] = torch.zeros_like(t)
        g.index_add_(0, idx[:, f].long(), rows[:, f])
    seen, g_tables = set(), []
    for t in ctx.tables:
        g_tables.append(None if id(t) in seen else by_table[id(t)])
        seen.add(id(t))
    g_dense = None
    if ctx.has_dense and g_out is not None:
        g_dense = g_out[:, out.shape[1] - ctx.n_dense:]
    return g_tables, None, g_dense, g_w, (g_b if g_lr is not None else None)


embedding_fm_lr.register_autograd(_embed_bwd, setup_context=_embed_setup)


# ---- torch.optim.Adam (coupled weight decay) on one tensor, in place -------------------------------------------------
@torch.library.custom_op("rechub_hip::adam_step_", mutates_args=("p", "g", "m", "v"))
def adam_step_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float, beta1: float,
               beta2: float, eps: float, weight_decay: float) -> None:
    """One dense Adam step of ``p`` (step = the 1-based step count AFTER the increment); ``g`` is re-zeroed, as the
    trainer's optimizer does instead of model.zero_grad() (ctr_trainer.py:97)."""
    import ctypes
    ops.require_hip(p, g, m, v)
    if not all(t.is_contiguous() and t.dtype == torch.float32 and t.numel() == p.numel() for t in (p, g, m, v)) or \
            p.numel() % 4:
        raise ValueError("rechub_hip::adam_step_: contiguous float32 tensors of one size (numel % 4 == 0)")
    dev = p.device
    hyper = torch.zeros(16, dtype=torch.float64, device=dev)
    hyper[:5] = torch.tensor([lr, beta1, beta2, eps, weight_decay], dtype=torch.float64)
    cnt = torch.full((1,), int(step) - 1, dtype=torch.int64, device=dev)
    _lib.call("rh_adam_prepare", _p(hyper), _p(cnt), _p(None), 0, _stream())
    desc = torch.tensor([p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()], dtype=torch.int64).to(dev)
    numel = (ctypes.c_int64 * 1)(p.numel())
    _lib.call("rh_adam_dense", _p(desc), 1, ctypes.cast(numel, ctypes.c_void_p), _p(hyper), 1, _stream())


@adam_step_.register_fake
def _(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay):
    return None


# =====================================================================================================================
# Round 4: the rest of SURVEY 8(b)'s op list.  The kernels behind these ops are the ones the layers of
# ``torch_rechub_amd.basic`` / the models call through ``autograd.Function``; here the SAME forward / backward bodies run
# under a stand-in for the autograd context (``_Ctx``), so that each direction is a dispatcher op of its own with a
# schema, a fake implementation and an autograd formula made of ops (AOT autograd can trace the backward too).  A
# backward op recomputes what the forward would have saved (these are the functional restatements: nothing is cached
# between two op calls).
class _Ctx(object):
    """What an ``autograd.Function`` body touches on its ``ctx``, outside autograd."""

    def __init__(self, needs_input_grad=()):
        self.needs_input_grad = tuple(needs_input_grad)
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


# ---- CrossNetV2: x_{l+1} = x0 * (W_l x_l) + b_l + x_l  (basic/layers.py:440-444) ----------------------------------------
def _cross_v2_layers(x, W, b):
    """[(x_l, y_l)] for every layer and the output; y_l = x_l W_l^T on the library GEMM, the rest ONE pass per layer."""
    ops.require_hip(x, W, b)
    if x.dim() != 2 or W.dim() != 3 or W.shape[1] != x.shape[1] or W.shape[2] != x.shape[1] or b.shape != W.shape[:2]:
        raise ValueError("rechub_hip::cross_net_v2: x (B, d), W (L, d, d), b (L, d)")
    x0 = x.contiguous()
    B, d = x0.shape
    xl, layers = x0, []
    fused = ops.cross_v2_shape_ok(x0, W)  # (one predicate with ops.cross_v2_layer_ok; the C entry points enforce the limits)
    for l in range(W.shape[0]):
        out = torch.empty_like(x0)
        if fused:  # ONE launch per layer: tile GEMM with the Hadamard + bias + residual epilogue (csrc/gemm.hip, round 5)
            y = torch.empty_like(x0)
            _lib.call("rh_cross_v2_fwd", _p(x0), _p(xl), _p(W[l].contiguous()), _p(b[l].contiguous()), B, d, _p(y), _p(out),
                      _stream())
        else:
            y = torch.mm(xl, W[l].t()).contiguous()
            _lib.call("rh_cross_v2_epilogue_fwd", _p(x0), _p(y), _p(b[l].contiguous()), _p(xl), B, d, _p(out), _stream())
        layers.append((xl, y))
        xl = out
    return x0, layers, xl


@torch.library.custom_op("rechub_hip::cross_net_v2", mutates_args=())
def cross_net_v2(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return _cross_v2_layers(x, W, b)[2]


@cross_net_v2.register_fake
def _(x, W, b):
    return torch.empty_like(x)


@torch.library.custom_op("rechub_hip::cross_net_v2_backward", mutates_args=())
def cross_net_v2_backward(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor,
                          g: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    x0, layers, _ = _cross_v2_layers(x, W, b)
    B, d = x0.shape
    g = g.contiguous()
    g_x0_total = torch.zeros_like(x0)
    g_W, g_b = torch.empty_like(W), torch.empty_like(b)
    for l in reversed(range(W.shape[0])):
        xl, y = layers[l]
        g_x0, g_y = torch.empty_like(x0), torch.empty_like(y)
        _lib.call("rh_cross_v2_epilogue_bwd", _p(x0), _p(y), _p(g), B, d, _p(g_x0), _p(g_y), _stream())
        g_x0_total += g_x0
        g_b[l] = g.sum(0)
        g_W[l] = torch.mm(g_y.t(), xl)
        if x0.dtype == torch.float32 and W.dtype == torch.float32 and 1 <= B <= 16384 and d <= 1024:
            g_next = torch.empty_like(g)  # g_y W_l + g: the input-gradient GEMM with the residual as its epilogue
            _lib.call("rh_cross_v2_dgrad", _p(g_y), _p(W[l].contiguous()), _p(g), B, d, _p(g_next), _stream())
            g = g_next
        else:
            g = (g + torch.mm(g_y, W[l])).contiguous()  # residual + through W_l
    return g + g_x0_total, g_W, g_b


@cross_net_v2_backward.register_fake
def _(x, W, b, g):
    return torch.empty_like(x), torch.empty_like(W), torch.empty_like(b)


def _cross_v2_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _cross_v2_bwd(ctx, g):
    x, W, b = ctx.saved_tensors
    return torch.ops.rechub_hip.cross_net_v2_backward(x, W, b, g)


cross_net_v2.register_autograd(_cross_v2_bwd, setup_context=_cross_v2_setup)


# ---- CrossNetMix: mixture of low-rank experts (basic/layers.py:470-506), all layers, csrc/moe.hip --------------------------
def _mix_params(U, V, C, bias, gating):
    L, E = U.shape[0], U.shape[1]
    d = U.shape[2]
    if V.shape != U.shape or C.shape[:2] != (L, E) or gating.shape[0] != E or bias.shape[0] != L or \
            bias.numel() != L * d or gating.numel() != E * d:
        raise ValueError("rechub_hip::cross_net_mix: U, V (L, E, d, r), C (L, E, r, r), bias (L, d) or (L, d, 1), "
                         "gating (E, d) or (E, 1, d)")
    return L, E, ([U[l] for l in range(L)] + [V[l] for l in range(L)] + [C[l] for l in range(L)] +
                  [bias[l].reshape(d, 1) for l in range(L)] + [gating[e].reshape(1, d) for e in range(E)])


@torch.library.custom_op("rechub_hip::cross_net_mix", mutates_args=())
def cross_net_mix(x: torch.Tensor, U: torch.Tensor, V: torch.Tensor, C: torch.Tensor, bias: torch.Tensor,
                  gating: torch.Tensor) -> torch.Tensor:
    L, E, params = _mix_params(U, V, C, bias, gating)
    if not ops.cross_moe_ok(x, L, E, U.shape[2], U.shape[3]):
        raise ValueError("rechub_hip::cross_net_mix: unsupported shape (rh_cross_moe_supported)")
    return ops._CrossMoeFn.forward(_Ctx(), x.contiguous(), L, E, *params)


@cross_net_mix.register_fake
def _(x, U, V, C, bias, gating):
    return torch.empty_like(x)


@torch.library.custom_op("rechub_hip::cross_net_mix_backward", mutates_args=())
def cross_net_mix_backward(x: torch.Tensor, U: torch.Tensor, V: torch.Tensor, C: torch.Tensor, bias: torch.Tensor,
                           gating: torch.Tensor, g: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor,
                                                                             torch.Tensor, torch.Tensor, torch.Tensor]:
    L, E, params = _mix_params(U, V, C, bias, gating)
    ctx = _Ctx()
    ops._CrossMoeFn.forward(ctx, x.contiguous(), L, E, *params)
    grads = ops._CrossMoeFn.backward(ctx, g.contiguous())
    g_x, rest = grads[0], grads[3:]
    # the gradients of bias / gating leave in the callers' shapes: (L, d) or the reference's (L, d, 1) layout
    # (basic/layers.py:468: nn.Parameter(torch.empty(input_dim, 1)) per layer), as the fake implementation below promises
    return (g_x, torch.stack(rest[0:L]), torch.stack(rest[L:2 * L]), torch.stack(rest[2 * L:3 * L]),
            torch.stack([t.reshape(-1) for t in rest[3 * L:4 * L]]).view_as(bias),
            torch.stack([t.reshape(-1) for t in rest[4 * L:4 * L + E]]).view_as(gating))


@cross_net_mix_backward.register_fake
def _(x, U, V, C, bias, gating, g):
    return (torch.empty_like(x), torch.empty_like(U), torch.empty_like(V), torch.empty_like(C), torch.empty_like(bias),
            torch.empty_like(gating))


def _mix_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _mix_bwd(ctx, g):
    return torch.ops.rechub_hip.cross_net_mix_backward(*ctx.saved_tensors, g)


cross_net_mix.register_autograd(_mix_bwd, setup_context=_mix_setup)


# ---- DIN target attention (models/ranking/din.py:77-92): the two kernels around the attention MLP ---------------------
#   din_attention_input(history (B, L, D), target (B, D)) -> (B * L, 4 D) = cat[t, h, t - h, t * h]     din.py:79-81
#   din_attention_pool(att_weight (B, L), history (B, L, D)) -> (B, D) = sum_L att_weight * history        din.py:89-92
# (the MLP between them is nn.Linear + BatchNorm1d + rechub_hip::dice; no padding mask)
@torch.library.custom_op("rechub_hip::din_attention_input", mutates_args=())
def din_attention_input(history: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return ops._AttInputFn.forward(_Ctx(), history, target)


@din_attention_input.register_fake
def _(history, target):
    B, L, D = history.shape
    return history.new_empty((B * L, 4 * D))


@torch.library.custom_op("rechub_hip::din_attention_input_backward", mutates_args=())
def din_attention_input_backward(history: torch.Tensor, target: torch.Tensor,
                                 g: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    ctx = _Ctx()
    ops._AttInputFn.forward(ctx, history, target)
    return ops._AttInputFn.backward(ctx, g)


@din_attention_input_backward.register_fake
def _(history, target, g):
    return torch.empty_like(history), torch.empty_like(target)


def _att_in_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _att_in_bwd(ctx, g):
    return torch.ops.rechub_hip.din_attention_input_backward(*ctx.saved_tensors, g)


din_attention_input.register_autograd(_att_in_bwd, setup_context=_att_in_setup)


@torch.library.custom_op("rechub_hip::din_attention_pool", mutates_args=())
def din_attention_pool(att_weight: torch.Tensor, history: torch.Tensor) -> torch.Tensor:
    return ops._AttPoolFn.forward(_Ctx(), att_weight, history)


@din_attention_pool.register_fake
def _(att_weight, history):
    B, L, D = history.shape
    return history.new_empty((B, D))


@torch.library.custom_op("rechub_hip::din_attention_pool_backward", mutates_args=())
def din_attention_pool_backward(att_weight: torch.Tensor, history: torch.Tensor,
                                g: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    ctx = _Ctx()
    ops._AttPoolFn.forward(ctx, att_weight, history)
    return ops._AttPoolFn.backward(ctx, g)


@din_attention_pool_backward.register_fake
def _(att_weight, history, g):
    return torch.empty_like(att_weight), torch.empty_like(history)


def _att_pool_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _att_pool_bwd(ctx, g):
    return torch.ops.rechub_hip.din_attention_pool_backward(*ctx.saved_tensors, g)


din_attention_pool.register_autograd(_att_pool_bwd, setup_context=_att_pool_setup)


# ---- masked embedding bag: gather + InputMask + Sum / Average / ConcatPooling (basic/layers.py:86-99,148-161,204-251) -----
_BAG_MODES = {"sum": 0, "mean": 1, "concat": 2}


@torch.library.custom_op("rechub_hip::embedding_bag_masked", mutates_args=())
def embedding_bag_masked(weight: torch.Tensor, idx: torch.Tensor, pad_sentinel: int, mode: str) -> torch.Tensor:
    """(B, D) for mode 'sum' / 'mean' (mean divides by the count of idx != pad_sentinel, + 1e-16: layers.py:229),
    (B, L, D) for 'concat' (no mask).  pad_sentinel = padding_idx, or -1 when the feature has none (layers.py:154-157)."""
    ops.require_hip(weight, idx)
    if mode not in _BAG_MODES or idx.dim() != 2 or idx.dtype not in (torch.int64, torch.int32):
        raise ValueError("rechub_hip::embedding_bag_masked: idx (B, L) integer, mode in sum | mean | concat")
    B, L = idx.shape
    V, D = weight.shape
    m = _BAG_MODES[mode]
    w = weight.contiguous()
    out = torch.empty((B, L, D) if m == 2 else (B, D), dtype=torch.float32, device=weight.device)
    _lib.call("rh_seq_pool_fwd", _p(w), V, _p(idx), 1 if idx.dtype == torch.int64 else 0, idx.stride(0), idx.stride(1), B, L, D,
              m, int(pad_sentinel), _p(out), out.stride(0), _p(ops.err_flag(weight.device)), _stream())
    return out


@embedding_bag_masked.register_fake
def _(weight, idx, pad_sentinel, mode):
    B, L = idx.shape
    D = weight.shape[1]
    return weight.new_empty((B, L, D) if mode == "concat" else (B, D))


@torch.library.custom_op("rechub_hip::embedding_bag_masked_backward", mutates_args=())
def embedding_bag_masked_backward(weight: torch.Tensor, idx: torch.Tensor, g: torch.Tensor, pad_sentinel: int,
                                  mode: str) -> torch.Tensor:
    """Dense gradient of the table (functional form: a fresh zero buffer receives the scatter; the trainers scatter into
    their persistent gradient buffers instead)."""
    ops.require_hip(weight, idx, g)
    B, L = idx.shape
    V, D = weight.shape
    g = g.contiguous()
    buf = torch.zeros((V, D), dtype=torch.float32, device=weight.device)
    _lib.call("rh_seq_pool_bwd", _p(buf), V, _p(idx), 1 if idx.dtype == torch.int64 else 0, idx.stride(0), idx.stride(1), B, L,
              D, _BAG_MODES[mode], int(pad_sentinel), int(pad_sentinel), _p(g), g.stride(0), 1.0, _p(ops.err_flag(weight.device)),
              _stream())  # (pad_sentinel >= 0 is the feature's padding_idx: nn.Embedding gives that row no gradient)
    return buf


@embedding_bag_masked_backward.register_fake
def _(weight, idx, g, pad_sentinel, mode):
    return torch.empty_like(weight)


def _bag_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])
    ctx.sentinel, ctx.mode = inputs[2], inputs[3]


def _bag_bwd(ctx, g):
    weight, idx = ctx.saved_tensors
    return torch.ops.rechub_hip.embedding_bag_masked_backward(weight, idx, g, ctx.sentinel, ctx.mode), None, None, None


embedding_bag_masked.register_autograd(_bag_bwd, setup_context=_bag_setup)


# ---- in-batch negatives (utils/match.py:104-161) ------------------------------------------------------------------------
@torch.library.custom_op("rechub_hip::inbatch_negative_sample", mutates_args=())
def inbatch_negative_sample(scores: torch.Tensor, k: int, hard: bool, seed: int, call: int) -> torch.Tensor:
    """(B, K) int64 negatives per row of the (B, B) score matrix, never the row's own column.  hard: the K largest
    off-diagonal scores (utils/match.py:124-131, deterministic); else K distinct columns uniformly at random from the
    counter-based stream (seed, call) -- the FUNCTIONAL form of the sampler's device-side state (the trainers advance
    ``call`` on the device): same (seed, call) -> same indices (the 'fast' stream of utils/match.py here; its distribution,
    not the reference's randperm indices -- DESIGN 5)."""
    ops.require_hip(scores)
    B = scores.shape[0]
    if scores.dim() != 2 or scores.shape[1] != B or not 1 <= k <= B - 1:
        raise ValueError("rechub_hip::inbatch_negative_sample: scores (B, B), 1 <= k <= B - 1")
    if hard:
        masked = scores.detach().clone()
        masked.fill_diagonal_(float("-inf"))
        return masked.topk(k, dim=1).indices
    st = torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, int(call)], dtype=torch.int64).to(scores.device)
    out = torch.empty((B, k), dtype=torch.int64, device=scores.device)
    _lib.call("rh_inbatch_sample", _p(st), B, int(k), _p(out), _stream())
    return out


@inbatch_negative_sample.register_fake
def _(scores, k, hard, seed, call):
    return scores.new_empty((scores.shape[0], k), dtype=torch.int64)

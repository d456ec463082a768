"""``torch.library`` registration of the stateless interaction kernels (namespace ``rechub_hip``): schema, HIP
implementation, fake (meta) implementation and autograd formula -- what SURVEY 8(b) calls the op-library form of the
boundary.  With these, ``FakeTensorMode`` / ``torch.export`` / ``torch.compile`` can trace through

    torch.ops.rechub_hip.fm(x, reduce_sum)                     FM.forward            basic/layers.py:313-319
    torch.ops.rechub_hip.cross_network(x, W, b)                CrossNetwork.forward  basic/layers.py:412-420
    torch.ops.rechub_hip.dice(x, alpha, eps)                   Dice.forward          basic/activation.py:15-25

without running a kernel (shape / dtype propagation), and ``torch.library.opcheck`` validates schema, fake impl and
autograd registration on the device (tests/test_gpu_kernels.py).  The layers of ``torch_rechub_amd.basic`` call the SAME
C entry points through ``autograd.Function`` (no dispatcher hop per call in the eager / hipGraph step); the embedding
gather, its backward and the optimizer are deliberately NOT pure ops -- they own persistent table-gradient buffers and
claim words, i.e. hidden state a functional schema cannot express -- and stay ``autograd.Function`` + C ABI.

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

"""DIEN's AUGRU at the DIN-shape configuration (B = 4096, T = 100, D = 16): the HIP recurrence (csrc/augru.hip) against the
per-step composition (one state product + elementwise ops per step under autograd), forward + backward, HIP events."""
import sys

import torch

sys.path.insert(0, ".")
from torch_rechub_amd import ops  # noqa: E402
from torch_rechub_amd.models.ranking.dien import AUGRU  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device("cuda:0")
    B, T, D = 4096, 100, 16
    torch.manual_seed(0)
    net = AUGRU(D).to(dev)
    x = torch.randn(B, T, D, device=dev, requires_grad=True)
    item = torch.randn(B, D, device=dev)
    lens = torch.randint(1, T + 1, (B,), device=dev)
    mask = torch.arange(T, device=dev)[None, :] < lens[:, None]

    def step():
        outs, h = net(x, item, mask)
        (outs.sum() + h.sum()).backward()

    fused = timed(step, 20)
    ok = ops.augru_ok
    ops.augru_ok = lambda *a: False  # the per-step composition
    loop = timed(step, 3)
    ops.augru_ok = ok
    xw = torch.randn(B, T, 3 * D, device=dev)
    attn = torch.rand(B, T, device=dev)
    U = torch.randn(D, 3 * D, device=dev) * 0.1
    fwd = timed(lambda: ops.augru(xw, attn, U), 50)
    xw.requires_grad_(True)

    def fb():
        ops.augru(xw, attn, U).sum().backward()

    both = timed(fb, 50)
    state_bytes = B * T * (3 * D + D + 1) * 4
    print(f"AUGRU layer fwd+bwd B={B} T={T} D={D}: HIP recurrence {fused:.3f} ms, per-step composition {loop:.1f} ms "
          f"({loop / fused:.0f}x)")
    print(f"rh_augru_fwd alone {fwd * 1e3:.0f} us ({state_bytes / fwd / 1e6:.0f} GB/s of xw + states), "
          f"fwd + bwd kernels + sum {both * 1e3:.0f} us")


if __name__ == "__main__":
    main()

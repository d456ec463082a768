"""Full-size checks on a real MI355X through size-independent properties (the numpy oracle would take minutes here):
the BASELINE.json configs[1] shape (26 Criteo tables, 33.76 M rows, B = 4096 .. 65536) and a configs[4]-size table
(100 M rows x 16 = 6.4 GB, byte offsets beyond 2^32)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CRITEO_VOCABS = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306,
                 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def criteo_tables():
    g = torch.Generator(device=dev()).manual_seed(7)
    return [torch.nn.Parameter(torch.randn(v, 16, device=dev(), generator=g) * 0.05) for v in CRITEO_VOCABS]


def batch(B, seed):
    g = torch.Generator(device=dev()).manual_seed(seed)
    idx = torch.stack([torch.randint(0, v, (B,), device=dev(), generator=g) for v in CRITEO_VOCABS], 1)
    dense = torch.rand(B, 13, device=dev(), generator=g)
    return idx, dense


@pytest.mark.parametrize("B", [4096, 65536])
def test_full_shape_gather_is_a_copy_and_fm_lr_properties(criteo_tables, B):
    from torch_rechub_amd import ops
    F, D = 26, 16
    idx, dense = batch(B, B)
    lr_w = torch.randn(1, F * D, device=dev())
    lr_b = torch.zeros(1, device=dev())
    call = ops.EmbedCall(criteo_tables, [None] * F, [idx[:, f] for f in range(F)], [dense[:, j] for j in range(13)],
                         want_fm=True, want_lr=True)
    with torch.no_grad():
        out, fm, lr = ops.fused_embedding(call, lr_w, lr_b)
        # gather == advanced indexing of the same tables, bit for bit; dense block is a copy
        for f in (0, 2, 8, 11, 20, 25):
            assert torch.equal(out[:, f * D:(f + 1) * D], criteo_tables[f].detach()[idx[:, f]])
        assert torch.equal(out[:, F * D:], dense)
        emb = out[:, :F * D].view(B, F, D).double()
        fm_ref = 0.5 * ((emb.sum(1)**2).sum(1) - (emb**2).sum((1, 2)))
        # fm is a difference of two ~1.0 sums: absolute error is a few ulp of those, not of the (small) result
        np.testing.assert_allclose(fm.squeeze(1).cpu().numpy(), fm_ref.cpu().numpy(), rtol=2e-5, atol=2e-6)
        # LR is linear in its weights: lr(2w) = 2 lr(w) with zero bias, exactly (power-of-two scaling)
        _, _, lr2 = ops.fused_embedding(call, lr_w * 2, lr_b)
        assert torch.equal(lr2, lr * 2)
    ops.check_errors()


def test_full_shape_backward_conserves_gradient_mass_and_touches_only_looked_up_rows(criteo_tables):
    from torch_rechub_amd import ops
    B, F, D = 4096, 26, 16
    idx, dense = batch(B, 11)
    call = ops.EmbedCall(criteo_tables, [None] * F, [idx[:, f] for f in range(F)], want_fm=True)
    out, fm, _ = ops.fused_embedding(call)
    g_out = torch.randn(B, F * D, device=dev())
    g_fm = torch.randn(B, 1, device=dev())
    torch.autograd.backward([out, fm], [g_out, g_fm])
    emb = out.detach().view(B, F, D)
    rows = g_out.view(B, F, D) + g_fm.view(B, 1, 1) * (emb.sum(1, keepdim=True) - emb)
    for f in range(F):
        grad = criteo_tables[f].grad
        # column sums are conserved by a scatter-add, whatever the duplicates
        np.testing.assert_allclose(grad.double().sum(0).cpu().numpy(), rows[:, f].double().sum(0).cpu().numpy(), rtol=1e-4,
                                   atol=1e-4)
        touched = torch.unique(idx[:, f])
        nz = (grad != 0).any(dim=1)
        assert int(nz.sum()) <= touched.numel() and bool(nz[touched].float().mean() > 0.99)
        mask = torch.ones(grad.shape[0], dtype=torch.bool, device=dev())
        mask[touched] = False
        assert not bool((grad[mask] != 0).any())  # rows that were not looked up stay exactly zero
    for w in criteo_tables:
        ops.grad_buffer(w).zero_()
        w._rh_dirty = False
        w.grad = None


@pytest.mark.parametrize("overlap", [False, True])
def test_full_shape_lazy_adam_equals_dense_adam_bitwise(criteo_tables, overlap):
    """Five steps at the real table sizes: blocked-lazy exact Adam == dense pass, bit for bit, after the flush -- also
    when the window sweep of each step is deferred to a side stream and runs concurrently with the next step's
    (refreshed) lookups and gradient writes."""
    from torch_rechub_amd import ops
    from torch_rechub_amd.optim import TableAdam
    F = 26
    A = [torch.nn.Parameter(t.detach().clone()) for t in criteo_tables]
    Bp = [torch.nn.Parameter(t.detach().clone()) for t in criteo_tables]
    dense = TableAdam(A, table_params=A, lr=1e-3, weight_decay=1e-5)
    lazy = TableAdam(Bp, table_params=Bp, lr=1e-3, weight_decay=1e-5, lazy_k=2)
    lazy.overlap_sweep = overlap
    for step in range(5):
        idx, _ = batch(4096, 100 + step)
        cols = [idx[:, f] for f in range(F)]
        key = tuple([c.data_ptr() for c in cols] + [idx.stride(0)] * F + list(range(F)))
        idesc = ops.EmbedCall._icache.get(key, dev())
        ops._pre_gather(Bp, [None] * F, idesc, 1, 4096, F, 16)  # what the forward does: refresh (+ fork the sweep)
        assert lazy._sweep_inflight == (overlap and step > 0)
        g_rows = torch.randn(4096, 16, device=dev())
        for f in range(F):
            dense_g = torch.zeros_like(A[f])
            dense_g.index_add_(0, idx[:, f], g_rows)  # computed ONCE (atomic order varies), then given to both
            for P in (A[f], Bp[f]):
                ops.grad_buffer(P).copy_(dense_g)
                P._rh_dirty = True
            del dense_g
        ops._log_touch(Bp, [None] * F, idesc, 1, 4096, F, 16, cols)
        lazy.step()
        dense.step()
        assert lazy._sweep_inflight == (overlap and step > 0) and lazy._sweep_pending == overlap  # joined by the NEXT refresh
    lazy.flush()
    torch.cuda.synchronize()
    for f in range(F):
        assert torch.equal(A[f].detach(), Bp[f].detach()), f"table {f}"
        assert torch.equal(dense.state[A[f]]["exp_avg_sq"], lazy.state[Bp[f]]["exp_avg_sq"]), f"table {f} v"
    before = [p.detach().clone() for p in Bp[:3]]
    lazy.flush()  # idempotent
    assert all(torch.equal(b, p.detach()) for b, p in zip(before, Bp))


def test_hundred_million_row_table_uses_64_bit_offsets():
    """configs[4] scale: one (100 M, 16) table = 6.4 GB; the last rows sit beyond a 32-bit byte/element offset."""
    from torch_rechub_amd import ops
    from torch_rechub_amd.optim import TableAdam
    V, D, B = 100_000_000, 16, 4096
    table = torch.nn.Parameter(torch.zeros(V, D, device=dev()))
    with torch.no_grad():
        table[V - 5:] = torch.arange(5 * D, device=dev(), dtype=torch.float32).view(5, D) + 1
        table[:3] = -1.0
    idx = torch.randint(0, V, (B,), device=dev())
    idx[:5] = torch.arange(V - 5, V, device=dev())
    idx[5:8] = torch.arange(3, device=dev())
    call = ops.EmbedCall([table], [None], [idx])
    out, _, _ = ops.fused_embedding(call)
    assert torch.equal(out[:5].detach(), table.detach()[V - 5:]) and torch.all(out[5:8] == -1)
    out.backward(torch.ones_like(out))
    grad = table.grad
    assert torch.all(grad[V - 5:] == 1) and float(grad.sum()) == B * D
    opt = TableAdam([table], table_params=[table], lr=0.5, weight_decay=0.0, lazy_k=4)
    ops._log_touch([table], [None], call.idesc(), 1, B, 1, D, [idx])
    opt.step()
    opt.flush()
    torch.cuda.synchronize()
    # first Adam step with bias correction moves every touched element by exactly -lr * sign(g) (up to eps)
    np.testing.assert_allclose((table.detach()[V - 5:] - (torch.arange(5 * D, device=dev()).view(5, D) + 1)).cpu().numpy(),
                               -0.5, rtol=1e-5)
    assert float(ops.grad_buffer(table).abs().sum()) == 0.0
    ops.check_errors()

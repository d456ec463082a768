"""Kernel-level parity on a real MI355X: every HIP kernel (through the C ABI, via torch_rechub_amd.ops) against
the numpy oracle (oracle/ctr_oracle.py, float64) on identical seeded inputs, plus the reference golden vectors.

Tolerances (fp32 path vs float64 oracle; reduction order differs, nothing else):
  gathers / copies: bit-exact;   FM / LR / cross / pooling outputs and gradients: rtol 1e-5 (+ atol scaled to the
  magnitude of the compared array);   Adam: rtol 2e-6 per step.
"""
import numpy as np
import pytest
import torch

from oracle import ctr_oracle as O

pytestmark = pytest.mark.gpu
F64 = np.float64


def dev():
    return torch.device("cuda:0")


def close(a, b, rtol=1e-5, atol_scale=1e-6, what=""):
    a = a.detach().cpu().numpy().astype(F64) if torch.is_tensor(a) else np.asarray(a, F64)
    b = np.asarray(b, F64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    atol = atol_scale * max(1.0, float(np.abs(b).max()) if b.size else 1.0)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=what)


def make_case(B, vocabs, D, ND, seed, idx_dtype=torch.int64, packed=True, shared=None, pads=None, std=0.5):
    g = torch.Generator().manual_seed(seed)
    vocabs = list(vocabs)
    tables = [torch.randn(v, D, generator=g) * std for v in vocabs]
    if shared:
        for f, src in shared.items():
            tables[f] = tables[src]
            vocabs[f] = vocabs[src]  # a shared_with field indexes the owner's table
    pads = pads or [None] * len(vocabs)
    for t, p in zip(tables, pads):
        if p is not None:
            t[p].zero_()
    idx = torch.stack([torch.randint(0, v, (B,), generator=g) for v in vocabs], 1).to(idx_dtype)
    dense = torch.rand(B, ND, generator=g)
    d = dev()
    dev_tables = {}
    wts = []
    for t in tables:
        if id(t) not in dev_tables:
            dev_tables[id(t)] = torch.nn.Parameter(t.to(d))
        wts.append(dev_tables[id(t)])
    if packed:
        idx_d = idx.to(d)
        idx_cols = [idx_d[:, f] for f in range(len(vocabs))]
        dense_d = dense.to(d)
        dense_cols = [dense_d[:, j] for j in range(ND)]
    else:
        idx_cols = [idx[:, f].contiguous().to(d) for f in range(len(vocabs))]
        dense_cols = [dense[:, j].contiguous().to(d) for j in range(ND)]
    np_by_id = {}
    for t in tables:  # shared tables stay ONE numpy array so that the oracle accumulates their gradient jointly
        np_by_id.setdefault(id(t), t.numpy().astype(F64))
    np_tables = [np_by_id[id(t)] for t in tables]
    return dict(tables=tables, np_tables=np_tables, wts=wts, pads=pads, idx=idx, idx_cols=idx_cols, dense=dense,
                dense_cols=dense_cols, B=B, F=len(vocabs), D=D, ND=ND, g=g)


CRITEO_LIKE = [3, 4, 10, 27, 105, 305, 583, 40, 1460, 24, 18, 15, 633, 5000, 20000, 3, 12517, 2173, 4, 93145, 10, 5652,
               7, 14992, 286181, 142572]


@pytest.mark.parametrize("B,vocabs,D,ND,idt,packed,fs", [
    (4096, CRITEO_LIKE, 16, 13, torch.int64, True, 0),
    (4096, CRITEO_LIKE, 16, 13, torch.int64, False, 1),
    (1000, CRITEO_LIKE, 16, 13, torch.int32, True, 2),
    (257, CRITEO_LIKE[:7], 16, 0, torch.int64, True, 8),
    (1, [5, 9], 16, 1, torch.int64, True, 0),
    (63, [11], 16, 2, torch.int64, False, 0),
    (300, [50] * 39, 16, 0, torch.int64, True, 4),
    (129, [7, 300, 41], 4, 3, torch.int64, True, 0),
    (129, [7, 300, 41], 8, 0, torch.int32, False, 0),
    (200, [7, 300, 41, 9], 32, 5, torch.int64, True, 0),
    (77, [7, 300, 41], 64, 1, torch.int64, True, 0),
    (33, [7, 300], 128, 2, torch.int64, True, 0),
])
def test_embed_fwd(B, vocabs, D, ND, idt, packed, fs):
    from torch_rechub_amd import ops
    c = make_case(B, vocabs, D, ND, seed=B + D, idx_dtype=idt, packed=packed)
    g = c["g"]
    F = c["F"]
    lr_w = torch.randn(1, F * D, generator=g).to(dev())
    lr_b = torch.randn(1, generator=g).to(dev())
    call = ops.EmbedCall(c["wts"], c["pads"], c["idx_cols"], c["dense_cols"], want_fm=True, want_lr=True,
                         field_split=fs)
    out, fm, lr = ops.fused_embedding(call, lr_w, lr_b)
    torch.cuda.synchronize()
    ops.check_errors()
    deep, y_fm, y_lr = O.deepfm_sparse_part(c["np_tables"], c["idx"].numpy(), lr_w.cpu().numpy().astype(F64),
                                            lr_b.cpu().numpy().astype(F64), c["dense"].numpy().astype(F64))
    assert out.shape == (B, F * D + ND)
    # gather + dense concat are pure copies: bit-exact against the fp32 tables
    exact = O.embedding_layer_squeeze([t.numpy() for t in c["tables"]], c["idx"].numpy(), c["dense"].numpy())
    assert np.array_equal(out.detach().cpu().numpy(), exact)
    close(fm, y_fm, what="fm")
    close(lr, y_lr, what="lr")


@pytest.fixture
def uniform_fwd_kernel():
    """Force the field-uniform forward kernel (the automatic choice from B * F >= 12288 * 26 on): RH_TUNE_FWD_PATH = 2."""
    from torch_rechub_amd import _lib
    _lib.call("rh_set_tuning", 7, 2)
    yield
    _lib.call("rh_set_tuning", 7, 0)


@pytest.mark.parametrize("B,vocabs,D,ND,idt,packed,want_lr", [
    (4096, CRITEO_LIKE, 16, 13, torch.int64, True, True),
    (1000, CRITEO_LIKE, 16, 13, torch.int32, False, True),
    (17, CRITEO_LIKE, 16, 21, torch.int64, True, True),      # more dense columns than the lanes prefetch (tail loop)
    (300, [50] * 39, 16, 0, torch.int64, True, True),        # 39 fields: a wavefront walks 10 > 8 of them (two phases)
    (300, [50] * 70, 8, 3, torch.int64, True, False),        # 18 fields per wavefront: three phases, no LR
    (1, [5, 9], 16, 1, torch.int64, True, True),             # fewer fields than wavefronts: two of them idle
    (63, [11], 16, 2, torch.int64, False, True),
    (129, [7, 300, 41], 4, 3, torch.int64, True, True),
    (129, [7, 300, 41], 8, 0, torch.int32, False, True),
    (200, [7, 300, 41, 9], 32, 5, torch.int64, True, True),
    (77, [7, 300, 41], 64, 1, torch.int64, True, True),
    (33, [7, 300], 128, 2, torch.int64, True, False),
])
def test_embed_fwd_field_uniform_kernel(uniform_fwd_kernel, B, vocabs, D, ND, idt, packed, want_lr):
    from torch_rechub_amd import ops
    c = make_case(B, vocabs, D, ND, seed=B + D + 1, idx_dtype=idt, packed=packed)
    g = c["g"]
    F = c["F"]
    lr_w = torch.randn(1, F * D, generator=g).to(dev())
    lr_b = torch.randn(1, generator=g).to(dev())
    call = ops.EmbedCall(c["wts"], c["pads"], c["idx_cols"], c["dense_cols"], want_fm=True, want_lr=want_lr)
    out, fm, lr = ops.fused_embedding(call, *((lr_w, lr_b) if want_lr else ()))
    torch.cuda.synchronize()
    ops.check_errors()
    deep, y_fm, y_lr = O.deepfm_sparse_part(c["np_tables"], c["idx"].numpy(), lr_w.cpu().numpy().astype(F64),
                                            lr_b.cpu().numpy().astype(F64), c["dense"].numpy().astype(F64))
    exact = O.embedding_layer_squeeze([t.numpy() for t in c["tables"]], c["idx"].numpy(), c["dense"].numpy())
    assert np.array_equal(out.detach().cpu().numpy(), exact)
    close(fm, y_fm, what="fm")
    if want_lr:
        close(lr, y_lr, what="lr")
    from torch_rechub_amd import _lib
    _lib.call("rh_set_tuning", 7, 1)  # the lane-split kernel on the same call
    out1, fm1, _ = ops.fused_embedding(call, *((lr_w, lr_b) if want_lr else ()))
    assert torch.equal(out1, out)
    close(fm1, y_fm, what="fm (lane-split kernel)")


def test_embed_fwd_field_uniform_kernel_flags_bad_indices(uniform_fwd_kernel):
    from torch_rechub_amd import ops
    c = make_case(64, [5, 6, 7, 8, 9], 16, 0, seed=5)
    c["idx_cols"][4][7] = 9  # == vocab, in the last wavefront's share of the fields
    ops.fused_embedding(ops.EmbedCall(c["wts"], c["pads"], c["idx_cols"]))
    with pytest.raises(IndexError):
        ops.check_errors()
    c["idx_cols"][4][7] = -1
    ops.fused_embedding(ops.EmbedCall(c["wts"], c["pads"], c["idx_cols"]))
    with pytest.raises(IndexError):
        ops.check_errors()


def test_embed_fwd_plain_gather_no_fm_no_lr_and_3d_view():
    from torch_rechub_amd import ops
    c = make_case(500, [9, 1000, 3], 16, 0, seed=3)
    call = ops.EmbedCall(c["wts"], c["pads"], c["idx_cols"])
    out, fm, lr = ops.fused_embedding(call)
    assert fm is None and lr is None
    assert np.array_equal(out.detach().view(500, 3, 16).cpu().numpy(),
                          O.embedding_gather([t.numpy() for t in c["tables"]], c["idx"].numpy()))


def test_embed_index_out_of_range_raises_index_error():
    from torch_rechub_amd import ops
    c = make_case(64, [5, 6], 16, 0, seed=5)
    c["idx_cols"][1][7] = 6  # == vocab
    call = ops.EmbedCall(c["wts"], c["pads"], c["idx_cols"])
    ops.fused_embedding(call)
    with pytest.raises(IndexError):
        ops.check_errors()
    c["idx_cols"][1][7] = -1
    ops.fused_embedding(ops.EmbedCall(c["wts"], c["pads"], c["idx_cols"]))
    with pytest.raises(IndexError):
        ops.check_errors()
    ops.check_errors()  # flag was cleared


@pytest.mark.parametrize("B,vocabs,D,shared,pads,spb", [
    (4096, CRITEO_LIKE, 16, None, None, 0),
    (1000, [3, 50, 700, 3, 9000], 16, {3: 0}, [None, 0, None, None, 5], 0),
    (130, [4, 600], 16, None, None, 64),
    (515, [4, 600, 31], 32, None, [1, None, None], 128),
    (99, [4, 600, 31], 8, {2: 0}, None, 0),
    (64, [40], 64, None, None, 0),
    (1, [40, 3], 16, None, None, 0),
])
@pytest.mark.parametrize("fwd_path", [1, 2])  # the forward kernel that saved S (the FM backward's per-sample field sum)
def test_embed_bwd(B, vocabs, D, shared, pads, spb, fwd_path):
    from torch_rechub_amd import _lib, ops
    _lib.call("rh_set_tuning", 7, fwd_path)
    try:
        _embed_bwd_case(B, vocabs, D, shared, pads, spb)
    finally:
        _lib.call("rh_set_tuning", 7, 0)


@pytest.mark.parametrize("B,vocabs,shared,pads", [
    (20000, CRITEO_LIKE, None, None),
    (9001, [3, 50, 700, 3, 9000, 31, 12], {3: 0}, [None, 0, None, None, 5, None, None]),
    (8200, [40], None, None),
    (8193, [7, 31], None, None),  # every field on the small-table path
])
def test_embed_bwd_large_batch(B, vocabs, shared, pads):
    """B > 8192: the launch without the chain's priority (and the batch sizes of the roofline sweep)."""
    _embed_bwd_case(B, vocabs, 16, shared, pads, 0)


def _embed_bwd_case(B, vocabs, D, shared, pads, spb):
    from torch_rechub_amd import ops
    ND = 2
    c = make_case(B, vocabs, D, ND, seed=B + 7 * D, shared=shared, pads=pads)
    g = c["g"]
    F = c["F"]
    lr_w = torch.nn.Parameter(torch.randn(1, F * D, generator=g).to(dev()))
    lr_b = torch.nn.Parameter(torch.randn(1, generator=g).to(dev()))
    call = ops.EmbedCall(c["wts"], c["pads"], c["idx_cols"], c["dense_cols"], want_fm=True, want_lr=True,
                         samples_per_block=spb)
    out, fm, lr = ops.fused_embedding(call, lr_w, lr_b)
    g_deep = torch.randn(B, F * D + ND, generator=g)
    g_fm = torch.randn(B, 1, generator=g)
    g_lr = torch.randn(B, 1, generator=g)
    torch.autograd.backward([out, fm, lr], [g_deep.to(dev()), g_fm.to(dev()), g_lr.to(dev())])
    torch.cuda.synchronize()
    ops.check_errors()
    grads, gw, gb, rows = O.deepfm_sparse_part_backward(c["np_tables"], c["idx"].numpy(),
                                                        lr_w.detach().cpu().numpy().astype(F64),
                                                        g_deep.numpy().astype(F64), g_fm.numpy().astype(F64),
                                                        g_lr.numpy().astype(F64), padding_idx=c["pads"])
    for f, w in enumerate(c["wts"]):
        assert w.grad is not None and w.grad.data_ptr() == ops.grad_buffer(w).data_ptr()
        close(w.grad, grads[f], rtol=2e-5, atol_scale=2e-6, what=f"table grad field {f}")
        if c["pads"][f] is not None:
            assert torch.all(w.grad[c["pads"][f]] == 0)
    close(lr_w.grad, gw, rtol=2e-5, atol_scale=2e-6, what="lr_w grad")
    close(lr_b.grad, gb, rtol=2e-5, atol_scale=2e-6, what="lr_b grad")


def test_embed_bwd_rows_sink_and_scatter_rows_equal_fused_scatter():
    """sink=1 (rows for the data-parallel exchange) followed by rh_embed_scatter_rows == sink=0."""
    from torch_rechub_amd import _lib, ops
    B, D = 700, 16
    vocabs = [3, 50, 7000, 12]
    c = make_case(B, vocabs, D, 0, seed=11, pads=[None, 1, None, None])
    F = c["F"]
    g = c["g"]
    call = ops.EmbedCall(c["wts"], c["pads"], c["idx_cols"], want_fm=True)
    out, fm, _ = ops.fused_embedding(call)
    g_out = torch.randn(B, F * D, generator=g).to(dev())
    g_fm = torch.randn(B, generator=g).to(dev())
    s_sum = out.detach().view(B, F, D).sum(1).contiguous()
    rows = torch.empty(B, F, D, device=dev())
    _lib.call("rh_embed_bwd", ops._p(call.fdesc(False)), ops._p(call.idesc()), 1, B, F, D, ops._p(g_out),
              g_out.stride(0), ops._p(out), out.stride(0), ops._p(s_sum), ops._p(g_fm), ops._p(None), ops._p(None),
              ops._p(None), 1.0, 1, ops._p(rows), 0, ops._p(ops.err_flag(dev())), ops._stream())
    _, _, _, rows_ref = O.deepfm_sparse_part_backward(c["np_tables"], c["idx"].numpy(), np.zeros((1, F * D)),
                                                      g_out.cpu().numpy().astype(F64), g_fm.cpu().numpy().astype(F64),
                                                      None, padding_idx=c["pads"])
    close(rows, rows_ref, what="rows sink")
    idx_all = torch.stack(c["idx_cols"], 1).contiguous()
    ops.scatter_rows(call, idx_all, rows)
    torch.cuda.synchronize()
    want = O.embedding_backward([t.shape for t in c["np_tables"]], c["idx"].numpy(), rows_ref, c["pads"])
    for f, w in enumerate(c["wts"]):
        close(ops.grad_buffer(w), want[f], rtol=2e-5, atol_scale=2e-6, what=f"scatter field {f}")


def test_embedding_golden_vectors(layers_golden):
    """The reference's own EmbeddingLayer output / gradients (shared table + padding_idx + dense) on the HIP path."""
    from torch_rechub_amd.basic.features import DenseFeature, SparseFeature
    from torch_rechub_amd.basic.layers import EmbeddingLayer
    g = layers_golden
    D = 16
    vocabs = [int(v) for v in g["emb.vocabs"]]
    feas = [SparseFeature(f"s{i}", vocabs[i], D) for i in range(5)]
    feas[3] = SparseFeature("s3", 7, D, padding_idx=0)
    feas.append(SparseFeature("s5", 11, D, shared_with="s1"))
    dense = [DenseFeature(f"d{i}") for i in range(3)]
    layer = EmbeddingLayer(dense + feas).to(dev())
    with torch.no_grad():
        for k, m in layer.embed_dict.items():
            m.weight.copy_(torch.from_numpy(g[f"emb.table.{k}"]))
    x = {f.name: torch.from_numpy(g["emb.idx"][:, i]).to(dev()) for i, f in enumerate(feas)}
    x.update({f.name: torch.from_numpy(g["emb.dense"][:, i]).to(dev()) for i, f in enumerate(dense)})
    sq = layer(x, dense + feas, squeeze_dim=True)
    assert np.array_equal(sq.detach().cpu().numpy(), g["emb.out_squeeze"])
    assert np.array_equal(layer(x, feas, squeeze_dim=False).detach().cpu().numpy(), g["emb.out_3d"])
    sq.backward(torch.from_numpy(g["emb.g_squeeze"]).to(dev()))
    for k, m in layer.embed_dict.items():
        close(m.weight.grad, g[f"emb.grad.{k}"], rtol=2e-5, atol_scale=2e-6, what=k)


@pytest.mark.parametrize("B,F,D", [(37, 6, 16), (1000, 26, 16), (65, 3, 7), (9, 5, 40), (3, 2, 100)])
@pytest.mark.parametrize("rs", [True, False])
def test_fm_standalone(B, F, D, rs):
    from torch_rechub_amd import ops
    g = torch.Generator().manual_seed(B + F + D)
    x = torch.randn(B, F, D, generator=g)
    xd = x.to(dev()).requires_grad_(True)
    y = ops.fm(xd, rs)
    close(y, O.fm_forward(x.numpy().astype(F64), rs), what="fm out")
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy.to(dev()))
    close(xd.grad, O.fm_backward(x.numpy().astype(F64), gy.numpy().astype(F64), rs), what="fm gx")


def test_fm_golden(layers_golden):
    from torch_rechub_amd.basic.layers import FM
    g = layers_golden
    x = torch.from_numpy(g["fm.x"]).to(dev()).requires_grad_(True)
    for rs in (1, 0):
        y = FM(reduce_sum=bool(rs))(x)
        close(y, g[f"fm.{rs}.out"], what="fm golden out")
        (gx,) = torch.autograd.grad(y, x, torch.from_numpy(g[f"fm.{rs}.g"]).to(dev()))
        close(gx, g[f"fm.{rs}.gx"], what="fm golden gx")


@pytest.mark.parametrize("pooling", ["sum", "mean", "concat"])
@pytest.mark.parametrize("tag,pad", [("pad0", 0), ("nopad", None)])
def test_seq_pool_golden(layers_golden, pooling, tag, pad):
    from torch_rechub_amd import ops
    g = layers_golden
    k = f"seq.{pooling}.{tag}"
    w = torch.nn.Parameter(torch.from_numpy(g[k + ".table"]).to(dev()))
    idx = torch.from_numpy(g[k + ".idx"]).to(dev())
    y = ops.seq_pool(w, idx, pooling, pad)
    ref = g[k + ".out"][:, 0]
    if pooling == "concat":
        assert np.array_equal(y.detach().cpu().numpy(), ref)
    else:
        close(y, ref, what=k)
    y.backward(torch.from_numpy(g[k + ".g"][:, 0]).to(dev()))
    close(w.grad, g[k + ".grad"], rtol=2e-5, atol_scale=2e-6, what=k + " grad")


@pytest.mark.parametrize("B,L,V,D,pooling,pad,idt", [
    (300, 100, 5000, 16, "concat", 0, torch.int64),
    (300, 100, 5000, 16, "mean", 0, torch.int64),
    (257, 50, 800, 16, "sum", None, torch.int32),
    (64, 7, 30, 64, "mean", 0, torch.int64),
    (5, 1, 30, 8, "sum", 0, torch.int64),
    (40, 33, 100, 32, "concat", None, torch.int64),
])
def test_seq_pool_vs_oracle(B, L, V, D, pooling, pad, idt):
    from torch_rechub_amd import ops
    g = torch.Generator().manual_seed(B + L)
    table = torch.randn(V, D, generator=g) * 0.5
    if pad is not None:
        table[pad].zero_()
    idx = torch.randint(1, V, (B, L), generator=g)
    lens = torch.randint(0, L + 1, (B,), generator=g)  # includes fully padded rows (count 0 -> mean 0)
    idx[torch.arange(L)[None, :] >= lens[:, None]] = 0
    w = torch.nn.Parameter(table.to(dev()))
    y = ops.seq_pool(w, idx.to(idt).to(dev()), pooling, pad)
    ref = O.seq_pool(table.numpy().astype(F64), idx.numpy(), pooling, pad)
    close(y, ref, what="seq out")
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy.to(dev()))
    close(w.grad, O.seq_pool_backward(table.shape, idx.numpy(), pooling, gy.numpy().astype(F64), pad), rtol=2e-5,
          atol_scale=2e-6, what="seq grad")


@pytest.mark.parametrize("nl", [1, 3, 6])
def test_cross_network_golden(layers_golden, nl):
    from torch_rechub_amd.basic.layers import CrossNetwork
    g = layers_golden
    d = g["cross.x"].shape[1]
    cn = CrossNetwork(d, nl).to(dev())
    with torch.no_grad():
        for l in range(nl):
            cn.w[l].weight.copy_(torch.from_numpy(g[f"cross.{nl}.W"][l:l + 1]))
            cn.b[l].copy_(torch.from_numpy(g[f"cross.{nl}.B"][l]))
    x = torch.from_numpy(g["cross.x"]).to(dev()).requires_grad_(True)
    y = cn(x)
    close(y, g[f"cross.{nl}.out"], rtol=2e-5, atol_scale=2e-6, what="cross out")
    y.backward(torch.from_numpy(g[f"cross.{nl}.g"]).to(dev()))
    close(x.grad, g[f"cross.{nl}.gx"], rtol=1e-4, atol_scale=1e-5, what="cross gx")
    gW = torch.cat([w.weight.grad for w in cn.w], 0)
    gB = torch.stack([b.grad for b in cn.b], 0)
    close(gW, g[f"cross.{nl}.gW"], rtol=1e-4, atol_scale=1e-5, what="cross gW")
    close(gB, g[f"cross.{nl}.gB"], rtol=1e-4, atol_scale=1e-5, what="cross gB")


@pytest.mark.parametrize("B,d,L", [(4096, 429, 3), (1000, 429, 4), (333, 64, 2), (50, 1, 1), (129, 65, 3), (70, 600, 2),
                                   (33, 1100, 1), (17, 2048, 1), (64, 429, 7), (5, 600, 5)])
def test_cross_network_vs_oracle(B, d, L):
    from torch_rechub_amd import ops
    g = torch.Generator().manual_seed(B + d + L)
    x = torch.randn(B, d, generator=g)
    W = torch.randn(L, d, generator=g) / np.sqrt(d)
    Bv = torch.randn(L, d, generator=g) * 0.1
    xd, Wd, Bd = (t.to(dev()).requires_grad_(True) for t in (x, W, Bv))
    y = ops.cross_network(xd, Wd, Bd)
    close(y, O.cross_network_forward(x.numpy().astype(F64), W.numpy().astype(F64), Bv.numpy().astype(F64)), rtol=2e-5,
          atol_scale=2e-6, what="cross out")
    gy = torch.randn(B, d, generator=g)
    y.backward(gy.to(dev()))
    gx, gW, gB = O.cross_network_backward(x.numpy().astype(F64), W.numpy().astype(F64), Bv.numpy().astype(F64),
                                          gy.numpy().astype(F64))
    close(xd.grad, gx, rtol=1e-4, atol_scale=1e-5, what="cross gx")
    close(Wd.grad, gW, rtol=1e-4, atol_scale=1e-5, what="cross gW")
    close(Bd.grad, gB, rtol=1e-4, atol_scale=1e-5, what="cross gB")


def test_cross_v2_and_mix_golden(layers_golden):
    from torch_rechub_amd.basic.layers import CrossNetMix, CrossNetV2
    g = layers_golden
    x = torch.from_numpy(g["cross.x"]).to(dev())
    d = x.shape[1]
    v2 = CrossNetV2(d, 2).to(dev())
    with torch.no_grad():
        for l in range(2):
            v2.w[l].weight.copy_(torch.from_numpy(g["crossv2.W"][l]))
            v2.b[l].copy_(torch.from_numpy(g["crossv2.B"][l]))
    close(v2(x), g["crossv2.out"], rtol=1e-4, atol_scale=1e-5, what="crossv2")
    mix = CrossNetMix(d, num_layers=2, low_rank=8, num_experts=3).to(dev())
    with torch.no_grad():
        for l in range(2):
            mix.u_list[l].copy_(torch.from_numpy(g["crossmix.U"][l]))
            mix.v_list[l].copy_(torch.from_numpy(g["crossmix.V"][l]))
            mix.c_list[l].copy_(torch.from_numpy(g["crossmix.C"][l]))
            mix.bias[l].copy_(torch.from_numpy(g["crossmix.bias"][l]).view(-1, 1))
        for e in range(3):
            mix.gating[e].weight.copy_(torch.from_numpy(g["crossmix.Wg"][e:e + 1]))
    close(mix(x), g["crossmix.out"], rtol=1e-4, atol_scale=1e-5, what="crossmix")


def test_dice_golden(layers_golden):
    from torch_rechub_amd.basic.activation import Dice
    g = layers_golden
    dice = Dice().to(dev())
    with torch.no_grad():
        dice.alpha.copy_(torch.from_numpy(g["dice.alpha"]))
    close(dice(torch.from_numpy(g["dice.x"]).to(dev())), g["dice.out"], what="dice")


@pytest.mark.parametrize("sm", [0, 1])
def test_activation_unit_golden(layers_golden, sm):
    from torch_rechub_amd.models.ranking.din import ActivationUnit
    g = layers_golden
    k = f"au.{sm}"
    au = ActivationUnit(16, dims=[12, 6], activation="dice", use_softmax=bool(sm))
    au.load_state_dict({n[len(k + ".sd."):]: torch.from_numpy(g[n]) for n in g.files if n.startswith(k + ".sd.")})
    au.to(dev()).eval()
    out = au(torch.from_numpy(g[k + ".hist"]).to(dev()), torch.from_numpy(g[k + ".tgt"]).to(dev()))
    close(out, g[k + ".out"], rtol=1e-4, atol_scale=1e-5, what="activation unit")


# ----------------------------------------------------------------------------------------------------------
def run_table_adam(shapes, steps, lr, wd, seed, grad_prob=0.5):
    """TableAdam on bare 'tables' with random sparse-ish gradients; returns (params, oracle params)."""
    from torch_rechub_amd import ops
    from torch_rechub_amd.optim import TableAdam
    g = torch.Generator().manual_seed(seed)
    ps = [torch.nn.Parameter((torch.randn(s, generator=g) * 0.1).to(dev())) for s in shapes]
    opt = TableAdam(ps, table_params=ps, lr=lr, weight_decay=wd)
    ref = [p.detach().cpu().numpy().astype(F64) for p in ps]
    ms = [np.zeros_like(r) for r in ref]
    vs = [np.zeros_like(r) for r in ref]
    for t in range(1, steps + 1):
        for i, p in enumerate(ps):
            gr = torch.randn(shapes[i], generator=g)
            rows = (torch.rand(shapes[i][0], generator=g) < grad_prob).float().unsqueeze(1)
            gr = gr * rows  # untouched rows have exactly zero data gradient
            buf = ops.grad_buffer(p)
            assert torch.all(buf == 0)  # re-zeroed by the previous step
            buf.copy_(gr.to(dev()))
            p._rh_dirty = True
            ref[i], ms[i], vs[i] = O.adam_step(ref[i], gr.numpy().astype(F64), ms[i], vs[i], t, lr=lr, weight_decay=wd)
        opt.step()
    torch.cuda.synchronize()
    return ps, ref, opt, ms, vs


@pytest.mark.parametrize("shapes,steps,lr,wd", [
    ([(3, 16), (1000, 16), (257, 16), (4, 16)], 5, 1e-2, 1e-3),
    ([(70000, 16)], 3, 1e-3, 1e-5),
    ([(5, 4), (1, 8), (1025, 32)], 4, 1e-3, 0.0),
])
def test_adam_dense_vs_oracle(shapes, steps, lr, wd):
    ps, ref, opt, ms, vs = run_table_adam(shapes, steps, lr, wd, seed=len(shapes) + steps)
    for p, r in zip(ps, ref):
        close(p, r, rtol=1e-5, atol_scale=1e-6, what="adam param")
    for i, p in enumerate(ps):
        close(opt.state[p]["exp_avg"], ms[i], rtol=1e-5, atol_scale=1e-7, what="adam m")
        close(opt.state[p]["exp_avg_sq"], vs[i], rtol=1e-5, atol_scale=1e-9, what="adam v")
    assert int(opt._t_step.item()) == steps
    assert abs(opt.state_dict()["state"][0]["step"].item() - steps) < 1e-6


def test_adam_dense_matches_torch_adam_on_device():
    """Same trajectory as stock torch.optim.Adam fed the same dense gradients (rows without data gradient move too)."""
    from torch_rechub_amd import ops
    from torch_rechub_amd.optim import TableAdam
    g = torch.Generator().manual_seed(9)
    w0 = torch.randn(5000, 16, generator=g) * 0.05
    a = torch.nn.Parameter(w0.clone().to(dev()))
    b = torch.nn.Parameter(w0.clone().to(dev()))
    oa = TableAdam([a], table_params=[a], lr=1e-3, weight_decay=1e-5)
    ob = torch.optim.Adam([b], lr=1e-3, weight_decay=1e-5)
    for t in range(6):
        gr = torch.zeros(5000, 16)
        rows = torch.randint(0, 5000, (300,), generator=g)
        gr[rows] = torch.randn(300, 16, generator=g)
        ops.grad_buffer(a).copy_(gr.to(dev()))
        a._rh_dirty = True
        b.grad = gr.to(dev())
        oa.step()
        ob.step()
    close(a, b.detach().cpu().numpy(), rtol=1e-5, atol_scale=1e-6, what="vs torch.optim.Adam")
    assert not torch.equal(a.detach().cpu(), w0)


def test_batch_gather_vs_oracle():
    from torch_rechub_amd.utils.data import DeviceDataLoader
    g = torch.Generator().manual_seed(4)
    N, F, ND, B = 1003, 26, 13, 128
    sparse = torch.randint(0, 1 << 40, (N, F), generator=g)
    dense = torch.rand(N, ND, generator=g)
    label = (torch.rand(N, generator=g) < 0.25).float()
    names = [f"C{i}" for i in range(F)]
    dn = [f"I{i}" for i in range(ND)]
    dl = DeviceDataLoader(sparse.to(dev()), names, dense.to(dev()), dn, label.to(dev()), B, shuffle=True)
    seen = []
    nb = 0
    for x, y in dl:
        nb += 1
        perm = dl.perm.cpu().numpy()
        b = y.shape[0]
        s_ref, d_ref, l_ref = O.batch_gather(perm, (nb - 1) * B, b, sparse.numpy(), dense.numpy(), label.numpy())
        assert np.array_equal(x.sparse.cpu().numpy(), s_ref)
        assert np.array_equal(x.dense.cpu().numpy(), d_ref)
        assert np.array_equal(y.cpu().numpy(), l_ref)
        assert np.array_equal(x["C3"].cpu().numpy(), s_ref[:, 3]) and np.array_equal(x["I12"].cpu().numpy(), d_ref[:, 12])
        seen.append(perm[(nb - 1) * B:(nb - 1) * B + b])
    assert nb == len(dl) == 8
    assert sorted(np.concatenate(seen).tolist()) == list(range(N))  # every row exactly once per epoch
    assert int(dl.pos.item()) == 0  # wrapped


def _lazy_vs_dense(lazy_k, small_rows, steps, shapes, seed, flush_every=None, nb=97):
    """Drive a dense TableAdam and a lazy one with identical lookup-style gradients; returns both after a flush."""
    from torch_rechub_amd import ops
    from torch_rechub_amd.optim import TableAdam
    g = torch.Generator().manual_seed(seed)
    init = [torch.randn(s, generator=g) * 0.1 for s in shapes]
    A = [torch.nn.Parameter(t.clone().to(dev())) for t in init]
    Bp = [torch.nn.Parameter(t.clone().to(dev())) for t in init]
    dense = TableAdam(A, table_params=A, lr=1e-2, weight_decay=1e-3)
    lazy = TableAdam(Bp, table_params=Bp, lr=1e-2, weight_decay=1e-3, lazy_k=lazy_k, lazy_small_rows=small_rows)
    for t in range(steps):
        idx_cols = []
        for i, s in enumerate(shapes):
            idx = torch.randint(0, s[0], (nb,), generator=g)
            if t % 5 == 3:
                idx[:] = idx[0]  # a step where the whole batch hits one row (duplicates -> one claim)
            rows_g = torch.randn(nb, s[1], generator=g)
            dense_g = torch.zeros(s)
            dense_g.index_add_(0, idx, rows_g)
            for P in (A[i], Bp[i]):
                ops.grad_buffer(P).copy_(dense_g.to(dev()))
                P._rh_dirty = True
            idx_cols.append(idx.to(dev()))
        key = tuple([c.data_ptr() for c in idx_cols] + [1] * len(shapes) + list(range(len(shapes))))
        idesc = ops.EmbedCall._icache.get(key, dev())
        # one field per table, all of one embed_dim
        ops._log_touch(Bp, [None] * len(shapes), idesc, 1, nb, len(shapes), shapes[0][1], idx_cols)
        lazy.step()  # consumes (and clears) its lookup log
        dense.step()
        if flush_every and (t + 1) % flush_every == 0:
            lazy.flush()
    lazy.flush()
    torch.cuda.synchronize()
    ops.check_errors()
    return A, Bp, dense, lazy


@pytest.mark.parametrize("lazy_k,small_rows,steps,flush_every", [(4, 16, 23, None), (16, 64, 40, 7), (2, 0, 5, None),
                                                                 (64, 16, 70, None)])
def test_adam_lazy_is_bit_identical_to_dense(lazy_k, small_rows, steps, flush_every):
    shapes = [(5, 16), (300, 16), (50, 16), (4099, 16)]
    A, Bp, dense, lazy = _lazy_vs_dense(lazy_k, small_rows, steps, shapes, seed=lazy_k + steps, flush_every=flush_every)
    for i, (a, b) in enumerate(zip(A, Bp)):
        assert torch.equal(a.detach(), b.detach()), f"table {i}: params differ"
        assert torch.equal(dense.state[a]["exp_avg"], lazy.state[b]["exp_avg"]), f"table {i}: exp_avg differs"
        assert torch.equal(dense.state[a]["exp_avg_sq"], lazy.state[b]["exp_avg_sq"]), f"table {i}: exp_avg_sq differs"
        from torch_rechub_amd import ops
        assert torch.all(ops.grad_buffer(b) == 0)  # every touched gradient row was re-zeroed
    for last in lazy._t_last:
        assert torch.all(last == steps)  # flush brought every row to the current step
    assert int(lazy._t_step.item()) == steps


def test_refresh_skips_padding_lookups_but_keeps_a_nonzero_padding_row_exact():
    """A post-padded history batch sends most of its lookups to padding_idx.  The pre-gather refresh skips those lookups
    (one claim word would take them all) and keeps the padding row itself current with ONE lane group per field and
    launch -- zero or not, dense Adam with coupled weight decay moves that row like every other (reference
    trainers/ctr_trainer.py:59-61).  Checked after every refresh against a dense twin: the padding row and the
    looked-up rows are bit-equal to what the dense optimizer holds at that step; the rows nobody looked up lag behind."""
    from torch_rechub_amd import ops
    from torch_rechub_amd.optim import TableAdam
    g = torch.Generator().manual_seed(17)
    V, D, B, L, pad = 5000, 16, 64, 12, 0
    w0 = torch.randn(V, D, generator=g) * 0.1  # padding row NOT zero
    A, Bp = torch.nn.Parameter(w0.clone().to(dev())), torch.nn.Parameter(w0.clone().to(dev()))
    dense = TableAdam([A], table_params=[A], lr=1e-2, weight_decay=1e-3)
    lazy = TableAdam([Bp], table_params=[Bp], lr=1e-2, weight_decay=1e-3, lazy_k=8, lazy_small_rows=8)
    lazy.overlap_sweep = False
    for step in range(20):
        idx = torch.randint(1, V, (B, L), generator=g)
        lens = torch.randint(1, L + 1, (B,), generator=g)
        idx[torch.arange(L)[None, :] >= lens[:, None]] = pad
        idx = idx.to(dev())
        flat = idx.view(-1)
        idesc = ops.EmbedCall._icache.get((flat.data_ptr(), 1, 0), dev())
        ops._pre_gather([Bp], [pad], idesc, 1, B * L, 1, D, training=True)  # what the sequence gather's forward does
        torch.cuda.synchronize()
        rows = torch.unique(flat)
        assert torch.equal(Bp.detach()[rows], A.detach()[rows]), f"step {step}: looked-up rows (padding row included) stale"
        assert torch.equal(Bp.detach()[pad], A.detach()[pad]) and bool(A.detach()[pad].abs().sum() > 0)
        gr = torch.zeros(V, D)
        live = flat.cpu()[flat.cpu() != pad]
        gr.index_add_(0, live, torch.randn(live.numel(), D, generator=g))
        for P in (A, Bp):
            ops.grad_buffer(P).copy_(gr.to(dev()))
            P._rh_dirty = True
        ops._log_touch([Bp], [pad], idesc, 1, B * L, 1, D, [flat])
        lazy.step()
        dense.step()
    assert not torch.equal(Bp.detach(), A.detach())  # lazy: rows outside the batches are behind until the flush
    lazy.flush()
    torch.cuda.synchronize()
    assert torch.equal(Bp.detach(), A.detach())


def test_adam_lazy_moves_heavily_looked_up_tables_to_dense_stepping():
    """Default placement (no explicit lazy_small_rows): at the first step a table whose rows are looked up often enough
    (rows <= 4 x lookups per step; 1500 lookups here) is stepped densely like the small tables, the others stay lazy; the
    result is the dense optimizer's either way, bit for bit."""
    shapes = [(300, 16), (6000, 16), (6001, 16), (40000, 16)]
    A, Bp, dense, lazy = _lazy_vs_dense(8, None, 19, shapes, seed=3, nb=1500)
    assert [lazy.table_k(p) for p in Bp] == [1, 1, 8, 8] and lazy.lazy_small_rows == 4096 and lazy.lazy_dense_ratio == 4.0
    for a, b in zip(A, Bp):
        assert torch.equal(a.detach(), b.detach())
        assert torch.equal(dense.state[a]["exp_avg_sq"], lazy.state[b]["exp_avg_sq"])
    # an explicit lazy_small_rows is the whole rule
    _, Bq, _, lazy2 = _lazy_vs_dense(8, 16, 3, shapes, seed=3, nb=1500)
    assert [lazy2.table_k(p) for p in Bq] == [8, 8, 8, 8]


def test_adam_lazy_rows_lag_at_most_k_steps():
    from torch_rechub_amd import ops
    A, Bp, dense, lazy = _lazy_vs_dense(8, 16, 30, [(3000, 16)], seed=1)
    # after the flush everything is current; run 5 more steps WITHOUT flush and check the lag bound
    for _ in range(5):
        lazy.step()
    torch.cuda.synchronize()
    lag = int(lazy._t_step.item()) - lazy._t_last[0]
    assert int(lag.min()) >= 0 and int(lag.max()) < 8


@pytest.mark.parametrize("B,C,p", [(4096, 256, 0.0), (4096, 128, 0.0), (37, 32, 0.0), (1000, 200, 0.0), (513, 16, 0.0),
                                   (8192, 64, 0.0), (2, 8, 0.0), (20000, 64, 0.0), (1000, 30, 0.0)])
def test_bn_relu_dropout_vs_torch_modules(B, C, p):
    """Fused epilogue == nn.BatchNorm1d -> ReLU -> Dropout(p=0) (outputs, running stats, all gradients) and eval mode."""
    from torch_rechub_amd import ops
    torch.manual_seed(B + C)
    h0 = (torch.randn(B, C) * 2 + torch.randn(C) * 3).to(dev())
    bn_ref = torch.nn.BatchNorm1d(C).to(dev())
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5)
        bn_ref.bias.normal_(0, 0.5)
    bn_mine = torch.nn.BatchNorm1d(C).to(dev())
    bn_mine.load_state_dict(bn_ref.state_dict())
    gy = torch.randn(B, C, device=dev())
    ha = h0.clone().requires_grad_(True)
    ya = torch.relu(bn_ref(ha))
    ya.backward(gy)
    hb = h0.clone().requires_grad_(True)
    yb = ops.bn_relu_dropout(hb, bn_mine, p)
    yb.backward(gy)
    close(yb, ya.detach().cpu().numpy(), rtol=2e-5, atol_scale=2e-6, what="bn out")
    close(hb.grad, ha.grad.cpu().numpy(), rtol=1e-4, atol_scale=1e-5, what="bn dx")
    close(bn_mine.weight.grad, bn_ref.weight.grad.cpu().numpy(), rtol=1e-4, atol_scale=1e-5, what="dgamma")
    close(bn_mine.bias.grad, bn_ref.bias.grad.cpu().numpy(), rtol=1e-4, atol_scale=1e-5, what="dbeta")
    close(bn_mine.running_mean, bn_ref.running_mean.cpu().numpy(), what="running_mean")
    close(bn_mine.running_var, bn_ref.running_var.cpu().numpy(), rtol=2e-5, what="running_var")
    assert int(bn_mine.num_batches_tracked) == int(bn_ref.num_batches_tracked) == 1
    bn_ref.eval()
    bn_mine.eval()
    with torch.no_grad():
        close(ops.bn_relu_dropout(h0, bn_mine, 0.3), torch.relu(bn_ref(h0)).cpu().numpy(), rtol=2e-5, atol_scale=2e-6,
              what="eval")


@pytest.mark.parametrize("B", [4096, 10000])  # one-launch column-owner path / three-launch path
def test_fused_dropout_statistics_and_backward_mask(B):
    from torch_rechub_amd import ops
    C, p = 256, 0.2
    bn = torch.nn.BatchNorm1d(C).to(dev())
    h = (torch.randn(B, C, device=dev()) + 1.0).requires_grad_(True)
    y = ops.bn_relu_dropout(h, bn, p)
    y2 = ops.bn_relu_dropout(h.detach(), bn, p)
    pos = torch.relu(bn(h.detach())) > 0  # (running stats moved, batch stats are the same) positions alive after relu
    kept = (y.detach() != 0) & pos
    frac = kept.sum().item() / pos.sum().item()
    assert abs(frac - (1 - p)) < 0.01  # keep probability
    assert (kept != ((y2 != 0) & pos)).any()  # a new mask per call (device-side counter)
    ref = (torch.relu(bn(h.detach())) / (1 - p)).detach()
    np.testing.assert_allclose(y.detach()[kept].cpu().numpy(), ref[kept].cpu().numpy(), rtol=1e-4, atol=1e-5)
    y.backward(torch.ones_like(y))
    # gradient flows only through kept positions: dropped & relu-dead columns contribute nothing to dbeta
    dbeta = (kept.float() / (1 - p)).sum(0)
    np.testing.assert_allclose(bn.bias.grad.cpu().numpy(), dbeta.cpu().numpy(), rtol=1e-4, atol=1e-3)


def _dice_ref(x, alpha, eps=1e-3):
    """basic/activation.py:15-25 in float64 torch (autograd gives the reference gradients)."""
    avg = x.mean(dim=1, keepdim=True)
    var = (torch.pow(x - avg, 2) + eps).sum(dim=1, keepdim=True)
    ps = torch.sigmoid((x - avg) / torch.sqrt(var))
    return ps * x + (1 - ps) * alpha * x


@pytest.mark.parametrize("N,C", [(409600 // 16, 256), (1000, 128), (77, 36), (5, 1), (300, 200), (64, 1100), (3, 2048)])
def test_dice_kernel_vs_reference_formula(N, C):
    from torch_rechub_amd import ops
    g = torch.Generator().manual_seed(N + C)
    x = torch.randn(N, C, generator=g) * 1.5 + 0.3
    alpha = torch.randn(1, generator=g)
    gy = torch.randn(N, C, generator=g)
    xr = x.double().requires_grad_(True)
    ar = alpha.double().requires_grad_(True)
    yr = _dice_ref(xr, ar)
    yr.backward(gy.double())
    close(O.dice_forward(x.numpy().astype(F64), float(alpha)), yr.detach().numpy(), rtol=1e-12, what="oracle==formula")
    xd = x.to(dev()).requires_grad_(True)
    ad = alpha.to(dev()).requires_grad_(True)
    y = ops.dice(xd, ad, 1e-3)
    y.backward(gy.to(dev()))
    close(y, yr.detach().numpy(), rtol=2e-5, atol_scale=2e-6, what="dice out")
    close(xd.grad, xr.grad.numpy(), rtol=1e-4, atol_scale=1e-5, what="dice gx")
    close(ad.grad, ar.grad.numpy(), rtol=1e-4, atol_scale=1e-5, what="dice galpha")


@pytest.mark.parametrize("B,L,D,strided", [(300, 100, 16, True), (4096, 50, 16, False), (33, 7, 64, True), (5, 1, 8, False),
                                           (64, 9, 4, True), (10, 33, 128, False)])
def test_din_attention_ends_vs_reference_formula(B, L, D, strided):
    """cat[t, h, t-h, t*h] and sum_l w_l h_l (models/ranking/din.py:80-81, 92), forward and backward."""
    from torch_rechub_amd import ops
    g = torch.Generator().manual_seed(B + L + D)
    hist_all = torch.randn(B, 2, L, D, generator=g)
    tgt_all = torch.randn(B, 2, D, generator=g)
    hist, tgt = (hist_all[:, 1], tgt_all[:, 1]) if strided else (hist_all[:, 0].contiguous(), tgt_all[:, 0].contiguous())
    w = torch.randn(B, L, generator=g)
    # reference (float64, autograd)
    hr, tr, wr = hist.double().requires_grad_(True), tgt.double().requires_grad_(True), w.double().requires_grad_(True)
    te = tr.unsqueeze(1).expand(-1, L, -1)
    att_in = torch.cat([te, hr, te - hr, te * hr], dim=-1).view(-1, 4 * D)
    pooled = (wr.unsqueeze(-1) * hr).sum(dim=1)
    g1 = torch.randn(B * L, 4 * D, generator=g)
    g2 = torch.randn(B, D, generator=g)
    torch.autograd.backward([att_in, pooled], [g1.double(), g2.double()])
    # HIP
    ha = hist_all.to(dev())
    ta = tgt_all.to(dev())
    hd = (ha[:, 1] if strided else ha[:, 0].contiguous()).detach().requires_grad_(True)
    td = (ta[:, 1] if strided else ta[:, 0].contiguous()).detach().requires_grad_(True)
    wd = w.to(dev()).requires_grad_(True)
    a = ops.din_att_input(hd, td)
    p = ops.din_att_pool(wd, hd)
    assert np.array_equal(a.detach().cpu().numpy()[:, :2 * D], att_in.detach().float().numpy()[:, :2 * D])  # copies
    close(a, att_in.detach().numpy(), rtol=1e-6, what="att_input")
    close(p, pooled.detach().numpy(), rtol=2e-5, atol_scale=2e-6, what="pooled")
    torch.autograd.backward([a, p], [g1.to(dev()), g2.to(dev())])
    close(hd.grad, hr.grad.numpy(), rtol=2e-5, atol_scale=2e-6, what="g_hist")
    close(td.grad, tr.grad.numpy(), rtol=2e-5, atol_scale=1e-5, what="g_tgt")
    close(wd.grad, wr.grad.numpy(), rtol=2e-5, atol_scale=2e-6, what="g_w")


@pytest.mark.parametrize("B,d,E", [(4096, 429, 4), (257, 45, 3), (5, 1, 1), (64, 1100, 2)])
def test_cross_mix_and_v2_epilogues_vs_reference_formula(B, d, E):
    """Epilogues of basic/layers.py:442-443 and :491-503 against the same expressions in float64 torch (autograd)."""
    from torch_rechub_amd import ops
    g = torch.Generator().manual_seed(B + d + E)
    x0, xl = torch.randn(B, d, generator=g), torch.randn(B, d, generator=g)
    uv = torch.randn(E, B, d, generator=g)
    gate = torch.softmax(torch.randn(B, E, generator=g), dim=1)
    bias = torch.randn(d, generator=g) * 0.3
    gy = torch.randn(B, d, generator=g)
    ref_in = [t.double().requires_grad_(True) for t in (x0, xl, uv, gate, bias)]
    r0, rl, ruv, rg, rb = ref_in
    ref = ((r0.unsqueeze(0) * (ruv + rb.view(1, 1, d))) * rg.t().unsqueeze(2)).sum(0) + rl
    ref.backward(gy.double())
    hip_in = [t.to(dev()).requires_grad_(True) for t in (x0, xl, uv, gate, bias)]
    out = ops.cross_mix_epilogue(*hip_in)
    out.backward(gy.to(dev()))
    close(out, ref.detach().numpy(), rtol=2e-5, atol_scale=2e-6, what="mix out")
    for name, h, r in zip(("g_x0", "g_xl", "g_uv", "g_gate", "g_bias"), hip_in, ref_in):
        close(h.grad, r.grad.numpy(), rtol=1e-4, atol_scale=1e-5, what="mix " + name)
    # CrossNetV2 epilogue: x0 * y + b + x
    y = torch.randn(B, d, generator=g)
    ref2_in = [t.double().requires_grad_(True) for t in (x0, y, bias, xl)]
    ref2 = ref2_in[0] * ref2_in[1] + ref2_in[2] + ref2_in[3]
    ref2.backward(gy.double())
    hip2_in = [t.to(dev()).requires_grad_(True) for t in (x0, y, bias, xl)]
    out2 = ops.cross_v2_epilogue(*hip2_in)
    out2.backward(gy.to(dev()))
    close(out2, ref2.detach().numpy(), rtol=1e-6, atol_scale=1e-6, what="v2 out")
    for name, h, r in zip(("g_x0", "g_y", "g_b", "g_x"), hip2_in, ref2_in):
        close(h.grad, r.grad.numpy(), rtol=2e-5, atol_scale=2e-6, what="v2 " + name)


# ----------------------------------------------------------------------------------------------------------
# csrc/linear.hip: split-batch MFMA weight gradient, fused output head, BCE
@pytest.mark.parametrize("B,N,K,pad", [(4096, 256, 429, 0), (4096, 128, 256, 0), (100, 1, 7, 0), (37, 70, 130, 3),
                                       (1, 5, 3, 0), (8191, 64, 64, 0), (50000, 36, 64, 0),
                                       # long reductions of 2 .. 8 tiles: ONE workgroup per row split computes the whole slab
                                       # (linear_wgrad_rows_kernel, round 6); 20 tiles: back to one workgroup per tile
                                       (40000, 128, 256, 0), (33001, 130, 70, 3), (70001, 256, 64, 0), (40000, 200, 300, 0)])
def test_linear_wgrad_vs_float64(B, N, K, pad):
    """dW = g^T x, db = colsum(g): f32 MFMA == k-ordered fmaf chain, so only the summation order differs from float64."""
    from torch_rechub_amd import ops
    gen = torch.Generator().manual_seed(B + N + K)
    g = torch.randn(B, N, generator=gen)
    xfull = torch.randn(B, K + pad, generator=gen)
    gd, xd = g.to(dev()), xfull.to(dev())[:, :K]  # row stride K + pad
    dW, db = ops.linear_wgrad(gd, xd)
    ref_w = g.double().t() @ xfull[:, :K].double()
    ref_b = g.double().sum(0)
    scale = float(np.sqrt(B))
    np.testing.assert_allclose(dW.cpu().numpy(), ref_w.numpy(), rtol=1e-5, atol=3e-6 * scale, err_msg="dW")
    np.testing.assert_allclose(db.cpu().numpy(), ref_b.numpy(), rtol=1e-5, atol=3e-6 * scale, err_msg="db")
    # deterministic (fixed split order)
    dW2, db2 = ops.linear_wgrad(gd, xd)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)
    dW3, none = ops.linear_wgrad(gd, xd, want_bias=False)
    assert none is None and torch.equal(dW, dW3)


def test_linear_function_matches_torch_autograd():
    from torch_rechub_amd import ops
    torch.manual_seed(3)
    lin = torch.nn.Linear(429, 256).to(dev())
    x0 = torch.randn(4096, 429, device=dev())
    gy = torch.randn(4096, 256, device=dev())
    xa = x0.clone().requires_grad_(True)
    lin(xa).backward(gy)
    want = (xa.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    lin.zero_grad()
    xb = x0.clone().requires_grad_(True)
    out = ops.linear(xb, lin.weight, lin.bias)
    close(out, lin(x0).detach().cpu().numpy(), rtol=1e-4, atol_scale=2e-6, what="y")  # own MFMA GEMM: other k order
    out.backward(gy)
    close(xb.grad, want[0].cpu().numpy(), rtol=1e-4, atol_scale=2e-6, what="g_x")
    close(lin.weight.grad, want[1].cpu().numpy(), rtol=1e-4, atol_scale=2e-6, what="dW")
    close(lin.bias.grad, want[2].cpu().numpy(), rtol=1e-4, atol_scale=2e-6, what="db")


@pytest.mark.parametrize("B,K,bias,nextra", [(4096, 128, True, 2), (5, 4, True, 0), (1, 1024, False, 1), (777, 36, True, 1),
                                             (70000, 64, True, 2), (4096, 557, True, 0), (33, 3, True, 1), (100, 1, False, 0),
                                             (512, 1023, True, 2)])
def test_head_sigmoid_vs_autograd(B, K, bias, nextra):
    from torch_rechub_amd import ops
    gen = torch.Generator().manual_seed(B + K)
    h0 = torch.randn(B, K, generator=gen).to(dev())
    lin = torch.nn.Linear(K, 1, bias=bias).to(dev())
    ex0 = [torch.randn(B, 1, generator=gen).to(dev()) for _ in range(nextra)]
    gy = torch.randn(B, generator=gen).to(dev())

    def run(fused):
        lin.zero_grad()
        h = h0.clone().requires_grad_(True)
        ex = [e.clone().requires_grad_(True) for e in ex0]
        if fused:
            y = ops.head_sigmoid(h, lin.weight, lin.bias, *ex)
        else:
            z = lin(h.double()) if False else lin(h)
            for e in ex:
                z = z + e
            y = torch.sigmoid(z.squeeze(1))
        y.backward(gy)
        return [y.detach(), h.grad, lin.weight.grad.clone()] + ([lin.bias.grad.clone()] if bias else []) + [e.grad for e in ex]

    got, want = run(True), run(False)
    names = ["y", "g_h", "g_w"] + (["g_b"] if bias else []) + [f"g_e{i}" for i in range(nextra)]
    for n, a, b in zip(names, got, want):
        assert a.shape == b.shape, n
        close(a, b.cpu().numpy(), rtol=2e-5, atol_scale=2e-6 * (np.sqrt(B) if n in ("g_w", "g_b") else 1.0), what=n)


@pytest.mark.parametrize("B", [1, 100, 4096, 100001])
def test_bce_mean_vs_torch(B):
    from torch_rechub_amd import ops
    gen = torch.Generator().manual_seed(B)
    y0 = torch.rand(B, generator=gen)
    if B >= 100:
        y0[:4] = torch.tensor([0.0, 1.0, 1e-30, 1.0 - 1e-7])  # the -100 clamp and the 1e-12 guard of the backward
    t = (torch.rand(B, generator=gen) < 0.3).float().to(dev())
    ya = y0.to(dev()).requires_grad_(True)
    la = torch.nn.BCELoss()(ya, t)
    la.backward()
    yb = y0.to(dev()).requires_grad_(True)
    assert ops.bce_ok(torch.nn.BCELoss(), yb, t)
    lb = ops.bce_mean(yb, t)
    (lb * 1.0).backward()
    assert lb.shape == la.shape
    close(lb, la.detach().cpu().numpy(), rtol=2e-6, what="bce")
    close(yb.grad, ya.grad.cpu().numpy(), rtol=1e-5, atol_scale=1e-7, what="bce grad")
    assert not ops.bce_ok(torch.nn.BCELoss(reduction="sum"), yb, t)


# ----------------------------------------------------------------------------------------------------------
# csrc/gemm.hip: f32-MFMA tile GEMMs of the MLP (forward + BN statistics epilogue, input gradient)
@pytest.mark.parametrize("M,N,K", [(4096, 256, 429), (4096, 128, 256), (4096, 1, 128), (100, 70, 33), (33, 5, 7),
                                   (1, 64, 64), (2048, 429, 256), (16384, 36, 64), (4000, 200, 130)])
def test_linear_forward_and_input_gradient_vs_float64(M, N, K, monkeypatch):
    from torch_rechub_amd import _lib, ops
    monkeypatch.setattr(ops, "_GEMM_MAX_M", 16384)  # the opt-in f32-MFMA tile GEMMs (RECHUB_OWN_GEMM=1)
    gen = torch.Generator().manual_seed(M + N + K)
    x, w, b = torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen) * 0.1, torch.randn(N, generator=gen)
    g = torch.randn(M, N, generator=gen)
    xd, wd, bd, gd = x.to(dev()).requires_grad_(True), w.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True), g.to(dev())
    assert ops._own_gemm(M, N, K)
    bn_probe = torch.nn.BatchNorm1d(N).to(dev())
    y, (stats, ctr) = ops.linear_stats(xd, wd, bd, bn_probe)
    assert int(bn_probe.num_batches_tracked) == 1 and ctr.shape == (1,)
    ref = x.double() @ w.double().t() + b.double()
    tol = 2e-6 * np.sqrt(K)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.numpy(), rtol=1e-5, atol=tol * float(ref.abs().max()), err_msg="y")
    # BN statistics epilogue: per 32-row slab (sum, M2 about the slab mean) of y
    R = _lib.call("rh_gemm_stats_rows", M, N)
    slabs = -(-M // R)
    assert R in (32, 64) and stats.shape == (slabs, 2, N)
    yd = y.detach().double().cpu()
    for k in (0, slabs // 2, slabs - 1):
        blk = yd[R * k:R * (k + 1)]
        np.testing.assert_allclose(stats[k, 0].cpu().numpy(), blk.sum(0).numpy(), rtol=1e-5, atol=1e-4, err_msg="slab sum")
        np.testing.assert_allclose(stats[k, 1].cpu().numpy(), ((blk - blk.mean(0))**2).sum(0).numpy(), rtol=1e-4, atol=1e-4,
                                   err_msg="slab M2")
    y.backward(gd)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), (g.double() @ w.double()).numpy(), rtol=1e-5,
                               atol=2e-6 * np.sqrt(N) * float((g.double() @ w.double()).abs().max()), err_msg="g_x")
    np.testing.assert_allclose(wd.grad.cpu().numpy(), (g.double().t() @ x.double()).numpy(), rtol=1e-5, atol=3e-6 * np.sqrt(M) * 4,
                               err_msg="g_w")
    np.testing.assert_allclose(bd.grad.cpu().numpy(), g.double().sum(0).numpy(), rtol=1e-5, atol=3e-6 * np.sqrt(M) * 4)
    # plain linear (no statistics) gives the same output bits
    assert torch.equal(ops.linear(xd.detach(), wd.detach(), bd.detach()), y.detach())


@pytest.mark.parametrize("B,C", [(4096, 256), (4000, 128), (100, 36), (8192, 64)])
def test_bn_from_gemm_statistics_matches_torch_modules(B, C, monkeypatch):
    """Linear -> BatchNorm1d -> ReLU with the statistics taken from the GEMM epilogue (one BN launch) == torch modules,
    including a column whose mean is ~2000x its spread (the slab-wise M2 does not cancel)."""
    from torch_rechub_amd import ops
    monkeypatch.setattr(ops, "_GEMM_MAX_M", 16384)
    torch.manual_seed(B + C)
    K = 48
    lin = torch.nn.Linear(K, C).to(dev())
    with torch.no_grad():
        lin.bias[0] = 10.0  # mean ~ 2000 x spread: sum-of-squares statistics would lose every digit of the variance
        lin.weight[0] *= 0.01
    bn_ref, bn_mine = torch.nn.BatchNorm1d(C).to(dev()), torch.nn.BatchNorm1d(C).to(dev())
    x = torch.randn(B, K, device=dev())
    gy = torch.randn(B, C, device=dev())
    ya = torch.relu(bn_ref(lin(x)))
    ya.backward(gy)
    want = [ya.detach(), lin.weight.grad.clone(), bn_ref.weight.grad.clone(), bn_ref.bias.grad.clone()]
    lin.zero_grad()
    h, stats = ops.linear_stats(x, lin.weight, lin.bias, bn_mine)
    assert stats is not None
    ctr0 = int(stats[1])
    yb = ops.bn_relu_dropout(h, bn_mine, 0.0, stats=stats)
    yb.backward(gy)
    torch.cuda.synchronize()
    assert int(ops._dropout_rng(dev())[1]) == ctr0 + 1  # the GEMM drew this call's dropout counter, once
    # column 0 sits at 10 +- 0.006: its fp32 inputs carry ~1e-4 of a standard deviation of rounding noise themselves
    close(yb[:, 1:], want[0][:, 1:].cpu().numpy(), rtol=1e-4, atol_scale=2e-6, what="bn(relu(linear)) out")
    # ... so it is judged against float64 batch-norm of the very same fp32 h (gamma = 1, beta = 0 at init)
    h0 = h[:, 0].detach().double()
    ref0 = torch.relu((h0 - h0.mean()) / torch.sqrt(h0.var(unbiased=False) + bn_mine.eps))
    close(yb[:, 0], ref0.cpu().numpy(), rtol=1e-3, atol_scale=1e-3, what="ill-conditioned column vs float64")
    close(lin.weight.grad[1:], want[1][1:].cpu().numpy(), rtol=1e-3, atol_scale=2e-5, what="dW through BN")
    close(bn_mine.weight.grad[1:], want[2][1:].cpu().numpy(), rtol=1e-4, atol_scale=1e-5, what="dgamma")
    close(bn_mine.bias.grad[1:], want[3][1:].cpu().numpy(), rtol=1e-4, atol_scale=1e-5, what="dbeta")
    close(bn_mine.running_mean, bn_ref.running_mean.cpu().numpy(), rtol=1e-5, what="running_mean")
    close(bn_mine.running_var[1:], bn_ref.running_var[1:].cpu().numpy(), rtol=1e-4, atol_scale=1e-6, what="running_var")
    var0 = 0.9 + 0.1 * float(h0.var(unbiased=True))
    close(bn_mine.running_var[:1], np.array([var0]), rtol=1e-3, what="running_var, ill-conditioned column vs float64")
    assert int(bn_mine.num_batches_tracked) == 1


def test_empty_batch_goes_through_every_op():
    """B = 0 (the tail of a dataset that divides evenly, an empty shard): every op returns an empty tensor of the right
    shape and its backward leaves zero gradients, as the reference's eager ops do."""
    from torch_rechub_amd import ops
    tables = [torch.nn.Parameter(torch.randn(v, 16, device=dev())) for v in (7, 300)]
    idx = torch.zeros((0, 2), dtype=torch.int64, device=dev())
    dn = torch.zeros((0, 3), device=dev())
    lr_w = torch.nn.Parameter(torch.randn(1, 32, device=dev()))
    lr_b = torch.nn.Parameter(torch.zeros(1, device=dev()))
    call = ops.EmbedCall(tables, [None, None], [idx[:, 0], idx[:, 1]], [dn[:, j] for j in range(3)], want_fm=True, want_lr=True)
    out, fm, lr = ops.fused_embedding(call, lr_w, lr_b)
    assert out.shape == (0, 35) and fm.shape == (0, 1) and lr.shape == (0, 1)
    (out.sum() + fm.sum() + lr.sum()).backward()
    assert all(float(t.grad.abs().sum()) == 0.0 for t in tables) and float(lr_w.grad.abs().sum()) == 0.0
    x = torch.zeros((0, 5, 16), device=dev(), requires_grad=True)
    assert ops.fm(x, True).shape == (0, 1)
    seq = ops.seq_pool(tables[1], torch.zeros((0, 4), dtype=torch.int64, device=dev()), "mean", None)
    assert seq.shape == (0, 16)
    xc = torch.zeros((0, 429), device=dev(), requires_grad=True)
    W = torch.randn(3, 429, device=dev(), requires_grad=True)
    Bv = torch.zeros(3, 429, device=dev(), requires_grad=True)
    yc = ops.cross_network(xc, W, Bv)
    assert yc.shape == (0, 429)
    yc.sum().backward()
    assert float(W.grad.abs().sum()) == 0.0
    lin = torch.nn.Linear(8, 4).to(dev())
    h = ops.linear(torch.zeros((0, 8), device=dev(), requires_grad=True), lin.weight, lin.bias)
    assert h.shape == (0, 4)
    h.sum().backward()
    assert float(lin.weight.grad.abs().sum()) == 0.0 and float(lin.bias.grad.abs().sum()) == 0.0
    ops.check_errors()


@pytest.mark.parametrize("B,C", [(4096, 64), (20000, 36), (300, 256), (409600, 128)])
def test_batchnorm_only_mode_vs_torch(B, C):
    """relu=False: the kernels do BatchNorm1d alone (the layers in front of Dice / PReLU), forward, backward, running stats."""
    from torch_rechub_amd import ops
    torch.manual_seed(B + C)
    h0 = (torch.randn(B, C) * 2 + torch.randn(C)).to(dev())
    bn_ref, bn_mine = torch.nn.BatchNorm1d(C).to(dev()), torch.nn.BatchNorm1d(C).to(dev())
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5)
        bn_ref.bias.normal_()
    bn_mine.load_state_dict(bn_ref.state_dict())
    gy = torch.randn(B, C, device=dev())
    ha = h0.clone().requires_grad_(True)
    bn_ref(ha).backward(gy)
    hb = h0.clone().requires_grad_(True)
    yb = ops.bn_relu_dropout(hb, bn_mine, 0.0, relu=False)
    yb.backward(gy)
    with torch.no_grad():
        want = torch.nn.functional.batch_norm(h0, None, None, bn_ref.weight, bn_ref.bias, True)
    close(yb, want.cpu().numpy(), rtol=2e-5, atol_scale=2e-6, what="bn out")
    assert bool((yb < 0).any())  # no rectification
    close(hb.grad, ha.grad.cpu().numpy(), rtol=1e-4, atol_scale=1e-5, what="dx")
    close(bn_mine.weight.grad, bn_ref.weight.grad.cpu().numpy(), rtol=1e-4, atol_scale=1e-5, what="dgamma")
    close(bn_mine.bias.grad, bn_ref.bias.grad.cpu().numpy(), rtol=1e-4, atol_scale=1e-5, what="dbeta")
    close(bn_mine.running_var, bn_ref.running_var.cpu().numpy(), rtol=2e-5, what="running_var")
    bn_mine.eval(), bn_ref.eval()
    with torch.no_grad():
        close(ops.bn_relu_dropout(h0, bn_mine, 0.0, relu=False), bn_ref(h0).cpu().numpy(), rtol=2e-5, atol_scale=2e-6, what="eval")


def test_dice_mlp_over_many_rows_takes_the_tile_gemm_with_the_statistics_epilogue(monkeypatch):
    """Round 6: the forward of a Linear over (B L) rows in front of BatchNorm1d -> Dice (DIN's attention MLP,
    models/ranking/din.py:75-85 over basic/layers.py:279-288) runs on the tile GEMM, whose epilogue hands the BatchNorm its
    per-slab (sum, M2) -- no statistics pass over the (B L, C) tensor (ops.linear_chunk_stats -> bn_dice / bn_dice_head with
    chunk_stats).  Against the float64 modules and against its own twin (library GEMM + statistics pass, RECHUB_AB=tallgemm=0):
    output, input gradient, every parameter gradient, running statistics, num_batches_tracked counted once."""
    from torch_rechub_amd import _lib, ops
    from torch_rechub_amd.basic.activation import Dice
    from torch_rechub_amd.basic.layers import MLP
    torch.manual_seed(5)
    N, K = 70016, 48
    x0 = (torch.randn(N, K) * 0.7).to(dev())
    gy = torch.randn(N, 1).to(dev())
    ref = MLP(K, output_layer=True, dims=[64, 32], activation="dice").to(dev())
    with torch.no_grad():
        for m in ref.mlp:
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_()
    seen = {"stats": 0, "plain": 0, "from_partial": 0}
    real_call = _lib.call

    def spy(name, *args):
        if name == "rh_linear_fwd":
            seen["stats" if args[10].value else "plain"] += 1  # (argument 10: the statistics array)
        elif name == "rh_bn_stats_from_partial":
            seen["from_partial"] += 1
        return real_call(name, *args)

    monkeypatch.setattr(_lib, "call", spy)
    outs = []
    for tall in (True, False):
        monkeypatch.setattr(ops, "TALL_GEMM_FWD", tall)
        m = MLP(K, output_layer=True, dims=[64, 32], activation="dice").to(dev())
        m.load_state_dict(ref.state_dict())
        m.train()
        for k in seen:
            seen[k] = 0
        x = x0.clone().requires_grad_(True)
        y = m(x)
        y.backward(gy)
        torch.cuda.synchronize()
        assert (seen["stats"] == 2 and seen["from_partial"] == 2) if tall else (seen["stats"] == 0 and seen["from_partial"] == 0), seen
        bns = [q for q in m.mlp if isinstance(q, torch.nn.BatchNorm1d)]
        assert all(int(b.num_batches_tracked) == 1 for b in bns)
        outs.append((y.detach(), x.grad.detach(), {n: p.grad.detach() for n, p in m.named_parameters()},
                     [b.running_var.detach().clone() for b in bns]))
    # float64 modules (torch BatchNorm1d in float64, the reference's Dice formula)
    r64 = MLP(K, output_layer=True, dims=[64, 32], activation="dice").double().to(dev())
    r64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in ref.state_dict().items()})
    r64.train()
    for q in r64.mlp:
        assert not isinstance(q, Dice) or q.alpha.dtype == torch.float64
    x64 = x0.double().requires_grad_(True)
    h = x64
    for q in r64.mlp:  # (module by module: the float64 input takes torch's own kernels, not the HIP path)
        h = _dice_ref(h, q.alpha) if isinstance(q, Dice) else q(h)
    h.backward(gy.double())
    for what, (y, gx, gp, rv) in zip(("tile GEMM + epilogue statistics", "library GEMM + statistics pass"), outs):
        close(y, h.detach().cpu().numpy(), rtol=3e-4, atol_scale=1e-5, what=f"{what}: out")
        close(gx, x64.grad.cpu().numpy(), rtol=2e-3, atol_scale=5e-5, what=f"{what}: dx")
        for n, p in r64.named_parameters():
            if n in ("mlp.0.bias", "mlp.4.bias"):
                continue  # (the bias of a Linear in front of a BatchNorm has a zero gradient: what is left is cancellation noise)
            close(gp[n], p.grad.cpu().numpy(), rtol=2e-3, atol_scale=5e-5, what=f"{what}: d {n}")
    for a, b in zip(outs[0][3], outs[1][3]):
        close(a, b.cpu().numpy(), rtol=2e-5, what="running_var: epilogue statistics vs statistics pass")


@pytest.mark.parametrize("N,C", [(25600, 256), (1000, 128), (77, 36), (300, 200), (409600, 64)])
def test_bn_dice_folded_vs_modules(N, C):
    """Linear's BatchNorm1d -> Dice with the normalisation folded into the Dice passes == the two modules one after the
    other (float64 Dice formula on torch's BatchNorm output): outputs, all gradients, running statistics, eval mode."""
    from torch_rechub_amd import ops
    torch.manual_seed(N + C)
    h0 = (torch.randn(N, C) * 1.5 + torch.randn(C)).to(dev())
    bn_ref, bn_mine = torch.nn.BatchNorm1d(C).to(dev()), torch.nn.BatchNorm1d(C).to(dev())
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5)
        bn_ref.bias.normal_()
    bn_mine.load_state_dict(bn_ref.state_dict())
    alpha0 = torch.randn(1)
    gy = torch.randn(N, C, device=dev())
    ha = h0.clone().requires_grad_(True)
    al_a = alpha0.double().to(dev()).requires_grad_(True)
    ya = _dice_ref(bn_ref(ha).double(), al_a)
    ya.backward(gy.double())
    hb = h0.clone().requires_grad_(True)
    al_b = alpha0.to(dev()).requires_grad_(True)
    yb = ops.bn_dice(hb, bn_mine, al_b, 1e-3)
    yb.backward(gy)
    close(yb, ya.detach().cpu().numpy(), rtol=2e-4, atol_scale=5e-6, what="out")
    close(hb.grad, ha.grad.cpu().numpy(), rtol=1e-3, atol_scale=2e-5, what="dh")
    close(bn_mine.weight.grad, bn_ref.weight.grad.cpu().numpy(), rtol=1e-3, atol_scale=2e-5, what="dgamma")
    close(bn_mine.bias.grad, bn_ref.bias.grad.cpu().numpy(), rtol=1e-3, atol_scale=2e-5, what="dbeta")
    close(al_b.grad, al_a.grad.cpu().numpy(), rtol=1e-3, atol_scale=1e-4, what="dalpha")
    close(bn_mine.running_mean, bn_ref.running_mean.cpu().numpy(), rtol=1e-5, what="running_mean")
    close(bn_mine.running_var, bn_ref.running_var.cpu().numpy(), rtol=2e-5, what="running_var")
    assert int(bn_mine.num_batches_tracked) == 1
    bn_ref.eval(), bn_mine.eval()
    with torch.no_grad():
        close(ops.bn_dice(h0, bn_mine, al_b.detach(), 1e-3), _dice_ref(bn_ref(h0).double(), al_a.detach()).cpu().numpy(),
              rtol=2e-4, atol_scale=5e-6, what="eval")


@pytest.mark.parametrize("N,C,bias", [(25600, 128, True), (1000, 64, True), (77, 36, False), (409600, 128, True), (300, 256, True)])
def test_bn_dice_head_vs_modules(N, C, bias):
    """BatchNorm1d -> Dice -> Linear(C, 1) (the tail of the ActivationUnit's MLP) with the Dice output never written ==
    the three modules one after the other (float64 Dice and dot product on torch's BatchNorm output): output, every
    gradient (h, gamma, beta, alpha, head weight, head bias), running statistics, eval mode."""
    from torch_rechub_amd import ops
    torch.manual_seed(N + C)
    h0 = (torch.randn(N, C) * 1.5 + torch.randn(C)).to(dev())
    bn_ref, bn_mine = torch.nn.BatchNorm1d(C).to(dev()), torch.nn.BatchNorm1d(C).to(dev())
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5)
        bn_ref.bias.normal_()
    bn_mine.load_state_dict(bn_ref.state_dict())
    lin = torch.nn.Linear(C, 1, bias=bias).to(dev())
    w64 = lin.weight.detach().double().clone().requires_grad_(True)
    b64 = lin.bias.detach().double().clone().requires_grad_(True) if bias else None
    alpha0 = torch.randn(1)
    gy = torch.randn(N, 1, device=dev())
    ha = h0.clone().requires_grad_(True)
    al_a = alpha0.double().to(dev()).requires_grad_(True)
    ya = _dice_ref(bn_ref(ha).double(), al_a) @ w64.t()
    if bias:
        ya = ya + b64
    ya.backward(gy.double())
    hb = h0.clone().requires_grad_(True)
    al_b = alpha0.to(dev()).requires_grad_(True)
    yb = ops.bn_dice_head(hb, bn_mine, al_b, 1e-3, lin)
    assert yb.shape == (N, 1)
    yb.backward(gy)
    close(yb, ya.detach().cpu().numpy(), rtol=2e-4, atol_scale=5e-6, what="out")
    close(hb.grad, ha.grad.cpu().numpy(), rtol=1e-3, atol_scale=2e-5, what="dh")
    close(bn_mine.weight.grad, bn_ref.weight.grad.cpu().numpy(), rtol=1e-3, atol_scale=2e-5, what="dgamma")
    close(bn_mine.bias.grad, bn_ref.bias.grad.cpu().numpy(), rtol=1e-3, atol_scale=2e-5, what="dbeta")
    close(al_b.grad, al_a.grad.cpu().numpy(), rtol=1e-3, atol_scale=1e-4, what="dalpha")
    close(lin.weight.grad, w64.grad.cpu().numpy(), rtol=1e-3, atol_scale=2e-5, what="d head weight")
    if bias:
        close(lin.bias.grad, b64.grad.cpu().numpy(), rtol=1e-3, atol_scale=2e-5, what="d head bias")
    close(bn_mine.running_var, bn_ref.running_var.cpu().numpy(), rtol=2e-5, what="running_var")
    bn_ref.eval(), bn_mine.eval()
    with torch.no_grad():
        want = _dice_ref(bn_ref(h0).double(), al_a.detach()) @ w64.detach().t()
        if bias:
            want = want + b64.detach()
        close(ops.bn_dice_head(h0, bn_mine, al_b.detach(), 1e-3, lin), want.cpu().numpy(), rtol=2e-4, atol_scale=5e-6,
              what="eval")


@pytest.mark.parametrize("rows,C,nscal,extra", [(2048, 256, 1, False), (1600, 128, 2, True), (300, 36, 2, True), (1, 5, 0, False),
                                                (513, 7, 8, True)])
def test_bn_finalize_bwd_tail_sums_every_partial_of_the_statistics_pass_in_one_launch(rows, C, nscal, extra):
    """rh_bn_finalize_bwd_tail: column sums of the (rows, 2, C) partials (= rh_bn_finalize_bwd) plus, in the same launch, the
    column sums of a (rows, C) array and nscal scalar rows -- against float64 sums; both instantiations (rows > 512 / <= 512)."""
    from torch_rechub_amd import _lib, ops
    g = torch.Generator().manual_seed(rows + C)
    part = torch.randn(rows, 2, C, generator=g)
    ex = torch.randn(rows, C, generator=g)
    sc = torch.randn(max(nscal, 1), rows, generator=g)
    d = lambda t: t.to(dev()).contiguous()
    part_d, ex_d, sc_d = d(part), d(ex), d(sc)
    stat = torch.zeros(6, C, device=dev())
    dgamma, dbeta = torch.empty(C, device=dev()), torch.empty(C, device=dev())
    ex_out = torch.full((C,), float("nan"), device=dev())
    sc_out = torch.full((max(nscal, 1),), float("nan"), device=dev())
    _lib.call("rh_bn_finalize_bwd_tail", ops._p(part_d), rows, C, ops._p(stat), ops._p(dgamma), ops._p(dbeta),
              ops._p(ex_d if extra else None), ops._p(ex_out if extra else None), ops._p(sc_d if nscal else None), nscal,
              ops._p(sc_out if nscal else None), ops._stream())
    torch.cuda.synchronize()
    want = part.double().sum(0)
    close(dbeta, want[0].numpy(), rtol=1e-5, atol_scale=1e-6, what="dbeta")
    close(dgamma, want[1].numpy(), rtol=1e-5, atol_scale=1e-6, what="dgamma")
    close(stat[2], want[0].numpy(), rtol=1e-5, atol_scale=1e-6, what="stat row 2")
    close(stat[3], want[1].numpy(), rtol=1e-5, atol_scale=1e-6, what="stat row 3")
    if extra:
        close(ex_out, ex.double().sum(0).numpy(), rtol=1e-5, atol_scale=1e-6, what="extra column sums")
    else:
        assert bool(torch.isnan(ex_out).all())
    if nscal:
        close(sc_out, sc[:nscal].double().sum(1).numpy(), rtol=1e-5, atol_scale=1e-6, what="scalar rows")
    # the plain entry point is the same launch without the tail
    dg2, db2 = torch.empty(C, device=dev()), torch.empty(C, device=dev())
    _lib.call("rh_bn_finalize_bwd", ops._p(part_d), rows, C, ops._p(stat), ops._p(dg2), ops._p(db2), ops._stream())
    torch.cuda.synchronize()
    assert torch.equal(dg2, dgamma) and torch.equal(db2, dbeta)
    with pytest.raises(RuntimeError):
        _lib.call("rh_bn_finalize_bwd_tail", ops._p(part_d), rows, C, ops._p(stat), ops._p(dgamma), ops._p(dbeta), ops._p(ex_d),
                  ops._p(None), ops._p(None), 0, ops._p(None), ops._stream())


# -- row-sharded tables (csrc/shard.hip) and the rectangular in-batch sampler -----------------------------------------
def test_shard_narrow_is_the_saturating_int32_cast_written_into_a_slice():
    """rh_shard_narrow == idx.clamp(-1, 2**31 - 1).to(int32) (sharding._localize, the wire format of the index all-gather), for
    a strided (N, F) view, ids beyond int32 and below -1, written into a slice of a larger buffer (bit-exact: integers)."""
    from torch_rechub_amd import ops
    g = torch.Generator().manual_seed(3)
    N, F = 1000, 7
    base = torch.randint(-5, 2**33, (N, F + 3), generator=g, dtype=torch.int64)
    base[0, 0], base[1, 1], base[2, 2], base[3, 3] = 2**31 - 1, 2**31, -1, -(2**40)
    idx = base.to(dev())[:, :F]  # rows contiguous, row stride F + 3
    buf = torch.full((3 * N, F), 77, dtype=torch.int32, device=dev())
    out = ops.shard_narrow(idx, buf[N:2 * N])
    torch.cuda.synchronize()
    assert out.data_ptr() == buf[N:2 * N].data_ptr()
    want = base[:, :F].clamp(min=-1, max=2**31 - 1).to(torch.int32)
    assert torch.equal(buf[N:2 * N].cpu(), want)
    assert bool((buf[:N] == 77).all()) and bool((buf[2 * N:] == 77).all())
    with pytest.raises(ValueError):
        ops.shard_narrow(idx.t(), buf[:F])



@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
def test_shard_localize_is_the_oracles(world, idx_dtype):
    from torch_rechub_amd import ops
    rng = np.random.default_rng(world)
    vocabs = [1, 3, 10, 17, 64, 1000, 100003]
    pads = [None, 0, 2, None, 63, None, 5]
    idx = np.stack([rng.integers(0, v, 777) for v in vocabs], axis=1)
    desc = torch.tensor(vocabs + [-1 if p is None else p for p in pads] + [-(-v // world) for v in vocabs],
                        dtype=torch.int64, device=dev())
    idx_d = torch.from_numpy(idx).to(idx_dtype).to(dev())
    for rank in range(world):
        got = ops.shard_localize(idx_d, desc, world, rank)
        assert got.dtype == torch.int32
        assert np.array_equal(got.cpu().numpy(), O.shard_localize(idx, vocabs, pads, world, rank))
    ops.check_errors()
    bad = idx_d.clone()
    bad[5, 2] = 10  # == vocab: out of range, as the reference's IndexError
    ops.shard_localize(bad, desc, world, 0)
    with pytest.raises(IndexError):
        ops.check_errors()
    assert ops.shard_localize(idx_d[:0], desc, world, 0).shape == (0, len(vocabs))  # empty batch: no launch


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gather_and_scatter_add_up_to_the_full_table_lookup(world):
    """Every rank's shard, in one process: the masked gathers sum (bit for bit) to the reference lookup, and the
    gradients the shards receive are the oracle's dense table gradient dealt out row by row, sink row untouched."""
    from torch_rechub_amd import ops
    rng = np.random.default_rng(10 + world)
    vocabs, pads, D, N = [3, 10, 305, 4001], [None, 2, None, 7], 16, 1500
    tables = [rng.standard_normal((v, D)).astype(np.float32) for v in vocabs]
    for t, p in zip(tables, pads):
        if p is not None:
            t[p] = 0
    idx = np.stack([rng.integers(0, v, N) for v in vocabs], axis=1)
    g_out = rng.standard_normal((N, len(vocabs) * D)).astype(np.float32)
    idx_d, g_d = torch.from_numpy(idx).to(dev()), torch.from_numpy(g_out).to(dev())
    want = O.embedding_gather(tables, idx).reshape(N, -1)
    want_grads = O.embedding_backward([t.shape for t in tables], idx, g_out.reshape(N, len(vocabs), D).astype(F64),
                                      padding_idx=pads)
    total = torch.zeros(N, len(vocabs) * D, device=dev())
    for rank in range(world):
        sinks = [-(-v // world) for v in vocabs]
        desc = torch.tensor(vocabs + [-1 if p is None else p for p in pads] + sinks, dtype=torch.int64, device=dev())
        shards = [torch.nn.Parameter(torch.from_numpy(O.shard_rows(t, world, rank)).to(dev())) for t in tables]
        loc = ops.shard_localize(idx_d, desc, world, rank)
        call = ops.EmbedCall(shards, sinks, [loc[:, f] for f in range(len(vocabs))], local_grads=True)
        out, _, _ = ops.fused_embedding(call)
        total += out.detach()
        out.backward(g_d)
        for f, (w, full) in enumerate(zip(shards, want_grads)):
            got = ops.grad_buffer(w).cpu().numpy()
            mine = full[rank::world]
            close(got[:mine.shape[0]], mine, rtol=1e-5, atol_scale=2e-6, what=f"rank {rank} table {f}")
            assert not got[mine.shape[0]:].any(), "sink / unused rows must not receive gradient"
    assert np.array_equal(total.cpu().numpy(), want)
    ops.check_errors()


def test_inbatch_sampler_stream_and_rank_slices():
    """Bit-exact against the oracle's restatement of the kernel's stream; slices drawn with (cols, row0) are the rows of
    the square problem (ranks of a job draw what one process draws for the global batch)."""
    from torch_rechub_amd import ops
    seed, K, C = 1234, 7, 96
    ops._sample_rng.clear()
    full = ops.inbatch_sample(C, K, dev(), seed)           # call counter 0
    again = ops.inbatch_sample(C, K, dev(), seed)          # call counter 1
    assert np.array_equal(full.cpu().numpy(), O.inbatch_sample_rows(seed, 0, C, C, 0, K))
    assert np.array_equal(again.cpu().numpy(), O.inbatch_sample_rows(seed, 1, C, C, 0, K))
    for row0, B in ((0, 32), (32, 32), (64, 32), (5, 50)):
        ops._sample_rng.clear()
        part = ops.inbatch_sample(B, K, dev(), seed, cols=C, row0=row0)
        assert torch.equal(part, full[row0:row0 + B])
    ops._sample_rng.clear()
    wide = ops.inbatch_sample(4, 2999, dev(), seed, cols=3000, row0=2996).cpu().numpy()
    assert np.array_equal(wide, O.inbatch_sample_rows(seed, 0, 4, 3000, 2996, 2999))
    with pytest.raises(RuntimeError):
        ops.inbatch_sample(8, 3, dev(), seed, cols=10, row0=5)  # rows 5..12 do not fit 10 columns


def test_inbatch_fast_stream_is_uniform_over_k_subsets_chi_square():
    """The DOCUMENTED deviation of stream="fast" from the reference (utils/match.py:141-145): other indices for the same
    seed, same law.  Each row must draw K distinct columns, never its own, and every other column equally often:
    chi-square over 4000 draws per row (14 degrees of freedom; 60 is beyond the 1 - 1e-7 quantile)."""
    from torch_rechub_amd import ops
    from torch_rechub_amd.utils.match import inbatch_negative_sampling
    B, K, calls = 16, 3, 4000
    ops._sample_rng.clear()
    counts = torch.zeros(B, B, dtype=torch.int64, device=dev())
    ones = torch.ones(B, K, dtype=torch.int64, device=dev())
    scores = torch.zeros(B, B, device=dev())
    g = torch.Generator(device=dev()).manual_seed(99)
    for _ in range(calls):
        idx = inbatch_negative_sampling(scores, neg_ratio=K, generator=g)  # GPU default: the fast stream
        counts.scatter_add_(1, idx, ones)
    c = counts.cpu().numpy().astype(F64)
    assert np.all(np.diag(c) == 0) and np.all(c.sum(1) == calls * K) and c.max() <= calls  # distinct, never own
    expect = calls * K / (B - 1)
    off = ~np.eye(B, dtype=bool)
    chi2 = (((c - expect)**2 / expect) * off).sum(1)
    assert chi2.max() < 60.0, chi2
    # the opt-in reference stream on the device: valid draws, reproducible from the generator state
    a = inbatch_negative_sampling(scores, neg_ratio=K, generator=torch.Generator(device=dev()).manual_seed(5),
                                  stream="reference")
    b = inbatch_negative_sampling(scores, neg_ratio=K, generator=torch.Generator(device=dev()).manual_seed(5),
                                  stream="reference")
    assert torch.equal(a, b) and not (a == torch.arange(B, device=dev()).unsqueeze(1)).any()
    assert all(len(set(r)) == K for r in a.cpu().tolist())


@pytest.mark.parametrize("B,F,D,ND", [(257, 26, 16, 13), (64, 3, 8, 0), (1, 5, 32, 2), (0, 4, 16, 1)])
def test_fused_rows_is_the_fused_gather_stage_on_rows_in_place(B, F, D, ND):
    """ops.fused_rows (what a row-sharded lookup feeds): flattened rows + dense values, FM and LR of layers.py:112-120,
    :313-319, :185-189 and their gradients, against the float64 oracle / torch autograd on the same rows."""
    from torch_rechub_amd import ops
    g = torch.Generator().manual_seed(B + F)
    emb = (torch.randn(B, F * D, generator=g) * 0.5).to(dev()).requires_grad_(True)
    dense = torch.rand(B, max(ND, 1), generator=g).to(dev())
    w = (torch.randn(1, F * D, generator=g) * 0.1).to(dev()).requires_grad_(True)
    b = torch.randn(1, generator=g).to(dev()).requires_grad_(True)
    out, fm, lr = ops.fused_rows(emb, F, [dense[:, j] for j in range(ND)], w, b, want_fm=True)
    x = emb.detach().cpu().numpy().astype(F64).reshape(B, F, D)
    assert np.array_equal(out[:, :F * D].detach().cpu().numpy(), emb.detach().cpu().numpy())
    assert np.array_equal(out[:, F * D:].detach().cpu().numpy(), dense[:, :ND].cpu().numpy())
    close(fm, O.fm_forward(x), what="fm")
    close(lr, O.lr_forward(x.reshape(B, F * D), w.detach().cpu().numpy().astype(F64), b.detach().cpu().numpy().astype(F64)),
          what="lr")
    G1 = torch.randn(out.shape, generator=g).to(dev())
    G2, G3 = torch.randn(B, 1, generator=g).to(dev()), torch.randn(B, 1, generator=g).to(dev())
    ((out * G1).sum() + (fm * G2).sum() + (lr * G3).sum()).backward()
    got = [emb.grad.clone(), w.grad.clone(), b.grad.clone()]
    e2 = emb.detach().double().requires_grad_(True)
    w2, b2 = w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    x3 = e2.view(B, F, D)
    fm2 = 0.5 * (x3.sum(1)**2 - (x3**2).sum(1)).sum(1, keepdim=True)
    lr2 = e2 @ w2.t() + b2
    ((e2 * G1[:, :F * D].double()).sum() + (fm2 * G2.double()).sum() + (lr2 * G3.double()).sum()).backward()
    close(got[0], e2.grad.cpu().numpy(), what="row gradients")
    close(got[1], w2.grad.cpu().numpy(), rtol=2e-5, atol_scale=2e-6, what="LR weight gradient")
    close(got[2], b2.grad.cpu().numpy(), rtol=2e-5, atol_scale=2e-6, what="LR bias gradient")
    plain, none_fm, none_lr = ops.fused_rows(emb.detach(), F)
    assert none_fm is None and none_lr is None and torch.equal(plain, emb.detach())
    ops.check_errors()


@pytest.mark.parametrize("B,T,D", [(37, 9, 16), (130, 5, 8), (64, 1, 4), (5, 12, 32), (300, 100, 16)])
def test_augru_recurrence_forward_and_backward_through_time(B, T, D):
    """ops.augru (csrc/augru.hip) against the float64 oracle, which tests/test_oracle_golden.py pins to the reference's
    AUGRU (dien.py:30-66): every state; gradients of xw, the attention weights and U for a random upstream gradient on
    every state.  Tolerance: fp32 recurrence of up to 100 dependent steps with the hardware exponent / reciprocal
    (v_exp_f32, v_rcp_f32: 1 ulp each) vs float64 -- states rtol 1e-4 + 1e-5 * max, gradients rtol 5e-4 + 1e-5 * max."""
    from torch_rechub_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + T * 10 + D)
    xw = (torch.randn(B, T, 3 * D, generator=g) * 0.8)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    attn = torch.rand(B, T, generator=g) * (torch.arange(T)[None, :] < lens[:, None])  # 0 on padded steps
    U = torch.randn(D, 3 * D, generator=g) * (0.5 / D**0.5)
    G = torch.randn(B, T, D, generator=g)
    xd, ad, ud = (t.to(dev()).requires_grad_(True) for t in (xw, attn, U))
    h_all = ops.augru(xd, ad, ud)
    want = O.augru_forward(xw.numpy().astype(F64), attn.numpy().astype(F64), U.numpy().astype(F64))
    close(h_all, want, rtol=1e-4, atol_scale=1e-5, what="states")
    h_all.backward(G.to(dev()))
    d_xw, d_attn, d_U = O.augru_backward(xw.numpy().astype(F64), attn.numpy().astype(F64), U.numpy().astype(F64),
                                         G.numpy().astype(F64))
    close(xd.grad, d_xw, rtol=5e-4, atol_scale=1e-5, what="d xw")
    close(ad.grad, d_attn, rtol=5e-4, atol_scale=1e-5, what="d attn")
    close(ud.grad, d_U, rtol=5e-4, atol_scale=1e-5, what="d U")
    # a padded step leaves the state where it was
    hs = h_all.detach().cpu()
    stay = (attn[:, 1:] == 0)
    assert torch.equal(hs[:, 1:][stay], hs[:, :-1][stay])


@pytest.mark.parametrize("B,T,H", [(70, 7, 16), (9, 30, 8), (33, 3, 32), (4, 5, 4)])
def test_gru_on_the_recurrence_kernel_is_torch_nn_gru(B, T, H):
    """ops.gru (nn.GRU's r, z, n cell expressed on csrc/augru.hip: update gate 1 - z, weight 1, bias_hh on the state
    product) against the float64 oracle of the same mapping -- which reproduces torch.nn.GRU to 1e-15 on CPU -- and
    against the module's own (library) forward on the device: outputs, input gradient and all four parameter gradients."""
    from torch_rechub_amd import ops
    torch.manual_seed(B + T + H)
    gru = torch.nn.GRU(H, H, batch_first=True).to(dev())
    x = torch.randn(B, T, H, device=dev(), requires_grad=True)
    G = torch.randn(B, T, H, device=dev())
    assert ops.gru_ok(gru, x)
    out = ops.gru(gru, x)
    (out * G).sum().backward()
    got = [x.grad.clone()] + [p.grad.clone() for p in gru.parameters()]
    x.grad = None
    gru.zero_grad()
    ref, _ = gru(x)
    (ref * G).sum().backward()
    want = [x.grad.clone()] + [p.grad.clone() for p in gru.parameters()]
    close(out, ref.detach().cpu().numpy(), rtol=1e-4, atol_scale=1e-5, what="outputs vs library GRU")
    for a, b, name in zip(got, want, ["x", "weight_ih", "weight_hh", "bias_ih", "bias_hh"]):
        close(a, b.cpu().numpy(), rtol=5e-4, atol_scale=2e-5, what=f"grad {name} vs library GRU")

    def gates(m):
        return np.concatenate([-m[H:2 * H], m[:H], m[2 * H:]], axis=0)

    w_ih, w_hh, b_ih, b_hh = (p.detach().cpu().numpy().astype(F64) for p in gru.parameters())
    xw = x.detach().cpu().numpy().astype(F64) @ gates(w_ih).T + gates(b_ih)
    close(out, O.augru_forward(xw, None, gates(w_hh).T, gates(b_hh)), rtol=1e-4, atol_scale=1e-5, what="outputs vs oracle")
    assert not ops.gru_ok(torch.nn.GRU(H, H, batch_first=True, num_layers=2).to(dev()), x)
    assert not ops.gru_ok(torch.nn.GRU(H, 12, batch_first=True).to(dev()), x)


def test_sharded_history_lookup_many_positions_of_one_table():
    """A concat-pooled history on a row-sharded table = L lookups of ONE table per sample (F = L = 100 'fields' sharing a
    shard): per-rank gathers sum bit-exactly to the reference lookup, the shard's gradient is the oracle's shared-table
    gradient dealt out row by row, padding positions and the sink row receive nothing."""
    from torch_rechub_amd import ops
    world, vocab, D, L, N = 2, 5003, 16, 100, 300
    rng = np.random.default_rng(77)
    table = rng.standard_normal((vocab, D)).astype(np.float32)
    table[0] = 0  # padding_idx = 0
    idx = rng.integers(1, vocab, (N, L))
    lens = rng.integers(1, L + 1, N)
    idx[np.arange(L)[None, :] >= lens[:, None]] = 0
    g_out = rng.standard_normal((N, L * D)).astype(np.float32)
    idx_d, g_d = torch.from_numpy(idx).to(dev()), torch.from_numpy(g_out).to(dev())
    want = O.embedding_gather([table] * L, idx).reshape(N, -1)
    shared = np.zeros((vocab, D), dtype=F64)
    O.embedding_backward([(vocab, D)] * L, idx, g_out.reshape(N, L, D).astype(F64), padding_idx=[0] * L,
                         out=[shared] * L)
    sink = -(-vocab // world)
    desc = torch.tensor([vocab] * L + [0] * L + [sink] * L, dtype=torch.int64, device=dev())
    total = torch.zeros(N, L * D, device=dev())
    for rank in range(world):
        shard = torch.nn.Parameter(torch.from_numpy(O.shard_rows(table, world, rank)).to(dev()))
        loc = ops.shard_localize(idx_d, desc, world, rank)
        call = ops.EmbedCall([shard] * L, [sink] * L, [loc[:, j] for j in range(L)], local_grads=True)
        out, _, _ = ops.fused_embedding(call)
        total += out.detach()
        out.backward(g_d)
        got = ops.grad_buffer(shard).cpu().numpy()
        mine = shared[rank::world]
        close(got[:mine.shape[0]], mine, rtol=1e-5, atol_scale=2e-6, what=f"rank {rank} shard gradient")
        assert not got[mine.shape[0]:].any()
    assert np.array_equal(total.cpu().numpy(), want)
    ops.check_errors()


def test_torch_library_ops_opcheck_and_values():
    """torch.ops.rechub_hip.{fm, cross_network, dice}: schema / fake impl / autograd registration validated by
    torch.library.opcheck on the device, values and gradients equal to the autograd.Function path of the layers."""
    import torch_rechub_amd.library  # noqa: F401
    from torch_rechub_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(33, 7, 16, generator=g).to(dev()).requires_grad_()
    z = torch.randn(65, 45, generator=g).to(dev()).requires_grad_()
    W = (torch.randn(3, 45, generator=g) / 8).to(dev()).requires_grad_()
    b = (torch.randn(3, 45, generator=g) / 8).to(dev()).requires_grad_()
    h = torch.randn(50, 36, generator=g).to(dev()).requires_grad_()
    alpha = torch.randn(1, generator=g).to(dev()).requires_grad_()
    tests = ("test_schema", "test_faketensor", "test_autograd_registration")
    torch.library.opcheck(torch.ops.rechub_hip.fm.default, (x, True), test_utils=tests)
    torch.library.opcheck(torch.ops.rechub_hip.cross_network.default, (z, W, b), test_utils=tests)
    torch.library.opcheck(torch.ops.rechub_hip.dice.default, (h, alpha, 1e-9), test_utils=tests)
    for a, bfn, args in ((torch.ops.rechub_hip.fm, ops.fm, (x, True)),
                         (torch.ops.rechub_hip.cross_network, ops.cross_network, (z, W, b)),
                         (torch.ops.rechub_hip.dice, ops.dice, (h, alpha, 1e-9))):
        leaves = [t for t in args if torch.is_tensor(t)]
        ya = a(*args)
        ga = torch.autograd.grad(ya.sum() * 0.5 + (ya * ya).sum(), leaves)
        yb = bfn(*args)
        gb = torch.autograd.grad(yb.sum() * 0.5 + (yb * yb).sum(), leaves)
        assert torch.equal(ya, yb)
        for u, v in zip(ga, gb):
            assert torch.allclose(u, v, rtol=1e-6, atol=1e-7)


def test_torch_library_round4_ops_opcheck_and_values():
    """torch.ops.rechub_hip.{cross_net_v2, cross_net_mix, din_attention_input, din_attention_pool, embedding_bag_masked,
    inbatch_negative_sample}: opcheck (schema, fake impl, autograd registration, AOT dispatch) on the device;
    values and gradients against the modules / autograd.Function paths the models run, which the reference fixtures pin."""
    import torch_rechub_amd.library  # noqa: F401
    from torch_rechub_amd import ops
    from torch_rechub_amd.basic.layers import CrossNetMix, CrossNetV2
    g = torch.Generator().manual_seed(4)
    tests = ("test_schema", "test_faketensor", "test_autograd_registration", "test_aot_dispatch_dynamic")
    B, d, L = 70, 44, 3

    def leaf(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g) * scale).to(dev()).requires_grad_()

    def same_grads(ya, la, yb, lb, rtol=2e-5):
        G = torch.randn(ya.shape, generator=g).to(dev())
        ga = torch.autograd.grad((ya * G).sum(), la)
        gb = torch.autograd.grad((yb * G).sum(), lb)
        for u, v in zip(ga, gb):
            close(u, v.detach().cpu().numpy(), rtol=rtol, atol_scale=2e-6, what="gradient")

    # CrossNetV2 against the module
    x, W, b = leaf(B, d), leaf(L, d, d, scale=0.1), leaf(L, d, scale=0.1)
    torch.library.opcheck(torch.ops.rechub_hip.cross_net_v2.default, (x, W, b), test_utils=tests)
    mod = CrossNetV2(d, L).to(dev())
    with torch.no_grad():
        for l in range(L):
            mod.w[l].weight.copy_(W[l])
            mod.b[l].copy_(b[l])
    xa, xb = x.detach().clone().requires_grad_(), x.detach().clone().requires_grad_()
    ya, yb = torch.ops.rechub_hip.cross_net_v2(xa, W, b), mod(xb)
    close(ya, yb.detach().cpu().numpy(), rtol=1e-5, atol_scale=1e-6, what="cross_net_v2")
    G = torch.randn(ya.shape, generator=g).to(dev())
    ga = torch.autograd.grad((ya * G).sum(), [xa, W, b])
    (yb * G).sum().backward()
    close(ga[0], xb.grad.cpu().numpy(), rtol=2e-5, atol_scale=2e-6, what="cross_net_v2 g_x")
    for l in range(L):
        close(ga[1][l], mod.w[l].weight.grad.cpu().numpy(), rtol=2e-5, atol_scale=2e-6, what=f"cross_net_v2 g_W[{l}]")
        close(ga[2][l], mod.b[l].grad.cpu().numpy(), rtol=2e-5, atol_scale=2e-6, what=f"cross_net_v2 g_b[{l}]")

    # CrossNetMix against the module (csrc/moe.hip path: rank 8, 4 experts)
    E, r = 4, 8
    mix = CrossNetMix(d, num_layers=L, low_rank=r, num_experts=E).to(dev())
    with torch.no_grad():
        for p_ in mix.parameters():
            p_.normal_(0, 0.2, generator=None)
    U = torch.stack([u.detach() for u in mix.u_list]).requires_grad_()
    V = torch.stack([v.detach() for v in mix.v_list]).requires_grad_()
    C = torch.stack([c.detach() for c in mix.c_list]).requires_grad_()
    bias = torch.stack([bb.detach().reshape(-1) for bb in mix.bias]).requires_grad_()
    gating = torch.stack([gg.weight.detach().reshape(-1) for gg in mix.gating]).requires_grad_()
    xm = leaf(B, d)
    torch.library.opcheck(torch.ops.rechub_hip.cross_net_mix.default, (xm, U, V, C, bias, gating), test_utils=tests)
    xa, xb = xm.detach().clone().requires_grad_(), xm.detach().clone().requires_grad_()
    ya, yb = torch.ops.rechub_hip.cross_net_mix(xa, U, V, C, bias, gating), mix(xb)
    close(ya, yb.detach().cpu().numpy(), rtol=1e-5, atol_scale=1e-6, what="cross_net_mix")
    G = torch.randn(ya.shape, generator=g).to(dev())
    ga = torch.autograd.grad((ya * G).sum(), [xa, U, V, C, bias, gating])
    (yb * G).sum().backward()
    close(ga[0], xb.grad.cpu().numpy(), rtol=5e-5, atol_scale=5e-6, what="cross_net_mix g_x")
    for l in range(L):
        close(ga[1][l], mix.u_list[l].grad.cpu().numpy(), rtol=5e-5, atol_scale=5e-6, what="cross_net_mix g_U")
        close(ga[2][l], mix.v_list[l].grad.cpu().numpy(), rtol=5e-5, atol_scale=5e-6, what="cross_net_mix g_V")
        close(ga[3][l], mix.c_list[l].grad.cpu().numpy(), rtol=5e-5, atol_scale=5e-6, what="cross_net_mix g_C")
        close(ga[4][l], mix.bias[l].grad.reshape(-1).cpu().numpy(), rtol=5e-5, atol_scale=5e-6, what="cross_net_mix g_bias")
    for e in range(E):
        close(ga[5][e], mix.gating[e].weight.grad.reshape(-1).cpu().numpy(), rtol=5e-5, atol_scale=5e-6, what="g_gating")

    # DIN attention kernels against eager torch (din.py:79-81, :89-92)
    hist, tgt, aw = leaf(9, 13, 16), leaf(9, 16), leaf(9, 13)
    torch.library.opcheck(torch.ops.rechub_hip.din_attention_input.default, (hist, tgt), test_utils=tests)
    torch.library.opcheck(torch.ops.rechub_hip.din_attention_pool.default, (aw, hist), test_utils=tests)
    t3 = tgt.unsqueeze(1).expand(-1, 13, -1)
    ref_in = torch.cat([t3, hist, t3 - hist, t3 * hist], dim=-1).view(-1, 64)
    ya = torch.ops.rechub_hip.din_attention_input(hist, tgt)
    assert torch.equal(ya, ref_in)
    same_grads(ya, [hist, tgt], ref_in, [hist, tgt])
    ref_pool = (aw.unsqueeze(-1) * hist).sum(dim=1)
    yp = torch.ops.rechub_hip.din_attention_pool(aw, hist)
    close(yp, ref_pool.detach().cpu().numpy(), rtol=1e-5, atol_scale=1e-6, what="din_attention_pool")
    same_grads(yp, [aw, hist], ref_pool, [aw, hist])

    # masked embedding bag against eager torch (layers.py:223-229, :247-251, :204-205)
    tab = leaf(60, 16)
    seq = torch.randint(0, 60, (11, 7), generator=g)
    seq[:, 4:] = 0  # post-padded
    seq = seq.to(dev())
    for mode, sentinel in (("sum", 0), ("mean", 0), ("concat", -1), ("mean", -1)):
        torch.library.opcheck(torch.ops.rechub_hip.embedding_bag_masked.default, (tab, seq, sentinel, mode), test_utils=tests)
        rows = tab[seq]
        mask = (seq != sentinel).float().unsqueeze(-1)
        ref = rows if mode == "concat" else ((rows * mask).sum(1) if mode == "sum" else (rows * mask).sum(1) / (mask.sum(1) + 1e-16))
        got = torch.ops.rechub_hip.embedding_bag_masked(tab, seq, sentinel, mode)
        close(got, ref.detach().cpu().numpy(), rtol=1e-5, atol_scale=1e-6, what=f"embedding_bag_masked {mode}")
        same_grads(got, [tab], ref, [tab])

    # in-batch negatives: hard = the reference's known answer (tests/test_inbatch_sampling.py:26-30), random = invariants
    scores = torch.tensor([[0.0, 0.1, 0.9], [0.2, 0.0, 0.8], [0.3, 0.7, 0.0]], device=dev())
    torch.library.opcheck(torch.ops.rechub_hip.inbatch_negative_sample.default, (scores, 1, True, 0, 0), test_utils=tests[:2])
    assert torch.ops.rechub_hip.inbatch_negative_sample(scores, 1, True, 0, 0).view(-1).tolist() == [2, 2, 1]
    big = torch.randn(256, 256, generator=g).to(dev())
    n1 = torch.ops.rechub_hip.inbatch_negative_sample(big, 20, False, 2022, 0)
    n2 = torch.ops.rechub_hip.inbatch_negative_sample(big, 20, False, 2022, 0)
    n3 = torch.ops.rechub_hip.inbatch_negative_sample(big, 20, False, 2022, 1)
    assert torch.equal(n1, n2) and not torch.equal(n1, n3)
    rows_ = torch.arange(256, device=dev()).unsqueeze(1)
    assert bool((n1 != rows_).all()) and int(n1.min()) >= 0 and int(n1.max()) < 256
    assert all(len(set(r_)) == 20 for r_ in n1.cpu().tolist())
    ops.check_errors()


def test_torch_library_functional_gather_and_adam_ops():
    """torch.ops.rechub_hip.embedding_fm_lr (the fused gather + FM + LR as a FUNCTIONAL op: per-lookup gradient rows ->
    one dense gradient per table in the autograd formula) and adam_step_ (mutated arguments declared in the schema):
    opcheck on the device; values and every gradient equal to the trainers' autograd.Function path (ops.fused_embedding,
    persistent gradient buffers) and to the oracle's Adam."""
    import torch_rechub_amd.library  # noqa: F401
    from torch_rechub_amd import ops
    g = torch.Generator().manual_seed(11)
    vocabs, B, D, ND = [7, 300, 41, 300], 53, 16, 3
    tabs = [(torch.randn(v, D, generator=g) * 0.3).to(dev()).requires_grad_() for v in vocabs[:3]]
    tabs.append(tabs[1])  # a shared table (shared_with): field 3 indexes table 1
    idx = torch.stack([torch.randint(0, v, (B,), generator=g) for v in vocabs], 1).to(dev())
    dense = torch.rand(B, ND, generator=g).to(dev()).requires_grad_()
    lr_w = (torch.randn(1, 4 * D, generator=g) * 0.2).to(dev()).requires_grad_()
    lr_b = torch.randn(1, generator=g).to(dev()).requires_grad_()
    tests = ("test_schema", "test_faketensor", "test_autograd_registration")
    torch.library.opcheck(torch.ops.rechub_hip.embedding_fm_lr.default, (tabs, idx, dense, lr_w, lr_b), test_utils=tests)
    out, fm, lr, _ = torch.ops.rechub_hip.embedding_fm_lr(tabs, idx, dense, lr_w, lr_b)
    G = torch.randn(out.shape, generator=g).to(dev())
    leaves = tabs[:3] + [dense, lr_w, lr_b]
    ga = torch.autograd.grad((out * G).sum() + 0.7 * fm.sum() + (lr * lr).sum(), leaves)
    # the trainers' path on twins of the same tensors
    tw = [t.detach().clone().requires_grad_() for t in tabs[:3]]
    tw.append(tw[1])
    d2, w2, b2 = dense.detach().clone().requires_grad_(), lr_w.detach().clone().requires_grad_(), lr_b.detach().clone().requires_grad_()
    call = ops.EmbedCall(tw, [None] * 4, [idx[:, f] for f in range(4)], dense=[d2[:, j] for j in range(ND)], want_fm=True,
                         want_lr=True)
    o2, f2, l2 = ops.fused_embedding(call, w2, b2)
    assert torch.equal(out, o2) and torch.equal(fm, f2) and torch.equal(lr, l2)
    ((o2 * G).sum() + 0.7 * f2.sum() + (l2 * l2).sum()).backward()
    ops.check_errors()
    for a, t in zip(ga[:3], tw[:3]):
        close(a, t.grad.detach().cpu().numpy(), rtol=1e-5, atol_scale=1e-6, what="table gradient")
    assert d2.grad is None  # the trainers' op treats the dense values as data; the functional op differentiates them:
    close(ga[3], G[:, 4 * D:].cpu().numpy(), what="dense gradient")  # out's dense block is a copy of `dense`
    close(ga[4], w2.grad.cpu().numpy(), rtol=1e-5, atol_scale=1e-6, what="lr weight gradient")
    close(ga[5], b2.grad.cpu().numpy(), rtol=1e-5, atol_scale=1e-6, what="lr bias gradient")
    # adam_step_: three steps of one tensor against the float64 oracle (coupled weight decay)
    p = torch.randn(40, 16, generator=g).to(dev())
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    pr, mr, vr = p.cpu().numpy().astype(F64), np.zeros((40, 16)), np.zeros((40, 16))
    torch.library.opcheck(torch.ops.rechub_hip.adam_step_.default,
                          (p.clone(), torch.randn(40, 16, generator=g).to(dev()), m.clone(), v.clone(), 1, 1e-2, 0.9, 0.999, 1e-8,
                           1e-3), test_utils=("test_schema", "test_faketensor"))
    for t in range(1, 4):
        gr = torch.randn(40, 16, generator=g)
        gd = gr.to(dev())
        torch.ops.rechub_hip.adam_step_(p, gd, m, v, t, 1e-2, 0.9, 0.999, 1e-8, 1e-3)
        assert not gd.any()  # re-zeroed in the pass
        pr, mr, vr = O.adam_step(pr, gr.numpy().astype(F64), mr, vr, t, lr=1e-2, weight_decay=1e-3)
        close(p, pr, rtol=1e-5, atol_scale=1e-6, what=f"adam_step_ step {t}")


@pytest.mark.parametrize("shapes", [
    [(1, 1000), (3, 429), (512, 429), (33, 64), (700, 1), (64, 7), (0, 50), (40, 300), (200, 5000)],
    [(2048, 65), (31, 31), (32, 256), (33, 256)] + [(1, 17)] * 40,  # > 32 items: two launches
])
def test_pack_grads_sums_partial_slabs_in_one_launch(shapes):
    """rh_pack_grads: flat[off + i] = sum_r src[r, i] (+ add[i]); plain copies, few parts, deep stacks of partial rows (the
    rows split over the wavefronts), tall-and-thin bias stacks, a parameter without gradient (zeros)."""
    import ctypes

    from torch_rechub_amd import _lib, ops
    g = torch.Generator().manual_seed(len(shapes))
    items = (_lib.PackItem * len(shapes))()
    keep, want, off = [], [], 0
    for i, (nparts, numel) in enumerate(shapes):
        stride = numel + (3 if i % 2 else 0)  # slabs with a pitch
        src = torch.randn(max(nparts, 1), stride, generator=g).to(dev())
        add = torch.randn(numel, generator=g).to(dev()) if i % 3 == 0 and nparts > 0 else None
        keep += [src, add]
        it = items[i]
        it.src, it.nparts, it.stride = (src.data_ptr(), nparts, stride) if nparts else (0, 0, 0)
        it.add = add.data_ptr() if add is not None else 0
        it.numel, it.dst_offset = numel, off
        ref = src[:nparts, :numel].double().sum(0) if nparts else torch.zeros(numel, dtype=torch.float64, device=dev())
        want.append(ref + (add.double() if add is not None else 0))
        off += numel
    flat = torch.full((off + 8,), 7.0, device=dev())
    _lib.call("rh_pack_grads", ctypes.cast(items, ctypes.c_void_p), len(shapes), ops._p(flat), ops._stream())
    torch.cuda.synchronize()
    got, ref = flat[:off].double(), torch.cat(want)
    scale = max(1.0, float(ref.abs().max()))
    assert float((got - ref).abs().max()) <= 2e-5 * scale
    assert torch.all(flat[off:] == 7.0)
    again = flat.clone()
    _lib.call("rh_pack_grads", ctypes.cast(items, ctypes.c_void_p), len(shapes), ops._p(flat), ops._stream())
    assert torch.equal(flat, again)  # fixed summation order: bit-reproducible


def _cross_net_mix_reference(x, U, V, C, bias, Wg):
    """CrossNetMix.forward as the reference writes it (basic/layers.py:470-506), in float64 on the CPU."""
    x0 = x.unsqueeze(2)
    xl = x0
    for i in range(len(U)):
        outs, scores = [], []
        for e in range(len(Wg)):
            scores.append(xl.squeeze(2) @ Wg[e].t())
            v = torch.tanh(torch.matmul(V[i][e].t(), xl))
            v = torch.tanh(torch.matmul(C[i][e], v))
            outs.append((x0 * (torch.matmul(U[i][e], v) + bias[i])).squeeze(2))
        moe = torch.matmul(torch.stack(outs, 2), torch.stack(scores, 1).softmax(1))
        xl = moe + xl
    return xl.squeeze(2)


@pytest.mark.parametrize("B,d,L,E,r,strided", [
    (300, 429, 3, 4, 32, True),    # the DCN-v2 configuration of BASELINE.json configs[2]
    (64, 37, 2, 3, 8, False),      # the golden fixture's shape: 24 threads per sample, 10 samples per pass
    (1, 5, 1, 1, 4, False),
    (1000, 64, 2, 2, 64, False),
    (257, 130, 8, 16, 16, True),   # the limits: 8 layers, 16 experts, 256 threads per sample
    (40, 33, 2, 4, 5, False),      # rank 5: not a kernel rank -> the batched-GEMM formulation
])
def test_cross_net_mix_two_product_form_vs_reference_formula(B, d, L, E, r, strided):
    from torch_rechub_amd import ops
    from torch_rechub_amd.basic.layers import CrossNetMix
    g = torch.Generator().manual_seed(B + d)
    mix = CrossNetMix(d, num_layers=L, low_rank=r, num_experts=E)
    with torch.no_grad():
        for b in mix.bias:
            b.copy_(torch.randn(b.shape, generator=g) * 0.3)
        for t in list(mix.u_list) + list(mix.v_list):
            t.mul_(3.0)
    x = torch.randn(B, d, generator=g)
    gy = torch.randn(B, d, generator=g)
    ref_params = [[t.detach().double().requires_grad_() for t in lst] for lst in
                  (mix.u_list, mix.v_list, mix.c_list, mix.bias, [m.weight for m in mix.gating])]
    xr = x.double().requires_grad_()
    want = _cross_net_mix_reference(xr, *ref_params)
    want.backward(gy.double())
    mix = mix.to(dev())
    assert bool(ops.cross_moe_ok(x.to(dev()), L, E, d, r)) == (r != 5)
    if strided:  # the embedding layer hands over a (B, d) view of a buffer with a 64-byte row pitch
        buf = torch.zeros(B, (d + 15) // 16 * 16, device=dev())
        buf[:, :d] = x.to(dev())
        xd = buf[:, :d].detach().requires_grad_()
    else:
        xd = x.to(dev()).requires_grad_()
    out = mix(xd)
    gyd = gy.to(dev())
    out.backward(gyd)
    torch.cuda.synchronize()
    assert torch.equal(gyd.cpu(), gy)  # (the backward accumulates in place in ITS OWN buffers only, never in the incoming gradient)
    close(out, want.detach().numpy(), rtol=2e-5, atol_scale=2e-6, what="crossmix out")
    close(xd.grad, xr.grad.numpy(), rtol=1e-4, atol_scale=1e-5, what="crossmix g_x")
    ours = (mix.u_list, mix.v_list, mix.c_list, mix.bias, [m.weight for m in mix.gating])
    for name, mine, theirs in zip("U V C bias gating".split(), ours, ref_params):
        for i, (a, b) in enumerate(zip(mine, theirs)):
            close(a.grad, b.grad.numpy(), rtol=1e-4, atol_scale=1e-5, what=f"crossmix g_{name}[{i}]")


def test_grouped_weight_gradient_launch_is_the_single_launches_bit_for_bit(monkeypatch):
    """rh_linear_wgrad_partial_group (round 4: the 2 L weight gradients of CrossNetMix's backward as ONE launch) runs the
    same workgroup body on the same (tile, split) decomposition as 2 L single launches: every parameter gradient of the
    stack must be bit-equal between the two forms (configs[2] shape: d = 429, 3 layers, 4 experts, rank 32)."""
    from torch_rechub_amd import ops
    from torch_rechub_amd.basic.layers import CrossNetMix
    torch.manual_seed(3)
    mix = CrossNetMix(429, num_layers=3, low_rank=32, num_experts=4).to(dev())
    x = torch.randn(4096, 432, device=dev())[:, :429]
    gy = torch.randn(4096, 429, device=dev())
    got = {}
    for flag in (True, False):
        monkeypatch.setattr(ops, "WGRAD_GROUP", flag)
        mix.zero_grad()
        xi = x.detach().clone().requires_grad_()
        mix(xi).backward(gy)
        got[flag] = [xi.grad.clone()] + [p_.grad.clone() for p_ in mix.parameters()]
    torch.cuda.synchronize()
    assert len(got[True]) == len(got[False]) > 10
    for a_, b_ in zip(got[True], got[False]):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("B,L,D,dims,softmax", [
    (37, 50, 16, [256, 128], False),   # configs[3] layer widths; 1850 rows: not a multiple of 32
    (5, 7, 8, [64], True),
    (64, 100, 4, [128, 64], False),
    (3, 11, 16, [192], False),
])
def test_activation_unit_first_layer_on_register_built_operand(B, L, D, dims, softmax, monkeypatch):
    """The ActivationUnit with its first Linear on the never-materialised [t, h, t-h, t*h] operand + BatchNorm statistics
    from the GEMM epilogue (csrc/dinmlp.hip) against the same module on the materialised operand (rh_din_att_input +
    library GEMM + statistics pass): outputs, running statistics and every gradient."""
    from torch_rechub_amd import ops
    from torch_rechub_amd.models.ranking.din import ActivationUnit
    g = torch.Generator().manual_seed(B + L)
    au = ActivationUnit(D, dims=dims, activation="dice", use_softmax=softmax)
    twin = ActivationUnit(D, dims=dims, activation="dice", use_softmax=softmax)
    twin.load_state_dict(au.state_dict())
    au, twin = au.to(dev()).train(), twin.to(dev()).train()
    big = torch.randn(B, L + 3, D, generator=g).to(dev())
    hist = big[:, 1:L + 1, :]  # rows contiguous inside a sample, sample stride (L + 3) * D
    tgt = torch.randn(B, D, generator=g).to(dev())
    gy = torch.randn(B, D, generator=g).to(dev())
    assert ops.din_att_l1_ok(hist, tgt, au.attention.mlp[0])
    h1, t1 = hist.detach().clone().requires_grad_(), tgt.detach().clone().requires_grad_()
    out = au(h1, t1)
    out.backward(gy)
    monkeypatch.setattr(ops, "din_att_l1_ok", lambda *a: False)
    h2, t2 = hist.detach().clone().requires_grad_(), tgt.detach().clone().requires_grad_()
    ref = twin(h2, t2)
    ref.backward(gy)
    torch.cuda.synchronize()

    def same(a, b, what, rtol=2e-4, atol=2e-5, floor=1.0):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        scale = max(floor, float(b.abs().max()))
        assert float((a - b).abs().max()) <= atol * scale + rtol * float(b.abs().max()), what

    same(out, ref, "attention output")
    same(h1.grad, h2.grad, "g_history")
    same(t1.grad, t2.grad, "g_target")
    # a Linear bias in front of BatchNorm has the exact gradient 0: what both paths hold there is rounding noise of the
    # size of one ulp of the layer's other gradients, so every parameter is compared on the scale of the largest gradient
    gmax = max(float(q.grad.abs().max()) for q in twin.parameters())
    for (n, p), (_, q) in zip(au.named_parameters(), twin.named_parameters()):
        same(p.grad, q.grad, f"g_{n}", floor=gmax)
    for (n, p), (_, q) in zip(au.named_buffers(), twin.named_buffers()):
        same(p, q, f"buffer {n}", rtol=1e-5, atol=1e-6)
    # inference (eval mode, no autograd): the fused first layer without the statistics epilogue + running statistics
    au.eval(), twin.eval()
    with torch.no_grad():
        ref_e = twin(hist, tgt)
        monkeypatch.undo()
        assert ops.din_att_l1_ok(hist, tgt, au.attention.mlp[0])
        out_e = au(hist, tgt)
    same(out_e, ref_e, "attention output (eval)")


@pytest.mark.parametrize("tag", ["w256", "w64", "w128sm", "w192d4"])
def test_activation_unit_wide_matches_reference_fixture_through_fused_first_layer(tag, monkeypatch):
    """The path configs[3] runs -- first attention Linear on the register-built operand (csrc/dinmlp.hip), BatchNorm
    statistics from its epilogue (rh_bn_stats_from_partial), tile-GEMM input gradient -- against the UNMODIFIED
    reference ActivationUnit in train mode (tests/golden/au_wide.npz, oracle/gen_golden.py::gen_au_wide; reference
    models/ranking/din.py:58-93 at the widths of examples/ranking/run_amazon_electronics.py:57): output, g_history,
    g_target, every parameter gradient, and the BatchNorm running statistics after the forward."""
    from torch_rechub_amd import ops
    from torch_rechub_amd.models.ranking.din import ActivationUnit
    from conftest import load_golden
    g = load_golden("au_wide.npz")
    k = tag + "."
    dims, sm = [int(v) for v in g[k + "dims"]], bool(int(g[k + "softmax"]))
    hist = torch.from_numpy(g[k + "hist"]).to(dev()).requires_grad_()
    tgt = torch.from_numpy(g[k + "tgt"]).to(dev()).requires_grad_()
    au = ActivationUnit(hist.shape[2], dims=dims, activation="dice", use_softmax=sm)
    au.load_state_dict({n[len(k + "sd0."):]: torch.from_numpy(g[n]) for n in g.files if n.startswith(k + "sd0.")})
    au = au.to(dev()).train()
    assert ops.din_att_l1_ok(hist, tgt, au.attention.mlp[0]), "fixture must reach the fused first layer"
    calls = []
    real = ops.din_att_l1
    monkeypatch.setattr(ops, "din_att_l1", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    out = au(hist, tgt)
    out.backward(torch.from_numpy(g[k + "G"]).to(dev()))
    torch.cuda.synchronize()
    ops.check_errors()
    assert calls, "the fused first layer (csrc/dinmlp.hip) did not run"
    close(out, g[k + "out"], rtol=1e-5, atol_scale=2e-6, what=f"{tag}: attention output")
    close(hist.grad, g[k + "g_hist"], rtol=1e-4, atol_scale=1e-5, what=f"{tag}: g_history")
    close(tgt.grad, g[k + "g_tgt"], rtol=1e-4, atol_scale=1e-5, what=f"{tag}: g_target")
    # a Linear bias in front of BatchNorm has the exact gradient 0 (rounding noise on both sides): every parameter is
    # compared on the scale of the largest gradient of the unit
    gmax = max(float(np.abs(g[k + "grad." + n]).max()) for n, _ in au.named_parameters())
    for n, p in au.named_parameters():
        want = g[k + "grad." + n]
        got = p.grad.detach().cpu().numpy()
        assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max() + 2e-6 * gmax, f"{tag}: grad of {n}"
    for n, b in au.named_buffers():
        want = g[k + "sd1." + n]
        if n.endswith("num_batches_tracked"):
            assert int(b) == int(want)
        else:
            close(b, want, rtol=1e-5, atol_scale=1e-6, what=f"{tag}: buffer {n}")


def test_din_att_l1_matches_linear_on_materialised_operand_and_chunk_statistics():
    from torch_rechub_amd import _lib, ops
    g = torch.Generator().manual_seed(3)
    B, L, D, N = 130, 100, 16, 256
    hist = torch.randn(B, L, D, generator=g).to(dev())
    tgt = torch.randn(B, D, generator=g).to(dev())
    W = (torch.randn(N, 4 * D, generator=g) * 0.2).to(dev())
    b = torch.randn(N, generator=g).to(dev())
    z, part = ops.din_att_l1(hist, tgt, W, b, True)
    t = tgt.unsqueeze(1).expand(-1, L, -1)
    att = torch.cat([t, hist, t - hist, t * hist], dim=-1).view(-1, 4 * D)
    want = torch.nn.functional.linear(att.double(), W.double(), b.double())
    assert float((z.double() - want).abs().max()) <= 2e-5
    rows = _lib.call("rh_din_att_l1_chunk_rows", B * L)
    assert part.shape == (-(-(B * L) // rows), 2, N)
    for k in range(part.shape[0]):  # every chunk's (sum, M2) against float64 on the same rows
        blk = want[k * rows:(k + 1) * rows]
        assert torch.allclose(part[k, 0].double(), blk.sum(0), rtol=1e-5, atol=1e-3)
        assert torch.allclose(part[k, 1].double(), ((blk - blk.mean(0)) ** 2).sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("shape", [(4096, 512), (1000, 257), (3,), (1, 1), (7, 5, 3)])
def test_prelu_single_slope_vs_torch(shape):
    """nn.PReLU() (activation_layer("prelu"): the DSSM towers) through rh_prelu_fwd / rh_prelu_bwd against ATen on the
    device: outputs and input gradient bit-equal (same per-element arithmetic), slope gradient up to summation order."""
    from torch_rechub_amd import ops
    g = torch.Generator().manual_seed(len(shape))
    x = torch.randn(*shape, generator=g).to(dev())
    x.view(-1)[0] = 0.0  # the kink: forward 0, backward takes the slope branch
    gy = torch.randn(*shape, generator=g).to(dev())
    mod = torch.nn.PReLU(init=0.3).to(dev())
    assert ops.prelu_ok(mod, x)
    x1 = x.clone().requires_grad_()
    y1 = ops.prelu(x1, mod.weight)
    y1.backward(gy)
    g_slope, mod.weight.grad = mod.weight.grad.clone(), None
    x2 = x.clone().requires_grad_()
    y2 = mod(x2)
    y2.backward(gy)
    assert torch.equal(y1, y2) and torch.equal(x1.grad, x2.grad)
    assert torch.allclose(g_slope, mod.weight.grad, rtol=1e-4, atol=1e-4 * max(1.0, float(mod.weight.grad.abs())))


@pytest.mark.parametrize("B,K,p", [(4096, 128, 0.2), (1000, 64, 0.0), (77, 256, 0.5), (8192, 128, 0.0)])
def test_head_backward_forms_the_batchnorm_sums_of_the_layer_below(B, K, p, monkeypatch):
    """Linear -> BatchNorm1d -> ReLU -> Dropout -> Linear(K, 1) + sigmoid head: with the head as the only consumer of the hidden
    layer its backward also emits the BatchNorm-backward column sums (rh_head_bwd_bn + rh_bn_relu_dropout_bwd_pre); every
    gradient must agree with the separate statistics launch (ops.FUSE_HEAD_BN = False) -- same mask, same arithmetic per element,
    only the summation partition differs."""
    from torch_rechub_amd import ops
    from torch_rechub_amd.basic.layers import MLP
    g = torch.Generator().manual_seed(B + K)
    x = torch.randn(B, 40, generator=g).to(dev())
    gy = torch.randn(B, generator=g).to(dev())
    grads = {}
    rng = dev() if isinstance(dev(), torch.device) else torch.device(dev())
    monkeypatch.setattr(ops, "FUSE_MLP_CHAIN", False)  # this test is about the layer-by-layer kernels
    for flag in (True, False):
        monkeypatch.setattr(ops, "FUSE_HEAD_BN", flag)
        torch.manual_seed(5)
        mlp = MLP(40, output_layer=True, dims=[K], dropout=p, activation="relu").to(dev()).train()
        ops._dropout_rng(rng).copy_(torch.tensor([1234, 0, 0, 0], device=rng))  # same dropout stream for both runs
        xi = x.clone().requires_grad_()
        y = mlp.sigmoid_head(xi)
        y.backward(gy)
        grads[flag] = [xi.grad.clone()] + [q.grad.clone() for q in mlp.parameters()]
    torch.cuda.synchronize()
    for a, b in zip(grads[True], grads[False]):
        scale = max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 1e-5 * scale + 1e-4 * float(b.abs().max())


@pytest.mark.parametrize("B,K0,dims,p", [(4096, 429, [256, 128], 0.2), (4096, 429, [256, 128], 0.0), (1000, 40, [64], 0.3),
                                         (77, 33, [128, 64, 32], 0.5), (2, 16, [8, 4], 0.0), (3000, 432, [1024, 256], 0.1)])
def test_mlp_chain_equals_the_layer_by_layer_kernels(B, K0, dims, p, monkeypatch):
    """ops._MlpChainFn (round 4: BatchNorm + ReLU + Dropout of a hidden layer applied on the operand load of the next GEMM /
    of the head, BatchNorm-backward sums from the head's backward / the input-gradient GEMM's epilogue; csrc/gemm.hip PRO /
    BNBWD, csrc/linear.hip head_bnact_fwd_kernel) against the layer-by-layer path it replaces (library GEMM -> rh_bn_relu_
    dropout_fwd -> ... -> rh_head_fwd and their backwards), reference torch_rechub/basic/layers.py:276-292 + deepfm.py:39-43:
    same dropout stream (same counters in the same order), so predictions, every gradient, the running statistics and
    num_batches_tracked must agree up to the summation order of the GEMMs and of the column sums."""
    from torch_rechub_amd import ops
    from torch_rechub_amd.basic.layers import MLP
    g = torch.Generator().manual_seed(B + K0)
    pitch = (K0 + 15) // 16 * 16
    x = torch.randn(B, pitch, generator=g).to(dev())[:, :K0]  # a row-padded view, as the fused gather hands over
    e0, e1 = torch.randn(B, 1, generator=g).to(dev()), torch.randn(B, generator=g).to(dev())
    gy = torch.randn(B, generator=g).to(dev())
    rng = dev() if isinstance(dev(), torch.device) else torch.device(dev())
    out, calls = {}, []
    real = ops.mlp_chain_sigmoid
    monkeypatch.setattr(ops, "mlp_chain_sigmoid", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    for flag in (True, False):
        monkeypatch.setattr(ops, "FUSE_MLP_CHAIN", flag)
        torch.manual_seed(5)
        mlp = MLP(K0, output_layer=True, dims=dims, dropout=p, activation="relu").to(dev()).train()
        with torch.no_grad():
            for m in mlp.mlp:
                if isinstance(m, torch.nn.BatchNorm1d):  # non-trivial affine, so that gamma / beta matter
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.uniform_(-0.5, 0.5)
        ops._dropout_rng(rng).copy_(torch.tensor([1234, 0, 0, 0], device=rng))  # same dropout stream for both runs
        xi, a0, a1 = x.clone().requires_grad_(), e0.clone().requires_grad_(), e1.clone().requires_grad_()
        for _ in range(2):  # two steps: the running statistics / counters advance the same way
            y = mlp.sigmoid_head(xi, a0, a1)
            y.backward(gy)
        ops.check_errors()
        out[flag] = dict(y=y.detach().clone(), gx=xi.grad.clone(), g0=a0.grad.clone(), g1=a1.grad.clone(),
                         params={n: q.grad.clone() for n, q in mlp.named_parameters()},
                         bufs={n: b_.clone() for n, b_ in mlp.named_buffers()})
    torch.cuda.synchronize()
    assert len(calls) == 2, "the fused chain did not run"
    A, R = out[True], out[False]
    np.testing.assert_allclose(A["y"].cpu().numpy(), R["y"].cpu().numpy(), rtol=1e-5, atol=2e-6)
    for k in ("gx", "g0", "g1"):
        scale = max(1e-6, float(R[k].abs().max()))
        assert float((A[k] - R[k]).abs().max()) <= 2e-5 * scale + 2e-8, k  # (+ an absolute floor: B = 2 gradients are ~1e-4)
    for n in R["params"]:
        scale = max(1e-3, float(R["params"][n].abs().max()))
        # (a Linear bias in front of BatchNorm has a mathematically zero gradient: rounding noise on both sides)
        noise = n.endswith(".bias") and any(n == f"mlp.{4 * i}.bias" for i in range(len(dims)))
        tol = 1e-4 * max(scale, float(R["gx"].abs().max()) * B ** 0.5) if noise else 2e-5 * scale
        assert float((A["params"][n] - R["params"][n]).abs().max()) <= tol, n
    for n in R["bufs"]:
        if n.endswith("num_batches_tracked"):
            assert int(A["bufs"][n]) == int(R["bufs"][n]) == 2, n
        else:
            np.testing.assert_allclose(A["bufs"][n].cpu().numpy(), R["bufs"][n].cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=n)


@pytest.mark.parametrize("fused_loss", [False, True])
def test_hidden_layer_and_prediction_with_two_consumers_each(fused_loss):
    """The head -> BatchNorm shortcut (rh_head_bwd_bn hands its column sums to the layer below) and the fused BCE (the
    loss gradient is formed inside the head's backward) both have to notice when the tensor they produced is NOT the only
    gradient: here the hidden activation h AND the prediction y each feed a second consumer.  Reference = the same
    modules in float64 on eager torch (what the reference's op chain computes)."""
    from torch_rechub_amd import ops
    from torch_rechub_amd.basic.layers import MLP
    B, K = 256, 64
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, 40, generator=g)
    t = (torch.rand(B, generator=g) < 0.3).float()
    c = torch.randn(B, K, generator=g)  # weights of the second consumer of h
    torch.manual_seed(9)
    mlp = MLP(40, output_layer=True, dims=[K], dropout=0.0, activation="relu").train()
    ref = MLP.__new__(MLP)
    torch.nn.Module.__init__(ref)
    import copy
    ref.mlp = copy.deepcopy(mlp.mlp).double()

    def run(net, xi, ti, ci, fused):
        mods = list(net.mlp)
        if fused is None:  # plain module chain (float64 reference)
            h = xi
            for m in mods[:-1]:
                h = m(h)
            y = torch.sigmoid(mods[-1](h).squeeze(1))
            loss = torch.nn.functional.binary_cross_entropy(y, ti)
        else:
            if fused:
                ops.fusion_begin(target=ti)
            h = net._run(mods[:-1], xi)
            assert ops.head_ok(h, mods[-1], ())
            y = ops.head_sigmoid(h, mods[-1].weight, mods[-1].bias)
            loss = ops.bce_mean(y, ti)
            if fused:
                assert type(loss.grad_fn).__name__ == "_FusedBceFnBackward"
                ops.fusion_end()
        total = loss + 0.37 * (h * ci).sum() / h.shape[0] + 0.11 * (y * y).sum()
        total.backward()
        return total

    xr = x.double().requires_grad_()
    want = run(ref, xr, t.double(), c.double(), None)
    mlp = mlp.to(dev())
    xd = x.to(dev()).requires_grad_()
    got = run(mlp, xd, t.to(dev()), c.to(dev()), fused_loss)
    torch.cuda.synchronize()
    ops.check_errors()
    assert abs(got.item() - want.item()) < 1e-5 * max(1.0, abs(want.item()))
    close(xd.grad, xr.grad.numpy(), rtol=1e-4, atol_scale=1e-5, what="g_x")
    for (n, p), (_, q) in zip(mlp.named_parameters(), ref.named_parameters()):
        if n == "mlp.0.bias":  # in front of BatchNorm: exact gradient 0, rounding noise on both sides
            continue
        close(p.grad, q.grad.numpy(), rtol=1e-4, atol_scale=1e-5, what=f"g_{n}")


@pytest.mark.parametrize("B,dims,p", [(4096, [256, 128, 64], 0.0), (37, [33], 0.0), (9000, [64, 32], 0.0), (300, [48], 0.3),
                                      (9000, [64], 0.5)])
def test_mlp_batchnorm_prelu_dropout_epilogue(B, dims, p, monkeypatch):
    """Linear -> BatchNorm1d -> nn.PReLU() -> Dropout of the two-tower MLPs (reference MLP(activation="prelu"),
    basic/layers.py:276-292) through the fused epilogue (rh_bn_prelu_dropout_fwd/bwd: one launch pair per direction, the
    slope gradient from the statistics pass).  p = 0: against the same modules in float64 on eager torch -- outputs, input
    gradient, every parameter gradient (slopes included) and the running statistics.  p > 0 (one layer): the kernel's own
    counter-based mask is read off the output (zeros), its rate is checked against p, and the float64 reference applies
    exactly that mask."""
    import copy
    from torch_rechub_amd import ops
    from torch_rechub_amd.basic.layers import MLP
    g = torch.Generator().manual_seed(B + len(dims))
    x = torch.randn(B, 40, generator=g)
    gy = torch.randn(B, dims[-1], generator=g)
    torch.manual_seed(3)
    mlp = MLP(40, output_layer=False, dims=dims, dropout=p, activation="prelu").train()
    with torch.no_grad():
        for m in mlp.mlp:
            if isinstance(m, torch.nn.PReLU):
                m.weight.uniform_(-0.4, 0.6)
    calls = []
    real = ops.bn_prelu_dropout
    monkeypatch.setattr(ops, "bn_prelu_dropout", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    if p == 0.0:
        ref = copy.deepcopy(mlp).double()
        xr = x.double().requires_grad_()
        yr = ref(xr)
        yr.backward(gy.double())
    else:
        ref = copy.deepcopy(mlp).double()  # pristine copy: running statistics not yet updated
        ref.mlp[3].p = 0.0               # the reference applies the kernel's own mask (read off the output below)
    mlp = mlp.to(dev())
    xd = x.to(dev()).requires_grad_()
    y = mlp(xd)
    y.backward(gy.to(dev()))
    torch.cuda.synchronize()
    if p > 0.0:
        keep = (y.detach() != 0).double().cpu()  # the kernel's mask (an exact zero of prelu(bn) has measure zero)
        rate = 1.0 - float(keep.mean())
        assert abs(rate - p) < 4 * (p * (1 - p) / keep.numel()) ** 0.5 + 1e-3, f"dropout rate {rate} vs p = {p}"
        xr = x.double().requires_grad_()
        yr = ref(xr) * keep / (1.0 - p)
        yr.backward(gy.double())
    assert len(calls) == len(dims), "the fused BatchNorm + PReLU epilogue did not run"
    close(y, yr.detach().cpu().numpy(), rtol=1e-4, atol_scale=1e-5, what="outputs")
    close(xd.grad, xr.grad.detach().cpu().numpy(), rtol=2e-4, atol_scale=2e-5, what="g_x")
    gmax = max(float(q.grad.abs().max()) for q in ref.parameters())
    for (n, a_), (_, b_) in zip(mlp.named_parameters(), ref.named_parameters()):
        if n.endswith("bias") and "mlp." in n and isinstance(mlp.mlp[int(n.split(".")[1])], torch.nn.Linear):
            continue  # a Linear bias in front of BatchNorm: exact gradient 0, rounding noise on both sides
        got, want = a_.grad.detach().cpu().numpy(), b_.grad.detach().cpu().numpy()
        assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max() + 2e-5 * gmax, f"grad of {n}"
    for (n, a_), (_, b_) in zip(mlp.named_buffers(), ref.named_buffers()):
        if not n.endswith("num_batches_tracked"):
            close(a_, b_.detach().cpu().numpy(), rtol=1e-4, atol_scale=1e-5, what=f"buffer {n}")


@pytest.mark.parametrize("B,d", [(4096, 64), (37, 16), (5, 100), (300, 1024), (1, 4)])
def test_l2_normalize_matches_functional_normalize(B, d):
    """ops.l2_normalize (csrc/match.hip) == F.normalize(x, p=2, dim=1) of the reference towers (models/matching/dssm.py:56,66)
    in float64: values and gradient, including rows whose norm is below eps (clamped denominator) and a strided input."""
    from torch_rechub_amd import ops
    g = torch.Generator().manual_seed(B + d)
    big = torch.randn(B, d + 4, generator=g)
    big[::7] *= 1e-3
    if B > 2:
        big[1] = 0.0          # zero row: y = 0, gradient g / eps
        big[2] = 1e-14        # norm below eps
    gy = torch.randn(B, d, generator=g).to(dev())
    xs = big.to(dev())[:, :d].detach().requires_grad_()  # a view: row stride d + 4
    assert ops.l2_normalize_ok(xs)
    y = ops.l2_normalize(xs)
    y.backward(gy)
    xr = big[:, :d].double().requires_grad_()
    yr = torch.nn.functional.normalize(xr, p=2, dim=1)
    yr.backward(gy.double().cpu())
    close(y, yr.detach().numpy(), rtol=1e-5, atol_scale=1e-6, what="normalised rows")
    live = (big[:, :d].double().norm(dim=1) > 1e-12).numpy()
    got, want = xs.grad.cpu().numpy(), xr.grad.numpy()
    np.testing.assert_allclose(got[live], want[live], rtol=2e-5, atol=2e-6 * max(1.0, float(np.abs(want[live]).max()) if live.any() else 1.0))
    if (~live).any():  # clamped rows: g / eps (1e12-scale numbers)
        np.testing.assert_allclose(got[~live], want[~live], rtol=1e-5)


@pytest.mark.parametrize("B,C,with_target", [(4096, 21, False), (37, 5, True), (1, 1, False), (300, 64, True), (513, 130, True)])
def test_cross_entropy_mean_matches_torch(B, C, with_target):
    """ops.cross_entropy_mean == torch.nn.CrossEntropyLoss()(logits, target) in float64 (target None = the trainer's zero
    targets, trainers/match_trainer.py:136): loss and gradient, large logits included (temperature 0.02 scales them by 50)."""
    from torch_rechub_amd import ops
    g = torch.Generator().manual_seed(B * 3 + C)
    logits = (torch.randn(B, C, generator=g) * 20).to(dev()).requires_grad_()
    target = torch.randint(0, C, (B,), generator=g) if with_target else None
    crit = torch.nn.CrossEntropyLoss()
    assert ops.cross_entropy_ok(crit, logits, None if target is None else target.to(dev()))
    loss = ops.cross_entropy_mean(logits, None if target is None else target.to(dev()))
    (loss * 1.7).backward()
    lr = logits.detach().double().cpu().requires_grad_()
    ref = crit(lr, target if target is not None else torch.zeros(B, dtype=torch.long))
    (ref * 1.7).backward()
    ops.check_errors()
    assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    close(logits.grad, lr.grad.numpy(), rtol=1e-5, atol_scale=1e-7, what="g_logits")


@pytest.mark.parametrize("B,C,row0,D,K", [(4096, 4096, 0, 64, 20), (300, 900, 300, 16, 7), (5, 5, 0, 100, 4),
                                           (64, 64, 0, 300, 1), (33, 40, 7, 8, 0)])
def test_inbatch_logits_without_the_score_matrix(B, C, row0, D, K):
    """ops.inbatch_logits (csrc/match.hip) == gather_inbatch_logits(user @ item.T, neg_indices) of the reference
    (utils/match.py:148-153), values and both gradients; rows of a rank's slice of a global batch (row0, C > B) included."""
    from torch_rechub_amd import ops
    from torch_rechub_amd.utils.match import gather_inbatch_logits
    g = torch.Generator().manual_seed(B + D)
    u = torch.randn(B, D, generator=g).to(dev())
    v = torch.randn(C, D, generator=g).to(dev())
    neg = torch.randint(0, C, (B, K), generator=g).to(dev())
    gl = torch.randn(B, 1 + K, generator=g).to(dev())
    assert ops.inbatch_logits_ok(u, v)
    u1, v1 = u.clone().requires_grad_(), v.clone().requires_grad_()
    out = ops.inbatch_logits(u1, v1, neg, row0)
    out.backward(gl)
    u2, v2 = u.double().requires_grad_(), v.double().requires_grad_()
    ref = gather_inbatch_logits(u2 @ v2.t(), neg, row_offset=row0)
    ref.backward(gl.double())
    torch.cuda.synchronize()
    ops.check_errors()
    for a, b, what in ((out, ref, "logits"), (u1.grad, u2.grad, "g_user"), (v1.grad, v2.grad, "g_item")):
        scale = max(1.0, float(b.abs().max()))
        assert float((a.double() - b).abs().max()) <= 2e-5 * scale, what


def test_sweep_gate_is_released_by_a_chain_start_or_by_its_fallback():
    """rh_adam_sweep_gate (round 5): behind opening number i a gate waits for a CHAIN START counted after that opening
    (rh_linear_fwd_gate: the last workgroup of the step's first own GEMM) -- and only without one for `fallback_ns`.  Round 4
    released the deferred sweep a fixed wall-clock time behind the opening (a 20 % cliff +-6 us).
    Measured on the side stream with HIP events: (a) opening + chain start -> released at once; (b) opening alone -> released
    after the fallback; (c) a chain start counted BEFORE the opening does not release it; (d) no opening at all -> the 2 s
    timeout raises RH_ERR_GATE_TIMEOUT instead of wedging the queue (not exercised here: it would cost the suite 2 s)."""
    from torch_rechub_amd import _lib, ops
    gate = torch.zeros(16, dtype=torch.int64, device=dev())
    err = ops.err_flag(dev())
    M, N, K = 256, 64, 64
    x, w, b = torch.randn(M, K, device=dev()), torch.randn(N, K, device=dev()), torch.zeros(N, device=dev())
    y = torch.empty(M, N, device=dev())
    side = torch.cuda.Stream()
    nul = torch.empty(0, device=dev())

    def chain_start():
        _lib.call("rh_linear_fwd_gate", ops._p(x), K, ops._p(w), K, ops._p(b), M, N, K, ops._p(y), N, 0, 0, 0, 0,
                  ops._p(gate), ops._stream())

    def gated_ms(expected, fallback_ns, between):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            e0.record()
            _lib.call("rh_adam_sweep_gate", ops._p(gate), expected, fallback_ns, ops._p(err), ops._stream())
            e1.record()
        between()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    def opening_then_chain():
        _lib.call("rh_adam_sweep_gate_open", ops._p(gate), ops._stream())
        chain_start()

    # (a) opening 1, then a chain start: released long before the 50 ms fallback
    assert gated_ms(1, 50_000_000, opening_then_chain) < 10.0
    assert gate[0].item() == 1 and gate[2].item() == 1
    np.testing.assert_allclose(y.cpu().numpy(), (x @ w.t()).cpu().numpy(), rtol=1e-4, atol=1e-4)  # (the GEMM is still a GEMM)
    # (c) + (b) a chain start BEFORE opening 2 does not count for it: the gate falls back after ~30 ms
    chain_start()
    torch.cuda.synchronize()
    t = gated_ms(2, 30_000_000, lambda: _lib.call("rh_adam_sweep_gate_open", ops._p(gate), ops._stream()))
    assert 25.0 < t < 200.0, t
    assert gate[0].item() == 2 and gate[2].item() == 2 and gate[4 + 2 * (2 & 3)].item() == 2
    # a gate reached late (its opening and chain start long past) adds no delay
    opening_then_chain()
    torch.cuda.synchronize()
    assert gated_ms(3, 50_000_000, lambda: None) < 10.0
    assert not (int(err.item()) & 64)


@pytest.mark.parametrize("B,d,L", [(64, 64, 2), (4096, 429, 3), (100, 30, 1), (2, 16, 2)])
def test_cross_v2_layer_on_the_tile_gemm_vs_float64_formula(B, d, L):
    """CrossNetV2.forward x <- x0 * (W_l x) + b_l + x (torch_rechub/basic/layers.py:440-444) as ONE launch per layer and
    direction (rh_cross_v2_fwd: tile GEMM + Hadamard / bias / residual epilogue; rh_cross_v2_dgrad: g_y W + g): output and
    every gradient against the formula in float64, and the same bits from the torch.library op."""
    from torch_rechub_amd import ops
    from torch_rechub_amd.basic.layers import CrossNetV2
    import torch_rechub_amd.library  # noqa: F401
    g = torch.Generator().manual_seed(B + d)
    net = CrossNetV2(d, L).to(dev())
    with torch.no_grad():
        for l in range(L):
            net.w[l].weight.copy_(torch.randn(d, d, generator=g) / d ** 0.5)
            net.b[l].copy_(torch.randn(d, generator=g) * 0.1)
    x = torch.randn(B, d, generator=g)
    assert ops.cross_v2_layer_ok(x.to(dev()), net.w[0])
    xa = x.to(dev()).requires_grad_(True)
    out = net(xa)
    up = torch.randn(B, d, generator=g)
    (out * up.to(dev())).sum().backward()
    xr = x.double().requires_grad_(True)
    Ws = [net.w[l].weight.detach().cpu().double().requires_grad_(True) for l in range(L)]
    bs = [net.b[l].detach().cpu().double().requires_grad_(True) for l in range(L)]
    xl = xr
    for l in range(L):
        xl = xr * (xl @ Ws[l].t()) + bs[l] + xl
    (xl * up.double()).sum().backward()
    close(out, xl.detach().numpy(), rtol=1e-4, atol_scale=1e-5, what="cross_v2 out")
    close(xa.grad, xr.grad.numpy(), rtol=2e-4, atol_scale=2e-5, what="cross_v2 g_x")
    for l in range(L):
        close(net.w[l].weight.grad, Ws[l].grad.numpy(), rtol=2e-4, atol_scale=2e-5, what=f"cross_v2 g_W[{l}]")
        close(net.b[l].grad, bs[l].grad.numpy(), rtol=2e-4, atol_scale=2e-5, what=f"cross_v2 g_b[{l}]")
    W = torch.stack([net.w[l].weight.detach() for l in range(L)])
    bb = torch.stack([net.b[l].detach() for l in range(L)])
    assert torch.equal(torch.ops.rechub_hip.cross_net_v2(x.to(dev()), W, bb), out.detach())

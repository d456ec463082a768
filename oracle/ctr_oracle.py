"""CPU oracle (numpy) for the CTR hot path of datawhalechina/torch-rechub v0.8.0.

TEST INFRASTRUCTURE ONLY.  Nothing under ``torch_rechub_amd/`` may import this package; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, as the checker.

Every function restates one reference op chain (citations are ``path:line`` under /root/reference) in
plain numpy, forward AND hand-derived backward, in float64 by default so that it can serve as the
"exact" side of an fp32 tolerance test.  Pinning: ``tests/test_oracle_golden.py`` checks every function
against golden vectors produced by the UNMODIFIED reference (``oracle/gen_golden.py`` imports
/root/reference and dumps ``tests/golden/*.npz``), and, when /root/reference is present, against the live
reference on fresh random inputs.  The reference's own tests hold no golden vectors for these ops
(SURVEY 8c), so the generated fixtures are the pin.
"""
import numpy as np


# ---------------------------------------------------------------------------------------------------
# EmbeddingLayer  torch_rechub/basic/layers.py:77-127
# ---------------------------------------------------------------------------------------------------
def embedding_gather(tables, idx):
    """Per-field row gather -> (B, F, D).  layers.py:83 ``embed_dict[name](x[name].long()).unsqueeze(1)`` + cat :110.

    tables: list of (vocab_f, D) arrays (entries may alias: shared_with, layers.py:85); idx: (B, F) ints.
    Bit-exact by construction (pure copy); out-of-range indices raise IndexError like nn.Embedding on CPU.
    """
    idx = np.asarray(idx)
    B, F = idx.shape
    D = tables[0].shape[1]
    out = np.empty((B, F, D), dtype=tables[0].dtype)
    for f in range(F):
        col = idx[:, f].astype(np.int64)
        if col.size and (col.min() < 0 or col.max() >= tables[f].shape[0]):
            raise IndexError("index out of range in self")
        out[:, f, :] = tables[f][col]
    return out


def embedding_layer_squeeze(tables, idx, dense=None):
    """``squeeze_dim=True`` layout: all sparse embeddings flattened first, dense values last (layers.py:112-120)."""
    emb = embedding_gather(tables, idx).reshape(idx.shape[0], -1)
    if dense is None or dense.shape[1] == 0:
        return emb
    return np.concatenate([emb, dense.astype(emb.dtype)], axis=1)


def embedding_backward(tables_shape, idx, g_rows, padding_idx=None, out=None):
    """Dense table gradients from per-lookup gradient rows (B, F, D): embedding_dense_backward semantics.

    Duplicates accumulate; rows equal to padding_idx[f] get no gradient (nn.Embedding(padding_idx), initializers.py:17).
    tables_shape: list of (vocab_f, D).  ``out``: optional list of arrays to accumulate into (shared tables alias).
    """
    idx = np.asarray(idx)
    B, F = idx.shape
    grads = out if out is not None else [np.zeros(s, dtype=g_rows.dtype) for s in tables_shape]
    for f in range(F):
        col = idx[:, f].astype(np.int64)
        g = g_rows[:, f, :]
        if padding_idx is not None and padding_idx[f] is not None and padding_idx[f] >= 0:
            keep = col != padding_idx[f]
            col, g = col[keep], g[keep]
        np.add.at(grads[f], col, g)
    return grads


# ---------------------------------------------------------------------------------------------------
# InputMask / pooling  layers.py:148-161, 204-251
# ---------------------------------------------------------------------------------------------------
def input_mask(idx, padding_idx=None):
    """(B, L) float mask: ``x != padding_idx`` or ``x != -1`` when padding_idx is None (layers.py:154-157)."""
    sentinel = -1 if padding_idx is None else padding_idx
    return (np.asarray(idx).astype(np.int64) != sentinel).astype(np.float64)


def seq_pool(table, idx, pooling, padding_idx=None):
    """Sequence feature: gather (B, L, D) then Sum/Average/ConcatPooling with the InputMask (layers.py:86-99)."""
    idx = np.asarray(idx).astype(np.int64)
    if idx.min() < 0 or idx.max() >= table.shape[0]:
        raise IndexError("index out of range in self")
    emb = table[idx]  # (B, L, D)
    if pooling == "concat":
        return emb  # layers.py:204-205 ignores the mask
    mask = input_mask(idx, padding_idx).astype(emb.dtype)  # (B, L)
    total = np.einsum("bl,bld->bd", mask, emb)  # bmm(mask, x), layers.py:251 / :227
    if pooling == "sum":
        return total
    if pooling == "mean":
        return total / (mask.sum(axis=1, keepdims=True) + 1e-16)  # layers.py:228-229
    raise ValueError("Sequence pooling method supports only pooling in %s, got %s." % (["sum", "mean"], pooling))


def seq_pool_backward(table_shape, idx, pooling, g_out, padding_idx=None):
    """Table gradient of seq_pool.  g_out: (B, D) for sum/mean, (B, L, D) for concat."""
    idx = np.asarray(idx).astype(np.int64)
    B, L = idx.shape
    grad = np.zeros(table_shape, dtype=g_out.dtype)
    if pooling == "concat":
        g_pos = g_out
    else:
        mask = input_mask(idx, padding_idx).astype(g_out.dtype)
        w = mask if pooling == "sum" else mask / (mask.sum(axis=1, keepdims=True) + 1e-16)
        g_pos = w[:, :, None] * g_out[:, None, :]
    flat_idx, flat_g = idx.reshape(-1), g_pos.reshape(B * L, -1)
    if padding_idx is not None:
        keep = flat_idx != padding_idx
        flat_idx, flat_g = flat_idx[keep], flat_g[keep]
    np.add.at(grad, flat_idx, flat_g)
    return grad


# ---------------------------------------------------------------------------------------------------
# FM  layers.py:313-319   /   LR  layers.py:185-189
# ---------------------------------------------------------------------------------------------------
def fm_forward(x, reduce_sum=True):
    """0.5 * [ (sum_f x)^2 - sum_f x^2 ], summed over the embed dim when reduce_sum (layers.py:314-319)."""
    square_of_sum = x.sum(axis=1)**2
    sum_of_square = (x**2).sum(axis=1)
    ix = square_of_sum - sum_of_square
    if reduce_sum:
        ix = ix.sum(axis=1, keepdims=True)
    return 0.5 * ix


def fm_backward(x, g_out, reduce_sum=True):
    """d fm / d x[b,f,d] = g * (sum_f' x[b,f',d] - x[b,f,d]);  g_out (B,1) if reduce_sum else (B,D)."""
    s = x.sum(axis=1, keepdims=True)
    g = g_out[:, None, :] if not reduce_sum else g_out.reshape(-1, 1, 1)
    return g * (s - x)


def lr_forward(x, w, b):
    """nn.Linear(input_dim, 1): x (B, n) @ w (1, n).T + b (1,) -> (B, 1)   (layers.py:183-189)."""
    return x @ w.reshape(1, -1).T + b.reshape(1, 1)


def deepfm_sparse_part(tables, idx, lr_w, lr_b, dense=None):
    """The fused unit of DeepFM.forward (models/ranking/deepfm.py:35-40): one gather feeding
    input_deep (B, F*D [+ n_dense]), y_fm (B,1) and y_linear (B,1)."""
    emb = embedding_gather(tables, idx)
    flat = emb.reshape(emb.shape[0], -1)
    deep_in = flat if dense is None else np.concatenate([flat, dense.astype(flat.dtype)], axis=1)
    return deep_in, fm_forward(emb, True), lr_forward(flat, lr_w, lr_b)


def deepfm_sparse_part_backward(tables, idx, lr_w, g_deep, g_fm, g_lr, padding_idx=None):
    """Backward of deepfm_sparse_part: per-lookup rows g = g_deep + g_lr*w + g_fm*(S - v).

    Returns (table grads list, lr_w grad (1, F*D), lr_b grad (1,), rows (B, F, D)).
    """
    emb = embedding_gather(tables, idx)
    B, F, D = emb.shape
    rows = np.zeros_like(emb)
    if g_deep is not None:
        rows += g_deep[:, :F * D].reshape(B, F, D)
    if g_lr is not None:
        rows += g_lr.reshape(B, 1, 1) * lr_w.reshape(1, F, D)
    if g_fm is not None:
        rows += fm_backward(emb, g_fm.reshape(B, 1), True)
    # shared tables (same array object) accumulate into one gradient
    uniq = {}
    grads = []
    for t in tables:
        if id(t) not in uniq:
            uniq[id(t)] = np.zeros(t.shape, dtype=rows.dtype)
        grads.append(uniq[id(t)])
    embedding_backward([t.shape for t in tables], idx, rows, padding_idx, out=grads)
    g_w = None if g_lr is None else (g_lr.reshape(B, 1) * emb.reshape(B, -1)).sum(axis=0, keepdims=True)
    g_b = None if g_lr is None else np.array([g_lr.sum()], dtype=rows.dtype)
    return grads, g_w, g_b, rows


# ---------------------------------------------------------------------------------------------------
# Cross networks  layers.py:412-420 (CrossNetwork), :440-444 (CrossNetV2), :470-506 (CrossNetMix)
# ---------------------------------------------------------------------------------------------------
def cross_network_forward(x, W, Bv):
    """x0 = x; per layer: xw = x @ w_l (Linear(d,1,bias=False)); x = x0 * xw + b_l + x   (layers.py:416-420)."""
    x0 = x
    for l in range(W.shape[0]):
        xw = x @ W[l].reshape(-1, 1)
        x = x0 * xw + Bv[l] + x
    return x


def cross_network_backward(x, W, Bv, g_out):
    """Returns (g_x, g_W (L,d), g_B (L,d))."""
    L = W.shape[0]
    x0 = x
    xs, ss = [], []
    cur = x
    for l in range(L):
        xs.append(cur)
        s = cur @ W[l].reshape(-1, 1)
        ss.append(s)
        cur = x0 * s + Bv[l] + cur
    G = g_out.copy()
    g_x0 = np.zeros_like(x)
    gW, gB = np.zeros_like(W), np.zeros_like(Bv)
    for l in reversed(range(L)):
        t = (G * x0).sum(axis=1, keepdims=True)
        gB[l] = G.sum(axis=0)
        gW[l] = (t * xs[l]).sum(axis=0)
        g_x0 += G * ss[l]
        G = G + t * W[l].reshape(1, -1)
    return G + g_x0, gW, gB


def cross_net_v2_forward(x, Ws, Bv):
    """x = x0 * (x @ W_l.T) + b_l + x with W_l (d,d) an nn.Linear weight (layers.py:440-444)."""
    x0 = x
    for l in range(len(Ws)):
        x = x0 * (x @ Ws[l].T) + Bv[l] + x
    return x


def cross_net_mix_forward(x, U, V, C, Wg, bias):
    """CrossNetMix (layers.py:470-506).  U, V: (L, E, d, r); C: (L, E, r, r); Wg: (E, d) gating weights (shared
    across layers); bias: (L, d).  Per layer: softmax over experts of gating_e(x_l); expert e:
    x0 * (U_e tanh(C_e tanh(V_e^T x_l)) + bias_l); x_{l+1} = sum_e gate_e * expert_e + x_l."""
    x0, xl = x, x
    L, E = U.shape[0], U.shape[1]
    for l in range(L):
        gate = xl @ Wg.T  # (B, E)
        gate = np.exp(gate - gate.max(axis=1, keepdims=True))
        gate = gate / gate.sum(axis=1, keepdims=True)
        outs = []
        for e in range(E):
            v = np.tanh(xl @ V[l, e])  # (B, r) == (V^T x)^T
            v = np.tanh(v @ C[l, e].T)
            uv = v @ U[l, e].T  # (B, d)
            outs.append(x0 * (uv + bias[l]))
        moe = sum(outs[e] * gate[:, e:e + 1] for e in range(E))
        xl = moe + xl
    return xl


# ---------------------------------------------------------------------------------------------------
# Dice  basic/activation.py:15-25   /  MLP pieces  layers.py:276-292
# ---------------------------------------------------------------------------------------------------
def dice_forward(x, alpha, eps=1e-3):
    """Row-wise (dim=1) mean; "var" = SUM over the row of (x-mean)^2 + eps (Q5); p*x + (1-p)*alpha*x."""
    avg = x.mean(axis=1, keepdims=True)
    var = ((x - avg)**2 + eps).sum(axis=1, keepdims=True)
    ps = 1.0 / (1.0 + np.exp(-(x - avg) / np.sqrt(var)))
    return ps * x + (1 - ps) * alpha * x


def batchnorm1d_train(x, weight, bias, eps=1e-5):
    """nn.BatchNorm1d in training mode: biased batch variance for normalisation."""
    mu = x.mean(axis=0, keepdims=True)
    var = x.var(axis=0, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * weight + bias


def batchnorm1d_eval(x, weight, bias, running_mean, running_var, eps=1e-5):
    return (x - running_mean) / np.sqrt(running_var + eps) * weight + bias


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def bce_loss(p, y):
    """torch.nn.BCELoss(reduction='mean') on probabilities, log clamped at -100 (trainers/ctr_trainer.py:68,88)."""
    lp = np.maximum(np.log(p), -100.0)
    l1p = np.maximum(np.log1p(-p), -100.0)
    return float(-(y * lp + (1 - y) * l1p).mean())


# ---------------------------------------------------------------------------------------------------
# DIN attention  models/ranking/din.py:77-92
# ---------------------------------------------------------------------------------------------------
def activation_unit_forward(history, target, mlp_fn, use_softmax=False):
    """att = MLP([t, h, t-h, t*h]) over B*L rows (no padding mask, Q6); out = sum_l att_l * h_l.

    mlp_fn maps a (B*L, 4D) array to (B*L, 1) (the attention MLP incl. BatchNorm over the B*L rows)."""
    B, L, D = history.shape
    t = np.broadcast_to(target[:, None, :], (B, L, D))
    att_in = np.concatenate([t, history, t - history, t * history], axis=-1)
    w = mlp_fn(att_in.reshape(-1, 4 * D)).reshape(B, L)
    if use_softmax:
        w = np.exp(w - w.max(axis=1, keepdims=True))
        w = w / w.sum(axis=1, keepdims=True)
    return (w[:, :, None] * history).sum(axis=1)


# ---------------------------------------------------------------------------------------------------
# torch.optim.Adam (coupled L2), dense over every row — trainers/ctr_trainer.py:59-61,99
# ---------------------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """One torch.optim.Adam step (amsgrad=False) on arrays; ``step`` is the 1-based step count AFTER increment.

    Mirrors torch/optim/adam.py::_single_tensor_adam: grad += wd*param; exp_avg.lerp_(grad, 1-b1);
    exp_avg_sq = b2*exp_avg_sq + (1-b2)*grad^2; denom = sqrt(exp_avg_sq)/sqrt(1-b2^t) + eps;
    param -= (lr/(1-b1^t)) * exp_avg/denom.  Returns new (p, m, v).
    """
    g = g + weight_decay * p
    m = m + (g - m) * (1 - beta1)
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1**step
    bc2 = 1 - beta2**step
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


# ---------------------------------------------------------------------------------------------------
# DataLoader batch assembly  utils/data.py:14-25 + default_collate
# ---------------------------------------------------------------------------------------------------
def batch_gather(perm, pos, B, sparse, dense, label):
    """Rows perm[(pos+b) % N] of the columnar dataset: what TorchDataset.__getitem__ + collate produce for a batch."""
    N = perm.shape[0]
    rows = perm[(pos + np.arange(B)) % N]
    return sparse[rows], (None if dense is None else dense[rows]), label[rows]


# ---------------------------------------------------------------------------------------------------
# Row-sharded tables: the lookup of layers.py:83-99 computed from per-rank shards (no reference code does this;
# the reference result it must equal is embedding_gather above on the full tables)
# ---------------------------------------------------------------------------------------------------
def shard_rows(table, world, rank):
    """Shard of a full (vocab, D) table: rows rank, rank + world, ... followed by zero rows up to ceil(vocab / world)
    and the all-zero sink row."""
    n = -(-table.shape[0] // world)
    local = np.zeros((n + 1, table.shape[1]), dtype=table.dtype)
    mine = table[rank::world]
    local[:mine.shape[0]] = mine
    return local


def shard_localize(idx, vocabs, pads, world, rank):
    """Index matrix (N, F) of the global batch -> int32 local rows of ``rank``'s shards: g // world when
    g % world == rank and g is not the field's padding_idx, else the sink row ceil(vocab / world).
    Out-of-range ids raise IndexError, like the nn.Embedding lookup they stand for."""
    idx = np.asarray(idx).astype(np.int64)
    out = np.empty(idx.shape, dtype=np.int32)
    for f in range(idx.shape[1]):
        g = idx[:, f]
        if g.size and (g.min() < 0 or g.max() >= vocabs[f]):
            raise IndexError("index out of range in self")
        sink = -(-vocabs[f] // world)
        own = (g % world == rank)
        if pads is not None and pads[f] is not None and pads[f] >= 0:
            own &= g != pads[f]
        out[:, f] = np.where(own, g // world, sink)
    return out


def sharded_embedding_gather(tables, idx, world, pads=None):
    """Sum over ranks of the gathers of their shards at the localised indices == embedding_gather(tables, idx) (with
    padding rows zero).  Exactly one term of every sum is non-zero, so the float sum is exact."""
    vocabs = [t.shape[0] for t in tables]
    total = None
    for r in range(world):
        loc = shard_localize(idx, vocabs, pads, world, r)
        part = embedding_gather([shard_rows(t, world, r) for t in tables], loc)
        total = part if total is None else total + part
    return total


# ---------------------------------------------------------------------------------------------------
# In-batch negative sampler of the HIP path (csrc/data.hip): the reference draws randperm(B-1)[:K] per row
# (utils/match.py:136-145) and pins only shape / no-self / distinctness / seed sensitivity (its tests); the kernel's
# stream (Floyd's subset algorithm over a counter hash) is restated here so that the device result is checked bit for bit.
# ---------------------------------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def _sample_hash(seed, ctr, idx):
    z = (idx * 0x9E3779B97F4A7C15 + seed + ctr * 0xD1B54A32D192ED03) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    z = z ^ (z >> 31)
    return z >> 32


def inbatch_sample_rows(seed, ctr, B, cols, row0, K):
    """(B, K): for global row r = row0 + i, K distinct columns of {0..cols-1} minus {r}."""
    out = np.empty((B, K), dtype=np.int64)
    n = cols - 1
    for i in range(B):
        own = row0 + i
        taken = set()
        for pos, j in enumerate(range(n - K, n)):
            t = _sample_hash(seed, ctr, own * K + pos) % (j + 1)
            pick = j if t in taken else t
            taken.add(pick)
            out[i, pos] = pick + (1 if pick >= own else 0)
    return out


# ---------------------------------------------------------------------------------------------------
# AUGRU  models/ranking/dien.py:17-66 (AUGRU_Cell.forward :30-36, the step loop of AUGRU.forward :60-66)
# ---------------------------------------------------------------------------------------------------
def augru_forward(xw, attn, U, state_bias=None):
    """h_all (B, T, D).  xw (B, T, 3D) = x_t [Wu|Wr|Wh] + [bu|br|bh]; attn (B, T); U (D, 3D) = [Uu|Ur|Uh]; h_0 = 0.
    Per step (dien.py:32-36): u = sig(xw_u + h Uu), r = sig(xw_r + h Ur), c = tanh(xw_h + r * (h Uh)),
    h' = (1 - a u) h + a u c.  ``state_bias`` (3D) is added to h U, ``attn`` None means a = 1: with those two the same
    recurrence is torch.nn.GRU's cell (update gate 1 - z), which DIEN's interest extractor uses (dien.py:101)."""
    B, T, D3 = xw.shape
    D = D3 // 3
    if attn is None:
        attn = np.ones((B, T), dtype=xw.dtype)
    if state_bias is None:
        state_bias = np.zeros(D3, dtype=xw.dtype)
    h = np.zeros((B, D), dtype=xw.dtype)
    out = np.empty((B, T, D), dtype=xw.dtype)
    for t in range(T):
        hu = h @ U + state_bias
        u = sigmoid(xw[:, t, :D] + hu[:, :D])
        r = sigmoid(xw[:, t, D:2 * D] + hu[:, D:2 * D])
        c = np.tanh(xw[:, t, 2 * D:] + r * hu[:, 2 * D:])
        g = attn[:, t:t + 1] * u
        h = (1 - g) * h + g * c
        out[:, t] = h
    return out


def augru_backward(xw, attn, U, g_hall, state_bias=None):
    """Hand-derived backward through time: returns (d_xw (B,T,3D), d_attn (B,T), d_U (D,3D)[, d_state_bias (3D)]) for
    upstream g_hall (B,T,D)."""
    B, T, D3 = xw.shape
    D = D3 // 3
    with_bias = state_bias is not None
    if attn is None:
        attn = np.ones((B, T), dtype=xw.dtype)
    if state_bias is None:
        state_bias = np.zeros(D3, dtype=xw.dtype)
    d_b = np.zeros(D3, dtype=xw.dtype)
    h_all = augru_forward(xw, attn, U, state_bias)
    d_xw = np.zeros_like(xw)
    d_attn = np.zeros_like(attn)
    d_U = np.zeros_like(U)
    dh = np.zeros((B, D), dtype=xw.dtype)
    for t in range(T - 1, -1, -1):
        dh = dh + g_hall[:, t]
        hp = h_all[:, t - 1] if t > 0 else np.zeros((B, D), dtype=xw.dtype)
        hu = hp @ U + state_bias
        u = sigmoid(xw[:, t, :D] + hu[:, :D])
        r = sigmoid(xw[:, t, D:2 * D] + hu[:, D:2 * D])
        q = hu[:, 2 * D:]
        c = np.tanh(xw[:, t, 2 * D:] + r * q)
        a = attn[:, t:t + 1]
        g = a * u
        dg = dh * (c - hp)
        dc = dh * g
        d_attn[:, t] = (dg * u).sum(1)
        dpu = dg * a * u * (1 - u)
        dpc = dc * (1 - c * c)
        dpr = dpc * q * r * (1 - r)
        dq = dpc * r
        d_xw[:, t] = np.concatenate([dpu, dpr, dpc], axis=1)
        d_hu = np.concatenate([dpu, dpr, dq], axis=1)
        d_U += hp.T @ d_hu
        d_b += d_hu.sum(0)
        dh = dh * (1 - g) + d_hu @ U.T
    return (d_xw, d_attn, d_U, d_b) if with_bias else (d_xw, d_attn, d_U)

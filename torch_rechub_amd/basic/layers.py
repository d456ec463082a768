"""Hot-path building blocks with the reference's constructor signatures, attribute names and
``state_dict`` keys (torch_rechub/basic/layers.py), executing on hand-written gfx950 kernels.

| class            | reference (basic/layers.py) | execution here                                        |
|------------------|-----------------------------|-------------------------------------------------------|
| EmbeddingLayer   | :33-127                     | ONE fused multi-field gather launch (ops.fused_embedding); sequence features: gather+pool kernel |
| InputMask        | :130-161                    | integer compare (API shim; the pooling kernel masks in-register) |
| Sum/Average/ConcatPooling | :192-251           | API shims for direct use; EmbeddingLayer fuses pooling into the gather |
| LR               | :164-189                    | nn.Linear (library GEMV); DeepFM fuses it into the gather kernel |
| MLP              | :254-292                    | nn.Linear / BatchNorm1d / activation / Dropout (hipBLASLt: true dense contraction) |
| FM               | :295-319                    | ops.fm kernel; DeepFM fuses it into the gather kernel |
| CrossNetwork     | :390-420                    | ops.cross_network: one wavefront per sample, all layers in registers |
| CrossNetV2       | :423-444                    | library GEMM + ONE fused Hadamard/bias/residual kernel (ops.cross_v2_epilogue) |
| CrossNetMix      | :447-506                    | experts batched into 3 GEMMs per layer (was 150 mm per step) + ONE fused bias/Hadamard/gate-mix/residual kernel |

Tables stay ``nn.Embedding`` modules inside ``embed_dict`` (checkpoint ABI:
``embedding.embed_dict.<feature>.weight``); kernels read them in place.
"""
import torch
from torch import nn

from .. import ops, sharding
from .activation import Dice, activation_layer
from .features import DenseFeature, SequenceFeature, SparseFeature


def _fusable_dim(d):
    q = d // 4
    return d % 4 == 0 and 1 <= q <= 32 and (q & (q - 1)) == 0


def _as_index(t):
    return t if t.dtype in (torch.int64, torch.int32) else t.long()


class PredictionLayer(nn.Module):
    """sigmoid for classification, identity for regression (reference layers.py:12-30)."""

    def __init__(self, task_type="classification"):
        super().__init__()
        if task_type not in ["classification", "regression"]:
            raise ValueError("task_type must be classification or regression")
        self.task_type = task_type

    def forward(self, x):
        return torch.sigmoid(x) if self.task_type == "classification" else x


class InputMask(nn.Module):
    """Float masks ``x != padding_idx`` (or ``!= -1`` when unset), one (B,1,...) slab per feature."""

    def forward(self, x, features):
        if not isinstance(features, list):
            features = [features]
        masks = []
        for fea in features:
            if not isinstance(fea, (SparseFeature, SequenceFeature)):
                raise ValueError("Only SparseFeature or SequenceFeature support to get mask.")
            sentinel = fea.padding_idx if fea.padding_idx is not None else -1
            masks.append((x[fea.name].long() != sentinel).unsqueeze(1).float())
        return torch.cat(masks, dim=1)


class ConcatPooling(nn.Module):
    """Identity on (B, L, D); the mask is ignored (reference layers.py:192-205)."""

    def forward(self, x, mask=None):
        return x


class SumPooling(nn.Module):
    """Masked sum over L: (B,1,L) x (B,L,D) -> (B,D) (reference layers.py:232-251)."""

    def forward(self, x, mask=None):
        if mask is None:
            return x.sum(dim=1)
        return (mask.transpose(1, 2) * x).sum(dim=1)


class AveragePooling(nn.Module):
    """Masked mean over L, denominator count + 1e-16 (reference layers.py:208-229)."""

    def forward(self, x, mask=None):
        if mask is None:
            return x.mean(dim=1)
        total = (mask.transpose(1, 2) * x).sum(dim=1)
        return total / (mask.sum(dim=-1).float() + 1e-16)


class EmbeddingLayer(nn.Module):
    """Multi-field embedding lookup; same contract as reference layers.py:33-127.

    forward(x, features, squeeze_dim=False):
      * dense only + squeeze_dim        -> (B, n_dense)
      * sparse, squeeze_dim=False       -> (B, n_features, D)   (or (B, n_seq, L, D) for concat pooling)
      * sparse (+dense), squeeze_dim    -> (B, sum(D) [+ n_dense]) with ALL sparse columns first, dense last (Q1)
    """

    def __init__(self, features):
        super().__init__()
        self.features = features
        self.embed_dict = nn.ModuleDict()
        self.n_dense = 0
        self.input_mask = InputMask()
        for fea in features:
            if fea.name in self.embed_dict:
                continue
            if isinstance(fea, (SparseFeature, SequenceFeature)) and fea.shared_with is None:
                self.embed_dict[fea.name] = fea.get_embedding_layer()
            elif isinstance(fea, DenseFeature):
                self.n_dense += 1

    def table_of(self, fea):
        return self.embed_dict[fea.name if fea.shared_with is None else fea.shared_with]

    def phys_dim(self, fea):
        """Row width the kernels see: ``embed_dim``, or the padded width of a PaddedEmbedding (initializers.py)."""
        return int(self.table_of(fea).weight.shape[1])

    _compact_cols = {}

    def compact(self, out, sparse_feas, n_tail=0):
        """(B, sum(phys) + n_tail) kernel output -> (B, sum(embed_dim) + n_tail): drops the zero padding columns of
        PaddedEmbedding tables (one index_select; identity when no table is padded)."""
        dims = tuple((self.phys_dim(f), f.embed_dim) for f in sparse_feas)
        if all(p == d for p, d in dims):
            return out
        key = (dims, n_tail, str(out.device))
        cols = EmbeddingLayer._compact_cols.get(key)
        if cols is None:
            if out.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("torch_rechub_amd: run the step eagerly once before capturing a hipGraph")
            idx, at = [], 0
            for p, d in dims:
                idx += list(range(at, at + d))
                at += p
            idx += list(range(at, at + n_tail))
            cols = torch.tensor(idx, dtype=torch.long, device=out.device)
            EmbeddingLayer._compact_cols[key] = cols
        return out.index_select(1, cols)

    def pad_lr_weight(self, lr_w, sparse_feas):
        """LR weight (1, sum(embed_dim)) -> (1, sum(phys)) with zeros under the padding columns (differentiable)."""
        if lr_w is None or all(self.phys_dim(f) == f.embed_dim for f in sparse_feas):
            return lr_w
        pieces, at = [], 0
        for f in sparse_feas:
            pieces.append(torch.nn.functional.pad(lr_w[:, at:at + f.embed_dim], (0, self.phys_dim(f) - f.embed_dim)))
            at += f.embed_dim
        return torch.cat(pieces, dim=1).contiguous()

    def fused(self, x, sparse_feas, dense_feas=(), lr_w=None, lr_b=None, want_fm=False):
        """The fused gather (+ FM, + LR, + dense append) on logical widths: (out, fm, lr)."""
        call = self.make_call(x, sparse_feas, dense_feas, want_fm=want_fm, want_lr=lr_w is not None)
        out, fm, lr = ops.fused_embedding(call, self.pad_lr_weight(lr_w, sparse_feas), lr_b)
        return self.compact(out, sparse_feas, len(call.dense)), fm, lr

    # -- helpers ---------------------------------------------------------------------------
    def _dense_columns(self, x, dense_feas):
        cols = []
        for fea in dense_feas:
            v = x[fea.name].float()
            cols.append(v if v.dim() > 1 else v.unsqueeze(1))
        return cols

    def make_call(self, x, sparse_feas, dense_feas=(), **kw):
        """EmbedCall for a list of plain sparse features of one fusable embed_dim (+ 1-D dense values)."""
        weights = [self.table_of(f).weight for f in sparse_feas]
        pads = [self.table_of(f).padding_idx for f in sparse_feas]
        idx = [_as_index(x[f.name]) for f in sparse_feas]
        dense = [x[f.name].float() for f in dense_feas]
        return ops.EmbedCall(weights, pads, idx, dense, **kw)

    def is_sharded(self, features):
        """True when a table behind ``features`` is row-sharded over the ranks (sharding.shard_tables)."""
        return any(sharding.is_sharded(self.table_of(f)) for f in features
                   if isinstance(f, (SparseFeature, SequenceFeature)))

    def can_fuse(self, x, features):
        """True when the whole list is one fused launch: plain sparse, one dim, 1-D dense values, replicated tables
        (a sharded lookup is gather -> reduce-scatter: FM / LR cannot ride in the gather kernel)."""
        dims = set()
        if self.is_sharded(features):
            return False
        for fea in features:
            if isinstance(fea, SparseFeature):
                dims.add(self.phys_dim(fea))
            elif isinstance(fea, SequenceFeature):
                return False
            elif x[fea.name].dim() != 1:
                return False
        return len(dims) == 1 and _fusable_dim(next(iter(dims)))

    def can_fuse_sharded(self, x, features):
        """True when the list is ONE row-sharded lookup: plain sparse features of one width whose tables are all
        sharded, 1-D dense values (appended by ops.fused_rows)."""
        dims = set()
        for fea in features:
            if isinstance(fea, SparseFeature):
                if not sharding.is_sharded(self.table_of(fea)):
                    return False
                dims.add(fea.embed_dim)
            elif isinstance(fea, SequenceFeature):
                return False
            elif x[fea.name].dim() != 1:
                return False
        return len(dims) == 1 and _fusable_dim(next(iter(dims)))

    def sharded_rows(self, x, sparse_feas):
        """(B, F*D) rows of row-sharded tables for the local batch (index all-gather, shard gather, reduce-scatter)."""
        return sharding.lookup([self.table_of(f) for f in sparse_feas], [_as_index(x[f.name]) for f in sparse_feas])

    def pieces(self, x, features):
        """The per-feature tensors of ``forward(x, features)`` as a list -- (B, D) per sparse feature, (B, L, D) per
        concat-pooled sequence feature -- for callers that take the (B, n, ...) result apart again feature by feature
        (DIN: din.py:40-47).  Saves the concatenation and, in the backward, a zero-filled (B, n, L, D) buffer, a copy and
        an add per slice."""
        out = self.forward(x, features, as_list=True)
        return list(out.unbind(1)) if torch.is_tensor(out) else out

    def forward(self, x, features, squeeze_dim=False, as_list=False):
        table_feas = [f for f in features if isinstance(f, (SparseFeature, SequenceFeature))]
        dense_feas = [f for f in features if not isinstance(f, (SparseFeature, SequenceFeature))]
        for fea in table_feas:
            if isinstance(fea, SequenceFeature) and fea.pooling not in ("sum", "mean", "concat"):
                raise ValueError("Sequence pooling method supports only pooling in %s, got %s." %
                                 (["sum", "mean"], fea.pooling))
        if not table_feas:
            if squeeze_dim and dense_feas:
                return torch.cat(self._dense_columns(x, dense_feas), dim=1)
            if squeeze_dim:
                raise ValueError("The input features can note be empty")
            raise ValueError("If keep the original shape:[batch_size, num_features, embed_dim], expected %s in feature "
                             "list, got %s" % ("SparseFeatures", features))

        if self.can_fuse(x, features):
            sparse = table_feas
            out, _, _ = self.fused(x, sparse, dense_feas if squeeze_dim else ())
            return out if squeeze_dim else out.reshape(out.shape[0], len(sparse), sparse[0].embed_dim)
        if self.can_fuse_sharded(x, features):
            rows = self.sharded_rows(x, table_feas)
            if squeeze_dim and dense_feas:  # dense values appended by one launch over the received rows (Q1)
                rows, _, _ = ops.fused_rows(rows, len(table_feas), [x[f.name].float() for f in dense_feas])
            return rows if squeeze_dim else rows.view(rows.shape[0], len(table_feas), -1)

        # general path: sequence features, mixed widths, row-sharded tables -> per-group launches, reference order
        pieces = [None] * len(table_feas)
        groups = {}
        for i, fea in enumerate(table_feas):
            if isinstance(fea, SparseFeature):
                groups.setdefault((self.phys_dim(fea), sharding.is_sharded(self.table_of(fea))), []).append(i)
        for (dim, sharded), members in groups.items():
            if not _fusable_dim(dim):
                raise RuntimeError(f"torch_rechub_amd: embed_dim={dim} has no HIP gather kernel "
                                   "(1 .. 128 supported, widths other than 4, 8, 16, 32, 64, 128 through padded storage); "
                                   "refusing to fall back to a CPU/eager path")
            feas = [table_feas[i] for i in members]
            if sharded:
                out = sharding.lookup([self.table_of(f) for f in feas], [_as_index(x[f.name]) for f in feas])
            else:
                out, _, _ = ops.fused_embedding(self.make_call(x, feas))
            for k, i in enumerate(members):
                pieces[i] = out[:, k * dim:k * dim + table_feas[i].embed_dim].unsqueeze(1)
        for i, fea in enumerate(table_feas):
            if isinstance(fea, SequenceFeature):
                if not _fusable_dim(self.phys_dim(fea)):
                    raise RuntimeError(f"torch_rechub_amd: sequence embed_dim={fea.embed_dim} has no HIP kernel (> 128)")
                table = self.table_of(fea)
                idx = _as_index(x[fea.name])
                if not sharding.is_sharded(table):
                    pooled = ops.seq_pool(table.weight, idx, fea.pooling, table.padding_idx)
                elif fea.pooling == "concat":  # (B, L, D): L lookups of one table per sample through the same exchange
                    L = idx.shape[1]
                    pooled = sharding.lookup([table] * L, [idx[:, j] for j in range(L)]).view(-1, L, fea.embed_dim)
                else:
                    pooled = sharding.pooled_lookup(table, idx, fea.pooling)
                if pooled.shape[-1] != fea.embed_dim:  # PaddedEmbedding: cut the zero padding columns off
                    pooled = pooled[..., :fea.embed_dim]
                pieces[i] = pooled.unsqueeze(1)
        if as_list and not squeeze_dim:
            return [p.squeeze(1) for p in pieces]
        sparse_emb = torch.cat(pieces, dim=1)
        if not squeeze_dim:
            return sparse_emb
        flat = sparse_emb.flatten(start_dim=1)
        if dense_feas:
            return torch.cat([flat] + self._dense_columns(x, dense_feas), dim=1)
        return flat


class LR(nn.Module):
    """Linear(input_dim, 1) with optional sigmoid (reference layers.py:164-189)."""

    def __init__(self, input_dim, sigmoid=False):
        super().__init__()
        self.sigmoid = sigmoid
        self.fc = nn.Linear(input_dim, 1, bias=True)

    def forward(self, x):
        y = self.fc(x)
        return torch.sigmoid(y) if self.sigmoid else y


class MLP(nn.Module):
    """[Linear, BatchNorm1d, activation, Dropout] per hidden size, optional Linear(.,1) (reference :254-292)."""

    def __init__(self, input_dim, output_layer=True, dims=None, dropout=0, activation="relu"):
        super().__init__()
        layers = []
        for width in (dims or []):
            layers += [nn.Linear(input_dim, width), nn.BatchNorm1d(width), activation_layer(activation),
                       nn.Dropout(p=dropout)]
            input_dim = width
        if output_layer:
            layers.append(nn.Linear(input_dim, 1))
        self.mlp = nn.Sequential(*layers)

    @staticmethod
    def _bn_ok(bn, x):
        return (isinstance(bn, nn.BatchNorm1d) and bn.affine and bn.track_running_stats and bn.momentum is not None and
                x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and (x.shape[0] > 1 or not bn.training) and
                not (torch.is_grad_enabled() and not bn.training and x.requires_grad))

    @classmethod
    def _fusable(cls, bn, act, drop, x):
        return cls._bn_ok(bn, x) and isinstance(act, nn.ReLU) and isinstance(drop, nn.Dropout)

    @staticmethod
    def _linear(lin, x):
        return ops.linear(x, lin.weight, lin.bias) if type(lin) is nn.Linear and ops.linear_ok(x, lin.weight) else lin(x)

    @staticmethod
    def head_after(mods, j, width, bn, dice_mod):
        """The output Linear(width, 1) when mods[j:] is nothing but inactive Dropouts and that Linear, else None."""
        while j < len(mods) - 1 and isinstance(mods[j], nn.Dropout) and (mods[j].p == 0 or not mods[j].training):
            j += 1
        if j == len(mods) - 1 and ops.bn_dice_head_ok(width, bn, dice_mod, mods[j]):
            return mods[j]
        return None

    @staticmethod
    def _chain_blocks(mods):
        """[(Linear, BatchNorm1d, p_drop)] when ``mods`` is nothing but [Linear, BatchNorm1d, ReLU, Dropout] groups, else None."""
        if not mods or len(mods) % 4:
            return None
        out = []
        for i in range(0, len(mods), 4):
            lin, bn, act, drop = mods[i:i + 4]
            if not (type(lin) is nn.Linear and type(bn) is nn.BatchNorm1d and type(act) is nn.ReLU and type(drop) is nn.Dropout):
                return None
            out.append((lin, bn, float(drop.p) if drop.training else 0.0))
        return out

    def _run(self, mods, x):
        # [Linear, BatchNorm1d, ReLU, Dropout] blocks: library GEMM + ONE fused epilogue (csrc/mlp.hip); any other
        # activation (dice, prelu, sigmoid ...) runs the modules as they are.  Every Linear's weight / bias gradient
        # is the split-batch MFMA kernel of csrc/linear.hip.
        i = 0
        while i < len(mods):
            if i + 3 < len(mods) + 0 and isinstance(mods[i], nn.Linear) and self._fusable(mods[i + 1], mods[i + 2],
                                                                                            mods[i + 3], x):
                lin, bn = mods[i], mods[i + 1]
                if type(lin) is nn.Linear and bn.training and ops.linear_ok(x, lin.weight):
                    h, stats = ops.linear_stats(x, lin.weight, lin.bias, bn)  # GEMM epilogue emits the BN statistics
                else:
                    h, stats = self._linear(lin, x), None
                x = ops.bn_relu_dropout(h, bn, mods[i + 3].p if mods[i + 3].training else 0.0, stats=stats)
                i += 4
            elif (i + 2 < len(mods) and isinstance(mods[i], nn.Linear) and type(mods[i + 1]) is nn.BatchNorm1d and
                  type(mods[i + 2]) is Dice and self._bn_ok(mods[i + 1], x) and mods[i].out_features <= 512 and
                  not (torch.is_grad_enabled() and not mods[i + 1].training)):
                # Linear -> BatchNorm1d -> Dice (DIN's ActivationUnit): the normalisation is folded into the Dice passes
                head = self.head_after(mods, i + 3, mods[i].out_features, mods[i + 1], mods[i + 2])
                lin, bn = mods[i], mods[i + 1]
                if type(lin) is nn.Linear and bn.training and ops.linear_ok(x, lin.weight):
                    # ((B L)-row inputs: the tile GEMM's epilogue emits the BatchNorm's per-slab statistics, round 6)
                    h, cst, crows = ops.linear_chunk_stats(x, lin.weight, lin.bias)
                else:
                    h, cst, crows = self._linear(lin, x), None, 0
                if head is not None:
                    # ... -> Linear(C, 1) at the end of the stack: the Dice output is never written (ops.bn_dice_head)
                    return ops.bn_dice_head(h, bn, mods[i + 2].alpha, mods[i + 2].epsilon, head, chunk_stats=cst, chunk_rows=crows)
                x = ops.bn_dice(h, bn, mods[i + 2].alpha, mods[i + 2].epsilon, chunk_stats=cst, chunk_rows=crows)
                i += 3
            elif (i + 3 < len(mods) and isinstance(mods[i], nn.Linear) and type(mods[i + 1]) is nn.BatchNorm1d and
                  self._bn_ok(mods[i + 1], x) and isinstance(mods[i + 3], nn.Dropout) and
                  ops.bn_prelu_dropout_ok(x, mods[i + 1], mods[i + 2])):
                # Linear -> BatchNorm1d -> nn.PReLU() -> Dropout (the two-tower MLPs): ONE epilogue each way as for ReLU
                drop = mods[i + 3]
                x = ops.bn_prelu_dropout(self._linear(mods[i], x), mods[i + 1], mods[i + 2], drop.p if drop.training else 0.0)
                i += 4
            elif (i + 1 < len(mods) and isinstance(mods[i], nn.Linear) and type(mods[i + 1]) is nn.BatchNorm1d and
                  self._bn_ok(mods[i + 1], x)):
                # Linear -> BatchNorm1d in front of Dice / PReLU / ...: the normalisation alone through the same kernels
                x = ops.bn_relu_dropout(self._linear(mods[i], x), mods[i + 1], 0.0, relu=False)
                i += 2
            elif isinstance(mods[i], nn.Linear):
                x = self._linear(mods[i], x)
                i += 1
            elif ops.prelu_ok(mods[i], x):
                x = ops.prelu(x, mods[i].weight)  # nn.PReLU() of the two-tower MLPs: one pass each way
                i += 1
            else:
                x = mods[i](x)
                i += 1
        return x

    def forward(self, x):
        return self._run(list(self.mlp), x)

    def sigmoid_head(self, x, *extras):
        """sigmoid((sum(extras) + mlp(x)).squeeze(1)): the output Linear(., 1), the wide / FM terms of DeepFM / WideDeep
        (deepfm.py:39-43, widedeep.py:35-39) and the sigmoid as ONE kernel (ops.head_sigmoid) when the shapes allow."""
        mods = list(self.mlp)
        if mods and type(mods[-1]) is nn.Linear and mods[-1].out_features == 1:
            blocks = self._chain_blocks(mods[:-1])
            if blocks and ops.mlp_chain_ok(x, blocks, mods[-1], extras):
                # every hidden layer is Linear -> BatchNorm1d -> ReLU -> Dropout (training): ONE autograd node, the
                # BatchNorm / ReLU / Dropout of a layer applied where the next GEMM / the head reads it (ops._MlpChainFn)
                return ops.mlp_chain_sigmoid(x, blocks, mods[-1], *extras)
            h = self._run(mods[:-1], x)
            if ops.head_ok(h, mods[-1], extras):
                return ops.head_sigmoid(h, mods[-1].weight, mods[-1].bias, *extras)
            y = self._linear(mods[-1], h)
        else:
            y = self._run(mods, x)
        for e in reversed(extras):
            y = e + y
        return torch.sigmoid(y.squeeze(1))


class FM(nn.Module):
    """0.5 * sum_d[(sum_f x)^2 - sum_f x^2] (reference layers.py:295-319)."""

    def __init__(self, reduce_sum=True):
        super().__init__()
        self.reduce_sum = reduce_sum

    def forward(self, x):
        return ops.fm(x, self.reduce_sum)


class CrossNetwork(nn.Module):
    """DCN cross layers x <- x0 * (w_l . x) + b_l + x (reference layers.py:390-420)."""

    def __init__(self, input_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        self.w = nn.ModuleList([nn.Linear(input_dim, 1, bias=False) for _ in range(num_layers)])
        self.b = nn.ParameterList([nn.Parameter(torch.zeros((input_dim,))) for _ in range(num_layers)])

    def forward(self, x):
        W = torch.cat([lin.weight for lin in self.w], dim=0)
        Bv = torch.stack(list(self.b), dim=0)
        return ops.cross_network(x, W, Bv)


class CrossNetV2(nn.Module):
    """DCN-v2 full-rank cross layers x <- x0 * (W_l x) + b_l + x (reference layers.py:423-444)."""

    def __init__(self, input_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        self.w = nn.ModuleList([nn.Linear(input_dim, input_dim, bias=False) for _ in range(num_layers)])
        self.b = nn.ParameterList([nn.Parameter(torch.zeros((input_dim,))) for _ in range(num_layers)])

    def forward(self, x):
        ops.require_hip(x)
        x0 = x
        for i in range(self.num_layers):
            if ops.cross_v2_layer_ok(x, self.w[i]):
                # ONE launch: the (d, d) product on the f32-MFMA tile GEMM with Hadamard + bias + residual as its epilogue
                x = ops.cross_v2_layer(x0, x, self.w[i].weight, self.b[i])
            else:  # (d, d) GEMM on hipBLASLt, then the fused Hadamard + bias + residual pass
                x = ops.cross_v2_epilogue(x0, self.w[i](x), self.b[i], x)
        return x


class CrossNetMix(nn.Module):
    """DCN-v2 mixture of low-rank experts (reference layers.py:447-506), experts batched.

    Per layer l and expert e:  g_e = gating_e(x_l);  v = tanh(V_e^T x_l);  v = tanh(C_e v);
    o_e = x0 * (U_e v + bias_l);  x_{l+1} = sum_e softmax(g)_e o_e + x_l.
    The reference loops layers x experts in Python on (B, d, 1) column vectors (150 mm per train
    step); here every layer is three batched GEMMs.  ``gating`` is shared across layers (Q11).
    """

    def __init__(self, input_dim, num_layers=2, low_rank=32, num_experts=4):
        super().__init__()
        self.num_layers = num_layers
        self.num_experts = num_experts
        mk = lambda *shape: nn.Parameter(nn.init.xavier_normal_(torch.empty(*shape)))
        self.u_list = nn.ParameterList([mk(num_experts, input_dim, low_rank) for _ in range(num_layers)])
        self.v_list = nn.ParameterList([mk(num_experts, input_dim, low_rank) for _ in range(num_layers)])
        self.c_list = nn.ParameterList([mk(num_experts, low_rank, low_rank) for _ in range(num_layers)])
        self.gating = nn.ModuleList([nn.Linear(input_dim, 1, bias=False) for _ in range(num_experts)])
        self.bias = nn.ParameterList([nn.Parameter(torch.zeros(input_dim, 1)) for _ in range(num_layers)])

    def forward(self, x):
        ops.require_hip(x)
        B, d = x.shape
        E = self.num_experts
        if (all(type(g) is nn.Linear and g.bias is None for g in self.gating) and
                ops.cross_moe_ok(x, self.num_layers, E, d, self.u_list[0].shape[2])):
            # two dense products + three fused passes per layer (csrc/moe.hip): the experts are the K dimension
            return ops.cross_moe(x, list(self.u_list), list(self.v_list), list(self.c_list), list(self.bias),
                                 [g.weight for g in self.gating])
        x0 = x
        xl = x
        Wg = torch.cat([g.weight for g in self.gating], dim=0)  # (E, d)
        for i in range(self.num_layers):
            U, V, C = self.u_list[i], self.v_list[i], self.c_list[i]
            r = V.shape[2]
            gate = torch.softmax(xl @ Wg.t(), dim=1)  # (B, E)
            v = torch.tanh(xl @ V.permute(1, 0, 2).reshape(d, E * r))  # (B, E*r)
            v = v.view(B, E, r).transpose(0, 1)  # (E, B, r)
            v = torch.tanh(torch.bmm(v, C.transpose(1, 2)))  # (E, B, r): C_e v
            uv = torch.bmm(v, U.transpose(1, 2))  # (E, B, d): U_e v
            if d <= 2048 and E <= 16:
                # bias + Hadamard with x0 + gate-weighted expert mix + residual: one pass (csrc/crossmix.hip)
                xl = ops.cross_mix_epilogue(x0, xl, uv, gate, self.bias[i])
            else:
                expert = x0.unsqueeze(0) * (uv + self.bias[i].view(1, 1, d))  # (E, B, d)
                xl = (expert * gate.t().unsqueeze(2)).sum(dim=0) + xl
        return xl


class SENETLayer(nn.Module):
    """SENET field gating: a = relu(W2 relu(W1 mean_d(x))), out = x * a[..., None] (reference layers.py:509-529)."""

    def __init__(self, num_fields, reduction_ratio=3):
        super().__init__()
        hidden = max(1, int(num_fields / reduction_ratio))
        self.mlp = nn.Sequential(nn.Linear(num_fields, hidden, bias=False), nn.ReLU(), nn.Linear(hidden, num_fields, bias=False),
                                 nn.ReLU())

    def forward(self, x):
        return x * self.mlp(x.mean(dim=-1)).unsqueeze(-1)


class BiLinearInteractionLayer(nn.Module):
    """(W v_i) * v_j for every field pair i < j (reference layers.py:532-565); W shared ("field_all"), one per left
    field ("field_each") or one per pair ("field_interaction").  Same parameters as the reference (``bilinear_layer`` is a
    Linear or a ModuleList of Linear(D, D, bias=False)); evaluated as one batched product over the pairs."""

    def __init__(self, input_dim, num_fields, bilinear_type="field_interaction"):
        super().__init__()
        self.bilinear_type = bilinear_type
        pairs = [(i, j) for i in range(num_fields) for j in range(i + 1, num_fields)]
        self.register_buffer("_left", torch.tensor([i for i, _ in pairs], dtype=torch.long), persistent=False)
        self.register_buffer("_right", torch.tensor([j for _, j in pairs], dtype=torch.long), persistent=False)
        if bilinear_type == "field_all":
            self.bilinear_layer = nn.Linear(input_dim, input_dim, bias=False)
        elif bilinear_type == "field_each":
            self.bilinear_layer = nn.ModuleList([nn.Linear(input_dim, input_dim, bias=False) for _ in range(num_fields)])
        elif bilinear_type == "field_interaction":
            self.bilinear_layer = nn.ModuleList([nn.Linear(input_dim, input_dim, bias=False) for _ in pairs])
        else:
            raise NotImplementedError()

    def forward(self, x):
        left, right = x[:, self._left], x[:, self._right]  # (B, P, D)
        if self.bilinear_type == "field_all":
            return self.bilinear_layer(left) * right
        w = torch.stack([lin.weight for lin in self.bilinear_layer])  # (F or P, D_out, D_in)
        if self.bilinear_type == "field_each":
            w = w[self._left]
        return torch.einsum("bpi,poi->bpo", left, w) * right


class InteractingLayer(nn.Module):
    """AutoInt's multi-head self-attention over the fields with a projected residual and a ReLU
    (reference layers.py:973-1044); W_Q / W_K / W_V / W_Res are applied as one product over their concatenation."""

    def __init__(self, embed_dim, num_heads=2, dropout=0.0, residual=True):
        super().__init__()
        if embed_dim % num_heads != 0:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.head_dim = embed_dim // num_heads
        self.scale = self.head_dim**-0.5
        self.residual = residual
        self.W_Q = nn.Linear(embed_dim, embed_dim, bias=False)
        self.W_K = nn.Linear(embed_dim, embed_dim, bias=False)
        self.W_V = nn.Linear(embed_dim, embed_dim, bias=False)
        self.W_Res = nn.Linear(embed_dim, embed_dim, bias=False) if residual else None
        self.dropout = nn.Dropout(dropout) if dropout > 0 else None

    def forward(self, x):
        B, Fn, D = x.shape
        H, Dh = self.num_heads, self.head_dim
        ws = [self.W_Q.weight, self.W_K.weight, self.W_V.weight] + ([self.W_Res.weight] if self.W_Res is not None else [])
        proj = (x.reshape(B * Fn, D) @ torch.cat(ws, dim=0).t()).view(B, Fn, len(ws), H, Dh)
        q, k, v = (proj[:, :, i].transpose(1, 2) for i in range(3))  # (B, H, F, Dh)
        w = torch.softmax((q @ k.transpose(-2, -1)) * self.scale, dim=-1)
        if self.dropout is not None:
            w = self.dropout(w)
        out = (w @ v).transpose(1, 2).reshape(B, Fn, D)
        if self.W_Res is not None:
            out = out + proj[:, :, 3].reshape(B, Fn, D)
        return torch.relu(out)


class CrossLayer(nn.Module):
    """One cross step without the residual: (w . x_i) * x_0 + b (reference layers.py:371-387; EDCN adds the residual)."""

    def __init__(self, input_dim):
        super().__init__()
        self.w = nn.Linear(input_dim, 1, bias=False)
        self.b = nn.Parameter(torch.zeros(input_dim))

    def forward(self, x_0, x_i):
        return self.w(x_i) * x_0 + self.b

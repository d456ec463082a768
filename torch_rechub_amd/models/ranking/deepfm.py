"""DeepFM (API mirror of torch_rechub/models/ranking/deepfm.py:14-43).

Reference forward: two full gathers of the same tables (deep + fm feature lists), LR on the flattened
fm embeddings, FM, MLP, sigmoid.  Here, when the fm features are plain sparse features of one width,
ONE kernel launch gathers every table once and emits the MLP input (sparse block + dense values),
the FM scalar and the LR scalar (ops.fused_embedding); its backward is one launch too.
Attribute / parameter names are the reference's (``linear.fc``, ``embedding.embed_dict``, ``mlp.mlp``).
"""
import torch

from ... import ops
from ...basic.features import SparseFeature
from ...basic.layers import FM, LR, MLP, EmbeddingLayer


class DeepFM(torch.nn.Module):

    def __init__(self, deep_features, fm_features, mlp_params):
        super().__init__()
        self.deep_features = deep_features
        self.fm_features = fm_features
        self.deep_dims = sum(fea.embed_dim for fea in deep_features)
        self.fm_dims = sum(fea.embed_dim for fea in fm_features)
        self.linear = LR(self.fm_dims)
        self.fm = FM(reduce_sum=True)
        self.embedding = EmbeddingLayer(deep_features + fm_features)
        self.mlp = MLP(self.deep_dims, **mlp_params)

    def _same_sparse_lists(self):
        deep_sparse = [f for f in self.deep_features if isinstance(f, SparseFeature)]
        return len(deep_sparse) == len(self.fm_features) and all(a is b for a, b in zip(deep_sparse, self.fm_features))

    def forward(self, x):
        emb = self.embedding
        dense = [f for f in self.deep_features if not isinstance(f, SparseFeature)]
        w, b = self.linear.fc.weight, self.linear.fc.bias
        if emb.can_fuse(x, self.fm_features):
            # replicated tables: the gather kernel emits the MLP input, the FM scalar and the LR scalar
            shared = self._same_sparse_lists() and emb.can_fuse(x, self.deep_features)
            input_deep, y_fm, y_linear = emb.fused(x, self.fm_features, dense if shared else (), w, b, want_fm=True)
        elif emb.can_fuse_sharded(x, self.fm_features):
            # row-sharded tables: ONE exchange for both feature lists, then the same fused stage over the received rows
            shared = self._same_sparse_lists() and emb.can_fuse_sharded(x, self.deep_features)
            rows = emb.sharded_rows(x, self.fm_features)
            input_deep, y_fm, y_linear = ops.fused_rows(rows, len(self.fm_features),
                                                        [x[f.name].float() for f in dense] if shared else (), w, b,
                                                        want_fm=True)
        else:
            shared = False
            input_fm = emb(x, self.fm_features, squeeze_dim=False)
            y_linear = self.linear(input_fm.flatten(start_dim=1))
            y_fm = self.fm(input_fm)
        if not shared:
            input_deep = emb(x, self.deep_features, squeeze_dim=True)
        return self.mlp.sigmoid_head(input_deep, y_linear, y_fm)

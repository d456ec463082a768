"""DSSM two-tower model (API mirror of torch_rechub/models/matching/dssm.py:16-72).

Both towers are EmbeddingLayer(squeeze_dim=True) -> MLP(no output layer) -> L2 normalise; ``forward`` returns
sigmoid(sum(user * item)) (``temperature`` is stored but, as in the reference :51, never applied), or one tower's
embedding when ``mode`` is "user" / "item".  The gathers run on the fused HIP kernels (sequence history features on the
gather+pool kernel); the towers' GEMMs are library calls.
"""
import torch
import torch.nn.functional as F

from ...basic.layers import MLP, EmbeddingLayer


class DSSM(torch.nn.Module):

    def __init__(self, user_features, item_features, user_params, item_params, temperature=1.0):
        super().__init__()
        self.user_features = user_features
        self.item_features = item_features
        self.temperature = temperature
        self.user_dims = sum(fea.embed_dim for fea in user_features)
        self.item_dims = sum(fea.embed_dim for fea in item_features)
        self.embedding = EmbeddingLayer(user_features + item_features)
        self.user_mlp = MLP(self.user_dims, output_layer=False, **user_params)
        self.item_mlp = MLP(self.item_dims, output_layer=False, **item_params)
        self.mode = None

    def forward(self, x):
        user_embedding = self.user_tower(x)
        item_embedding = self.item_tower(x)
        if self.mode == "user":
            return user_embedding
        if self.mode == "item":
            return item_embedding
        return torch.sigmoid(torch.mul(user_embedding, item_embedding).sum(dim=1))

    def user_tower(self, x):
        if self.mode == "item":
            return None
        input_user = self.embedding(x, self.user_features, squeeze_dim=True)
        return F.normalize(self.user_mlp(input_user), p=2, dim=1)

    def item_tower(self, x):
        if self.mode == "user":
            return None
        input_item = self.embedding(x, self.item_features, squeeze_dim=True)
        return F.normalize(self.item_mlp(input_item), p=2, dim=1)

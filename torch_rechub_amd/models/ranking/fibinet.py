"""FiBiNET (API mirror of torch_rechub/models/ranking/fibinet.py:15-42): SENET field gating + bilinear interactions.

Reference forward: gather (B, F, D) -> SENETLayer -> BiLinearInteractionLayer on the raw and on the gated embeddings
-> concat -> MLP -> sigmoid.  Same constructor, attribute names and state_dict keys (``senet_layer.mlp.{0,2}.weight``,
``bilinear_interaction.bilinear_layer[.k].weight``, ``mlp.mlp.*``); the gather is the fused HIP launch, the MLP the
fused BatchNorm / head path, and the F (F - 1) / 2 per-pair Linear calls of the reference run as ONE batched product.
"""
import torch

from ...basic.features import SparseFeature
from ...basic.layers import MLP, BiLinearInteractionLayer, EmbeddingLayer, SENETLayer


class FiBiNet(torch.nn.Module):

    def __init__(self, features, mlp_params, reduction_ratio=3, bilinear_type="field_interaction", **kwargs):
        super().__init__()
        self.features = features
        self.embedding = EmbeddingLayer(features)
        embedding_dim = max(fea.embed_dim for fea in features)
        num_fields = len([fea for fea in features if isinstance(fea, SparseFeature) and fea.shared_with is None])
        self.senet_layer = SENETLayer(num_fields, reduction_ratio)
        self.bilinear_interaction = BiLinearInteractionLayer(embedding_dim, num_fields, bilinear_type)
        self.dims = num_fields * (num_fields - 1) * embedding_dim
        self.mlp = MLP(self.dims, **mlp_params)

    def forward(self, x):
        embed_x = self.embedding(x, self.features)
        embed_senet = self.senet_layer(embed_x)
        both = torch.cat([self.bilinear_interaction(embed_x), self.bilinear_interaction(embed_senet)], dim=1)
        return self.mlp.sigmoid_head(both.flatten(start_dim=1))

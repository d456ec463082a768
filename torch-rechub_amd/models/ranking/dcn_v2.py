"""DCN-v2 (API mirror of torch_rechub/models/ranking/dcn_v2.py:13-59)."""
import torch

from ...basic.layers import LR, MLP, CrossNetMix, CrossNetV2, EmbeddingLayer


class DCNv2(torch.nn.Module):

    def __init__(self, features, n_cross_layers, mlp_params, model_structure="parallel", use_low_rank_mixture=True,
                 low_rank=32, num_experts=4, **kwargs):
        super().__init__()
        self.features = features
        self.dims = sum(fea.embed_dim for fea in features)
        self.embedding = EmbeddingLayer(features)
        if use_low_rank_mixture:
            self.crossnet = CrossNetMix(self.dims, n_cross_layers, low_rank=low_rank, num_experts=num_experts)
        else:
            self.crossnet = CrossNetV2(self.dims, n_cross_layers)
        self.model_structure = model_structure
        assert self.model_structure in ["crossnet_only", "stacked", "parallel"], \
            "model_structure={} not supported!".format(self.model_structure)
        if self.model_structure == "stacked":
            self.stacked_dnn = MLP(self.dims, output_layer=False, **mlp_params)
            final_dim = mlp_params["dims"][-1]
        if self.model_structure == "parallel":
            self.parallel_dnn = MLP(self.dims, output_layer=False, **mlp_params)
            final_dim = mlp_params["dims"][-1] + self.dims
        if self.model_structure == "crossnet_only":
            final_dim = self.dims
        self.linear = LR(final_dim)

    def forward(self, x):
        embed_x = self.embedding(x, self.features, squeeze_dim=True)
        cross_out = self.crossnet(embed_x)
        if self.model_structure == "crossnet_only":
            final_out = cross_out
        elif self.model_structure == "stacked":
            final_out = self.stacked_dnn(cross_out)
        else:
            final_out = torch.cat([cross_out, self.parallel_dnn(embed_x)], dim=1)
        return torch.sigmoid(self.linear(final_out).squeeze(1))

"""DCN-v2 (API mirror of torch_rechub/models/ranking/dcn_v2.py:13-59).

Three wirings of one cross stack (``CrossNetMix`` with low-rank experts, or full-rank ``CrossNetV2``) and one MLP:
"crossnet_only", "stacked" (MLP after the cross stack) and "parallel" (both on the embeddings, concatenated), followed
by an LR and a sigmoid.  Module names follow the reference (``crossnet``, ``stacked_dnn`` / ``parallel_dnn``,
``linear``) so its checkpoints load.  The dense contractions stay library GEMMs; everything around them is
``rh_cross_v2_epilogue_*`` / ``rh_cross_mix_epilogue_*`` (basic/layers.py).
"""
import torch
from torch import nn

from ... import ops
from ...basic.layers import LR, MLP, CrossNetMix, CrossNetV2, EmbeddingLayer

_STRUCTURES = ("crossnet_only", "stacked", "parallel")


class DCNv2(nn.Module):

    def __init__(self, features, n_cross_layers, mlp_params, model_structure="parallel", use_low_rank_mixture=True,
                 low_rank=32, num_experts=4, **kwargs):
        super().__init__()
        assert model_structure in _STRUCTURES, "model_structure={} not supported!".format(model_structure)
        width = sum(f.embed_dim for f in features)
        self.features, self.dims, self.model_structure = features, width, model_structure
        self.embedding = EmbeddingLayer(features)
        self.crossnet = (CrossNetMix(width, n_cross_layers, low_rank=low_rank, num_experts=num_experts)
                         if use_low_rank_mixture else CrossNetV2(width, n_cross_layers))
        out_width = width
        if model_structure == "stacked":
            self.stacked_dnn = MLP(width, output_layer=False, **mlp_params)
            out_width = mlp_params["dims"][-1]
        elif model_structure == "parallel":
            self.parallel_dnn = MLP(width, output_layer=False, **mlp_params)
            out_width = width + mlp_params["dims"][-1]
        self.linear = LR(out_width)

    def forward(self, x):
        h = self.embedding(x, self.features, squeeze_dim=True)
        if self.model_structure == "parallel" and getattr(self, "parallel_branches", True) and h.is_cuda:
            # (round 6: the cross stack and the MLP read the same embeddings and meet at the concatenation -- two independent
            # chains of small launches (B x 400-wide products fill a fraction of the 256 CUs): side by side on two streams,
            # forward and backward (ops.run_beside); the reference runs one after the other, dcn_v2.py:52-56.  Same kernels,
            # same arithmetic.  ``model.parallel_branches = False`` restores the sequence)
            z, deep = ops.run_beside(lambda: self.crossnet(h), lambda: self.parallel_dnn(h), side_inputs=(h,))
            z = torch.cat((z, deep), dim=1)
        else:
            z = self.crossnet(h)
            if self.model_structure == "stacked":
                z = self.stacked_dnn(z)
            elif self.model_structure == "parallel":
                z = torch.cat((z, self.parallel_dnn(h)), dim=1)
        fc = self.linear.fc
        if ops.head_ok(z, fc, ()):
            return ops.head_sigmoid(z, fc.weight, fc.bias)
        return torch.sigmoid(fc(z).squeeze(1))

"""Deep Interest Network (API mirror of torch_rechub/models/ranking/din.py:16-93).

Quirks kept on purpose: the attention applies NO padding mask, padded positions go through the
BatchNorm statistics of the attention MLP (B*L rows) and receive a weight; history feature i pairs with
target feature i by position.
"""
import torch
from torch import nn

from ... import ops
from ...basic.activation import Dice
from ...basic.layers import MLP, EmbeddingLayer


class DIN(nn.Module):

    def __init__(self, features, history_features, target_features, mlp_params, attention_mlp_params):
        super().__init__()
        self.features = features
        self.history_features = history_features
        self.target_features = target_features
        self.num_history_features = len(history_features)
        self.all_dims = sum(fea.embed_dim for fea in features + history_features + target_features)
        self.embedding = EmbeddingLayer(features + history_features + target_features)
        self.attention_layers = nn.ModuleList(
            [ActivationUnit(fea.embed_dim, **attention_mlp_params) for fea in self.history_features])
        self.mlp = MLP(self.all_dims, activation="dice", **mlp_params)

    def forward(self, x):
        embed_x_features = self.embedding(x, self.features)  # (B, n_feat, D)
        # per-feature tensors instead of the reference's (B, n_hist, L, D) / (B, n_tgt, D) stacks that din.py:40-47 takes
        # apart again: same values, no concatenation here and no zero-fill + copy + add per slice in the backward
        hist = self.embedding.pieces(x, self.history_features)  # n_hist x (B, L, D)
        tgt = self.embedding.pieces(x, self.target_features)  # n_tgt x (B, D)
        if self.num_history_features == 2 and getattr(self, "attention_branches", True) and hist[0].is_cuda:
            # (round 6: the two activation units side by side on two streams -- the bandwidth-bound Dice passes of one under
            # the matrix products of the other; the reference runs them one after the other, din.py:48-52.  Same kernels, same
            # arithmetic; configs[3] 5.30 -> 5.09 ms per step.  ``model.attention_branches = False`` restores the sequence)
            pooled = list(ops.run_beside(lambda: self.attention_layers[0](hist[0], tgt[0]),
                                         lambda: self.attention_layers[1](hist[1], tgt[1]), side_inputs=(hist[1], tgt[1])))
        else:
            pooled = [self.attention_layers[i](hist[i], tgt[i]) for i in range(self.num_history_features)]
        mlp_in = torch.cat(pooled + tgt + [embed_x_features.flatten(start_dim=1)], dim=1)
        return torch.sigmoid(self.mlp(mlp_in).squeeze(1))


class ActivationUnit(nn.Module):
    """DIN local activation unit: weight_l = MLP([t, h_l, t-h_l, t*h_l]); out = sum_l weight_l * h_l."""

    def __init__(self, emb_dim, dims=None, activation="dice", use_softmax=False):
        super().__init__()
        if dims is None:
            dims = [36]
        self.emb_dim = emb_dim
        self.use_softmax = use_softmax
        self.attention = MLP(4 * self.emb_dim, dims=dims, activation=activation)

    def forward(self, history, target):
        B, L, D = history.shape
        if not ops.din_dim_ok(D):
            # widths outside 4, 8, 16, 32, 64, 128 (PaddedEmbedding tables): the reference's op chain (din.py:77-92) on
            # the device -- gathers, the attention MLP's fused layers and the optimizer are still the HIP path
            ops.require_hip(history, target)
            t = target.unsqueeze(1).expand(-1, L, -1)
            att_input = torch.cat([t, history, t - history, t * history], dim=-1).view(-1, 4 * D)
            att_weight = self.attention(att_input).view(-1, L)
            if self.use_softmax:
                att_weight = att_weight.softmax(dim=-1)
            return (att_weight.unsqueeze(-1) * history).sum(dim=1)
        mods = list(self.attention.mlp)
        fused = (len(mods) >= 3 and ops.din_att_l1_ok(history, target, mods[0]) and type(mods[1]) is nn.BatchNorm1d and
                 type(mods[2]) is Dice and ops.bn_dice_ok(history.new_empty((2, mods[0].out_features)), mods[1], mods[2]) and
                 not (torch.is_grad_enabled() and not mods[1].training))
        if fused:
            # first layer on the operand built in registers, BatchNorm statistics as its epilogue (csrc/dinmlp.hip):
            # the (B*L, 4D) tensor of din.py:81-85 never exists in the forward
            z, stats = ops.din_att_l1(history, target, mods[0].weight, mods[0].bias, mods[1].training)
            rows = ops._lib.call("rh_din_att_l1_chunk_rows", B * L) if stats is not None else 0
            head = self.attention.head_after(mods, 3, mods[0].out_features, mods[1], mods[2])
            if head is not None:  # one hidden layer: its Dice output only feeds the Linear(., 1)
                att_weight = ops.bn_dice_head(z, mods[1], mods[2].alpha, mods[2].epsilon, head, chunk_stats=stats,
                                              chunk_rows=rows).view(-1, L)
            else:
                x = ops.bn_dice(z, mods[1], mods[2].alpha, mods[2].epsilon, chunk_stats=stats, chunk_rows=rows)
                att_weight = self.attention._run(mods[3:], x).view(-1, L)
        else:
            att_input = ops.din_att_input(history, target)  # (B*L, 4D) = [t, h, t-h, t*h], one kernel
            att_weight = self.attention(att_input).view(-1, L)
        if self.use_softmax:
            att_weight = att_weight.softmax(dim=-1)
        return ops.din_att_pool(att_weight, history)  # sum_l w_l * h_l, one kernel

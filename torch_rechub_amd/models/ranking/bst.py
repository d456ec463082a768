"""Behavior Sequence Transformer (API mirror of torch_rechub/models/ranking/bst.py:16-90).

Reference forward: history features (concat pooling) fused per time step, the target appended as the last position,
absolute positional table added, ``nn.TransformerEncoder`` under a key-padding mask (a step is padding only when ALL
history features are padding there; the target never is), the target position's output joined with the target and
profile embeddings for the MLP.  Attribute / checkpoint names are the reference's (``embedding``, ``pos_embedding``,
``transformer_layers``, ``mlp``).  Here every lookup is a HIP gather (plain features: one fused launch; histories: the
gather + concat kernel), the encoder is the library's, the MLP the fused BatchNorm path.
"""
import torch
from torch import nn

from ...basic.layers import MLP, EmbeddingLayer


class BST(nn.Module):

    def __init__(self, features, history_features, target_features, mlp_params, nhead=8, dropout=0.2, num_layers=1,
                 max_seq_len=51):
        super().__init__()
        self.features, self.history_features, self.target_features = features, history_features, target_features
        self.max_seq_len = max_seq_len
        self.item_dim = sum(f.embed_dim for f in history_features)
        target_dim = sum(f.embed_dim for f in target_features)
        if self.item_dim != target_dim:
            raise ValueError(f"sum of history_features embed_dim ({self.item_dim}) must equal sum of target_features "
                             f"embed_dim ({target_dim})")
        if self.item_dim % nhead != 0:
            raise ValueError(f"item_dim ({self.item_dim}) must be divisible by nhead ({nhead})")
        self.all_dims = sum(f.embed_dim for f in features + target_features) + self.item_dim
        self.embedding = EmbeddingLayer(features + history_features + target_features)
        self.pos_embedding = nn.Embedding(max_seq_len, self.item_dim)
        self.pos_embedding._rh_dense = True  # max_seq_len rows read as one slice: a dense parameter, not a lookup table
        layer = nn.TransformerEncoderLayer(d_model=self.item_dim, nhead=nhead, dropout=dropout, activation=nn.LeakyReLU(),
                                           batch_first=True)
        # (a LeakyReLU feed-forward never takes the nested-tensor fast path; saying so avoids the constructor's warning)
        self.transformer_layers = nn.TransformerEncoder(layer, num_layers=num_layers, enable_nested_tensor=False)
        self.mlp = MLP(self.all_dims, **mlp_params)

    def _padding_steps(self, x):
        """(B, T) True where every history feature holds its padding id (0 when the feature names none)."""
        pad = None
        for fea in self.history_features:
            here = x[fea.name].long() == (0 if fea.padding_idx is None else fea.padding_idx)
            pad = here if pad is None else pad & here
        return pad

    def forward(self, x):
        profile = self.embedding(x, self.features, squeeze_dim=True)
        history = self.embedding(x, self.history_features)  # (B, H, T, D)
        target = self.embedding(x, self.target_features)  # (B, K, D)
        B, H, T, _ = history.shape
        steps = history.permute(0, 2, 1, 3).reshape(B, T, self.item_dim)  # per step: the H history vectors side by side
        seq = torch.cat([steps, target.reshape(B, 1, self.item_dim)], dim=1)
        if T + 1 > self.max_seq_len:
            raise ValueError(f"sequence length {T + 1} exceeds max_seq_len {self.max_seq_len}")
        seq = seq + self.pos_embedding.weight[:T + 1]
        key_padding = torch.cat([self._padding_steps(x), torch.zeros((B, 1), dtype=torch.bool, device=seq.device)], dim=1)
        encoded = self.transformer_layers(seq, src_key_padding_mask=key_padding)
        mlp_in = torch.cat([encoded[:, -1, :], target.flatten(start_dim=1), profile], dim=1)
        return torch.sigmoid(self.mlp(mlp_in).squeeze(1))

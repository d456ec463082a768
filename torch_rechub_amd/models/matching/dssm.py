"""DSSM two-tower model (API mirror of torch_rechub/models/matching/dssm.py:16-72).

Both towers are EmbeddingLayer(squeeze_dim=True) -> MLP(no output layer) -> L2 normalise; ``forward`` returns
sigmoid(sum(user * item)) (``temperature`` is stored but, as in the reference :51, never applied), or one tower's
embedding when ``mode`` is "user" / "item" (the other tower then yields None).  The gathers run on the fused HIP
kernels (sequence history features on the gather+pool kernel), the towers' BatchNorm through csrc/mlp.hip, their GEMMs
are library calls.
"""

import torch
import torch.nn.functional as F
from torch import nn

from ... import ops
from ...basic.layers import MLP, EmbeddingLayer


class DSSM(nn.Module):

    def __init__(self, user_features, item_features, user_params, item_params, temperature=1.0):
        super().__init__()
        self.user_features, self.item_features = user_features, item_features
        self.temperature = temperature
        self.user_dims, self.item_dims = (sum(f.embed_dim for f in feas) for feas in (user_features, item_features))
        self.embedding = EmbeddingLayer(user_features + item_features)
        self.user_mlp = MLP(self.user_dims, output_layer=False, **user_params)
        self.item_mlp = MLP(self.item_dims, output_layer=False, **item_params)
        self.mode = None

    def _tower(self, x, features, mlp):
        h = mlp(self.embedding(x, features, squeeze_dim=True))
        # F.normalize(h, p=2, dim=1) (dssm.py:56,66) as one HIP launch each way (csrc/match.hip)
        return ops.l2_normalize(h) if ops.l2_normalize_ok(h) else F.normalize(h, p=2, dim=1)

    def _mlp_norm(self, h, mlp):
        h = mlp(h)
        return ops.l2_normalize(h) if ops.l2_normalize_ok(h) else F.normalize(h, p=2, dim=1)

    def towers(self, x):
        """(user_tower(x), item_tower(x)) with the two MLPs side by side (extension; the reference runs them one after the
        other, trainers/match_trainer.py:112-113).  Both gathers stay on the calling stream -- they refresh optimizer state
        in order -- then the item tower's MLP + normalisation run on a second stream beside the user tower's: each of their
        ~12 launches per direction fills a fraction of the chip.  On by default since round 6 (``model.tower_branches = False``
        restores the sequence): in round 5 the configs[4] step was bounded by the deferred window sweep (0.823 ms either way);
        with the round-6 sweep the chain is the longer path and the branches take it 0.776 -> 0.746 ms (same box)."""
        if self.mode is not None or not getattr(self, "tower_branches", True):
            return self.user_tower(x), self.item_tower(x)
        hu = self.embedding(x, self.user_features, squeeze_dim=True)
        hi = self.embedding(x, self.item_features, squeeze_dim=True)
        if not hu.is_cuda:
            return self._mlp_norm(hu, self.user_mlp), self._mlp_norm(hi, self.item_mlp)
        return ops.run_beside(lambda: self._mlp_norm(hu, self.user_mlp), lambda: self._mlp_norm(hi, self.item_mlp),
                              side_inputs=(hi,))

    def user_tower(self, x):
        return None if self.mode == "item" else self._tower(x, self.user_features, self.user_mlp)

    def item_tower(self, x):
        return None if self.mode == "user" else self._tower(x, self.item_features, self.item_mlp)

    def forward(self, x):
        u, v = self.user_tower(x), self.item_tower(x)
        if self.mode in ("user", "item"):
            return u if self.mode == "user" else v
        return torch.sigmoid((u * v).sum(dim=1))

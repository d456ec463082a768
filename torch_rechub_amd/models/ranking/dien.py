"""Deep Interest Evolution Network (API mirror of torch_rechub/models/ranking/dien.py:17-176).

``forward`` returns ``(prediction (B,), alpha * auxiliary loss)`` -- train it with ``CTRTrainer(loss_mode=False)``.
Parameter names are the reference's (``interest_extractor_layers.i`` = ``nn.GRU``, ``interest_evolving_layers.i`` =
``AUGRU`` with ``augru_cell.{Wu,Uu,bu,Wr,Ur,br,Wh,Uh,bh}`` and ``Wa``), so checkpoints interchange.

What differs is how the same numbers are produced, so that the step has no host synchronisation and a static shape
(hipGraph-capturable); the reference syncs three times per history feature (``seq_lens.cpu()``, ``has_hist.any()``,
boolean-mask indexing):

* interest extractor: the reference packs the first ``len`` steps of every row (``len`` = number of non-padding ids,
  dien.py:134-143).  A GRU is causal, so running it over the padded block and zeroing the steps >= len gives the same
  outputs; rows without history come out zero either way.  The ``nn.GRU`` module keeps its parameters (checkpoint
  ABI); its cell runs on the recurrence kernel of csrc/augru.hip (update gate 1 - z, weight 1, bias_hh on the state
  product: ``ops.gru``).
* auxiliary loss (dien.py:106-121): mean BCE over the valid (step, step + 1) pairs, written as a weighted sum divided by
  the pair count (0 when there is none) instead of indexing the valid rows out.
* AUGRU (dien.py:38-66): attention = softmax of (x Wa) . target over the valid steps; a padded step has weight 0 and
  leaves the state untouched, rows without history keep the zero state.  The three input projections of all steps are
  ONE product; the recurrence itself and its backward through time are one HIP launch each (``ops.augru``,
  csrc/augru.hip: one lane per sample, state in registers) for embed_dim 4 / 8 / 16 / 32, a per-step product otherwise.
"""
import torch
import torch.nn.functional as F
from torch import nn
from torch.nn import Parameter, init

from ... import ops
from ...basic.layers import MLP, EmbeddingLayer


def _xavier(*shape):
    return init.xavier_uniform_(Parameter(torch.empty(*shape)))


class AUGRU_Cell(nn.Module):
    """GRU cell whose update gate is scaled by the step's attention weight (paper Eq. 16)."""

    def __init__(self, embed_dim):
        super().__init__()
        for gate in ("u", "r", "h"):  # creation order fixes the RNG stream and the state_dict order: W, U, b per gate
            setattr(self, "W" + gate, _xavier(embed_dim, embed_dim))
            setattr(self, "U" + gate, _xavier(embed_dim, embed_dim))
            setattr(self, "b" + gate, _xavier(1, embed_dim))

    def input_weights(self):
        return torch.cat([self.Wu, self.Wr, self.Wh], dim=1), torch.cat([self.bu, self.br, self.bh], dim=1)

    def state_weights(self):
        return torch.cat([self.Uu, self.Ur, self.Uh], dim=1)

    def step(self, xw, h, a, U):
        """xw: the step's input projections + biases (B, 3D); h: (B, D); a: (B, 1)."""
        D = h.shape[1]
        hu = h @ U
        u = torch.sigmoid(xw[:, :D] + hu[:, :D])
        r = torch.sigmoid(xw[:, D:2 * D] + hu[:, D:2 * D])
        cand = torch.tanh(xw[:, 2 * D:] + r * hu[:, 2 * D:])
        gate = a * u
        return (1 - gate) * h + gate * cand

    def forward(self, x, h_1, a):
        W, b = self.input_weights()
        return self.step(x @ W + b, h_1, a, self.state_weights())


class AUGRU(nn.Module):

    def __init__(self, embed_dim):
        super().__init__()
        self.embed_dim = embed_dim
        self.augru_cell = AUGRU_Cell(embed_dim)
        self.Wa = _xavier(embed_dim, embed_dim)

    @staticmethod
    def _project(x2d, W, b=None):
        """x2d W (+ b) over all B*T rows through ops.linear, whose weight / bias gradient is the split-batch MFMA kernel
        (the reduction over B*T rows is what a library GEMM is slow at)."""
        return ops.linear(x2d, W.t(), None if b is None else b.reshape(-1))

    def attention(self, x, item, mask=None):
        B, T, D = x.shape
        scores = (self._project(x.reshape(B * T, D), self.Wa).view(B, T, D) * item.unsqueeze(1)).sum(-1)  # (B, T)
        if mask is None:
            return torch.softmax(scores, dim=1)
        some = mask.any(dim=1, keepdim=True)
        attn = torch.softmax(scores.masked_fill(~mask, float("-inf")), dim=1)
        # rows without a valid step (softmax of all -inf): uniform weights, as dien.py:57-60
        return torch.where(some, attn, torch.full_like(attn, 1.0 / attn.shape[1]))

    def forward(self, x, item, mask=None):
        B, T, D = x.shape
        attn = self.attention(x, item, mask)
        W, b = self.augru_cell.input_weights()
        xw = self._project(x.reshape(B * T, D), W, b).view(B, T, 3 * D)
        U = self.augru_cell.state_weights()
        if ops.augru_ok(xw, D):  # the whole recurrence (and its backward through time) as one HIP launch
            outs = ops.augru(xw, attn, U)
            return outs, outs[:, -1]
        h = x.new_zeros(B, D)
        outs = []
        for t in range(T):
            h = self.augru_cell.step(xw[:, t], h, attn[:, t:t + 1], U)
            outs.append(h)
        return torch.stack(outs, dim=1), h


class DIEN(nn.Module):

    def __init__(self, features, history_features, neg_history_features, target_features, mlp_params, alpha=0.2):
        super().__init__()
        self.alpha = alpha
        self.features, self.history_features = features, history_features
        self.neg_history_features, self.target_features = neg_history_features, target_features
        self.all_dims = sum(f.embed_dim for f in features + history_features + target_features)
        self.embedding = EmbeddingLayer(features + history_features + neg_history_features + target_features)
        self.interest_extractor_layers = nn.ModuleList(
            [nn.GRU(f.embed_dim, f.embed_dim, batch_first=True) for f in history_features])
        self.interest_evolving_layers = nn.ModuleList([AUGRU(f.embed_dim) for f in history_features])
        self.mlp = MLP(self.all_dims, activation="dice", **mlp_params)
        self.BCELoss = nn.BCELoss()

    def auxiliary(self, outs, pos_emb, neg_emb, mask=None):
        """h_t should score e_{t+1} high and the sampled negative low (paper Eq. 7), over pairs of valid steps."""
        h, pos, neg = outs[:, :-1], pos_emb[:, 1:], neg_emb[:, 1:]
        if mask is None:
            valid = torch.ones(h.shape[:2], dtype=torch.bool, device=h.device)
        else:
            valid = mask[:, :-1] & mask[:, 1:]
        w = valid.to(h.dtype)
        pairs = w.sum()
        p_pos = torch.sigmoid((h * pos).sum(-1))
        p_neg = torch.sigmoid((h * neg).sum(-1))
        total = (F.binary_cross_entropy(p_pos, torch.ones_like(p_pos), weight=w, reduction="sum") +
                 F.binary_cross_entropy(p_neg, torch.zeros_like(p_neg), weight=w, reduction="sum"))
        return total / pairs.clamp(min=1.0)

    def forward(self, x):
        profile = self.embedding(x, self.features, squeeze_dim=True)
        history = self.embedding(x, self.history_features)  # (B, H, T, D)
        negatives = self.embedding(x, self.neg_history_features)
        target = self.embedding(x, self.target_features)  # (B, H, D)
        T = history.shape[2]
        steps = torch.arange(T, device=history.device).unsqueeze(0)
        aux = 0
        evolved = []
        for i, fea in enumerate(self.history_features):
            seq = history[:, i]
            mask = self.embedding.input_mask(x, fea).squeeze(1).bool()  # (B, T)
            packed = (steps < mask.sum(dim=1, keepdim=True)).unsqueeze(-1)  # the steps pack_padded_sequence keeps
            extractor = self.interest_extractor_layers[i]
            # the GRU through the same recurrence kernel as the AUGRU when the width allows; the library's otherwise
            states = ops.gru(extractor, seq) if ops.gru_ok(extractor, seq) else extractor(seq)[0]
            interests = states * packed
            aux = aux + self.auxiliary(interests, seq, negatives[:, i], mask)
            _, h = self.interest_evolving_layers[i](interests, target[:, i], mask)
            evolved.append(h * mask.any(dim=1, keepdim=True))
        mlp_in = torch.cat(evolved + [target.flatten(start_dim=1), profile], dim=1)
        return torch.sigmoid(self.mlp(mlp_in).squeeze(1)), self.alpha * aux

"""MatchTrainer (API mirror of torch_rechub/trainers/match_trainer.py:12-263) on the HIP hot path.

Same constructor and training modes {0: point-wise BCE, 1: pair-wise BPR, 2: list-wise softmax}; with
``in_batch_neg=True`` the step is: user/item towers -> (B, B) scores (library GEMM) -> batched in-batch negative
sampling -> gather [positive, negatives] -> cross entropy / BPR (match_trainer.py:118-138).  Everything else — optimizer
(TableAdam, lazy-exact), gradient bucket, RCCL data parallelism, hipGraph step — is inherited from CTRTrainer; in-batch
negatives are taken from the local batch of each rank (the reference's in-batch branch is single-device, :119), or, with
``global_negatives=True``, from the items of all ranks: the step one process would take on the global batch.
"""
import os

import torch
import tqdm

from .. import ops, sharding
from ..basic.loss_func import BPRLoss
from ..utils.match import gather_inbatch_logits, inbatch_negative_sampling, random_inbatch_negatives
from .ctr_trainer import CTRTrainer


class MatchTrainer(CTRTrainer):

    def __init__(self, model, mode=0, in_batch_neg=False, in_batch_neg_ratio=None, hard_negative=False,
                 sampler_seed=None, optimizer_fn=torch.optim.Adam, optimizer_params=None, regularization_params=None,
                 scheduler_fn=None, scheduler_params=None, n_epoch=10, earlystop_patience=10, device="cpu", gpus=None,
                 model_path="./", model_logger=None, global_negatives=False, sampler_stream="fast",
                 deterministic_logits=None, **kw):
        if in_batch_neg and not (hasattr(model, "user_tower") and hasattr(model, "item_tower")):
            raise ValueError(f"Model {type(model).__name__} does not support in-batch negative sampling. "
                             "Only two-tower models with user_tower() and item_tower() methods are supported, "
                             "such as DSSM, YoutubeDNN, MIND, GRU4Rec, SINE, ComiRec, SASRec, NARM, STAMP, etc.")
        if mode not in (0, 1, 2):
            raise ValueError("mode only contain value in %s, but got %s" % ([0, 1, 2], mode))
        super().__init__(model, optimizer_fn=optimizer_fn, optimizer_params=optimizer_params,
                         regularization_params=regularization_params, scheduler_fn=scheduler_fn,
                         scheduler_params=scheduler_params, n_epoch=n_epoch, earlystop_patience=earlystop_patience,
                         device=device, gpus=gpus, model_path=model_path, model_logger=model_logger, **kw)
        self.mode = mode
        self.in_batch_neg = in_batch_neg
        self.in_batch_neg_ratio = in_batch_neg_ratio
        self.hard_negative = hard_negative
        # global_negatives: with more than one rank, score the local users against the items of EVERY rank (one
        # all-gather of the (B, d) item embeddings; its backward is a reduce-scatter) -- the in-batch step of one
        # process on the global batch.  Off: each rank samples inside its own batch.
        self.global_negatives = bool(global_negatives)
        # deterministic_logits: the direct in-batch logits (csrc/match.hip: no (B, C) score matrix) accumulate the item
        # tower's gradient with row-wide float atomics, whose order -- hence the last bit of that gradient and of a whole
        # training run -- is not fixed.  True keeps the reference's matmul + gather form (trainers/match_trainer.py:118-138,
        # fixed summation order, ~9 % slower at configs[4]); default False.
        self.deterministic_logits = bool(deterministic_logits)
        # "fast": one HIP launch, own counter-based stream (distribution-preserving, hipGraph-replayable);
        # "reference": the reference's randperm-per-row draw, bit-identical indices for the same sampler_seed
        if sampler_stream not in ("fast", "reference"):
            raise ValueError("sampler_stream must be 'fast' or 'reference'")
        self.sampler_stream = sampler_stream
        if sampler_stream == "reference" and in_batch_neg and not hard_negative:
            self.use_graph = False  # B randperm launches + host control flow per step
        self._sampler_generator = None
        if sampler_seed is not None:
            self._sampler_generator = torch.Generator(device=self.device)
            self._sampler_generator.manual_seed(sampler_seed)
        if mode == 0:
            self.criterion = torch.nn.CrossEntropyLoss() if in_batch_neg else torch.nn.BCELoss()
        elif mode == 1:
            self.criterion = BPRLoss()
        else:
            self.criterion = torch.nn.CrossEntropyLoss()

    def _prepare_target(self, y):
        return y.float() if self.mode == 0 else y.long()

    def _zero_targets(self, n):
        t = getattr(self, "_targets", None)
        if t is None or t.numel() != n:
            t = torch.zeros(n, dtype=torch.long, device=self.device)
            self._targets = t  # class 0 = the positive logit; constant, so kept out of the per-step work
        return t

    def _compute_loss(self, x_dict, y):
        if self.in_batch_neg:
            if hasattr(self.model, "towers"):  # both towers, their MLPs side by side on two streams (models/matching/dssm.py)
                user_embedding, item_embedding = self.model.towers(x_dict)
            else:
                user_embedding = self.model.user_tower(x_dict)
                item_embedding = self.model.item_tower(x_dict)
            if user_embedding is None or item_embedding is None:
                raise ValueError("Model must return user/item embeddings when in_batch_neg is True.")
            if user_embedding.dim() > 2 and user_embedding.size(1) == 1:
                user_embedding = user_embedding.squeeze(1)
            if item_embedding.dim() > 2 and item_embedding.size(1) == 1:
                item_embedding = item_embedding.squeeze(1)
            if user_embedding.dim() != 2 or item_embedding.dim() != 2:
                raise ValueError(f"In-batch negative sampling requires 2D embeddings, got shapes "
                                 f"{user_embedding.shape} and {item_embedding.shape}")
            row0 = 0
            if self.global_negatives and self.dp is not None:
                row0 = self.dp.rank * item_embedding.size(0)
                item_embedding = sharding.gather_rows(item_embedding, self.dp.group)
            if (not self.hard_negative and ops.inbatch_logits_ok(user_embedding, item_embedding) and
                    not self.deterministic_logits):
                # random negatives read 1 + K of a row's scores: the wanted dot products straight from the tower outputs
                # (csrc/match.hip), no (B, C) score matrix and no (B, C)-sized gradients
                neg_indices = random_inbatch_negatives(user_embedding.size(0), item_embedding.size(0),
                                                       user_embedding.device, neg_ratio=self.in_batch_neg_ratio,
                                                       generator=self._sampler_generator, row_offset=row0,
                                                       stream=self.sampler_stream)
                logits = ops.inbatch_logits(user_embedding, item_embedding, neg_indices, row0)
            else:
                scores = torch.matmul(user_embedding, item_embedding.t())
                neg_indices = inbatch_negative_sampling(scores, neg_ratio=self.in_batch_neg_ratio,
                                                        hard_negative=self.hard_negative,
                                                        generator=self._sampler_generator, row_offset=row0,
                                                        stream=self.sampler_stream)
                logits = gather_inbatch_logits(scores, neg_indices, row_offset=row0)
            if self.mode == 1:
                loss = self.criterion(logits[:, 0], logits[:, 1:], in_batch_neg=True)
            elif ops.cross_entropy_ok(self.criterion, logits, None):
                # CrossEntropyLoss()(logits, zeros) (match_trainer.py:136): logsumexp - logits[:, 0], one launch each way
                loss = ops.cross_entropy_mean(logits)
            else:
                loss = self.criterion(logits, self._zero_targets(logits.size(0)))
        elif self.mode == 1:
            pos_score, neg_score = self.model(x_dict)
            loss = self.criterion(pos_score, neg_score)
        else:
            loss = self.criterion(self.model(x_dict), y)
        return self._add_reg(loss)

    def evaluate(self, model, data_loader):
        self.flush()
        model.eval()
        targets, predicts = [], []
        with torch.no_grad():
            for x_dict, y in tqdm.tqdm(data_loader, desc="validation", smoothing=0, mininterval=1.0,
                                       disable=not self.show_progress):
                y_pred = model(self._to_device(x_dict))
                targets.append(y.detach().float().reshape(-1).cpu())
                predicts.append(y_pred.detach().float().reshape(-1).cpu())
        ops.check_errors(self.device)
        return self.evaluate_fn(torch.cat(targets).numpy(), torch.cat(predicts).numpy())

    def inference_embedding(self, model, mode, data_loader, model_path):
        assert mode in ["user", "item"], "Invalid mode={}.".format(mode)
        self.flush()
        model.mode = mode
        model.load_state_dict(torch.load(os.path.join(model_path, "model.pth"), map_location=self.device,
                                         weights_only=True))
        model = model.to(self.device)
        model.eval()
        predicts = []
        with torch.no_grad():
            for x_dict in tqdm.tqdm(data_loader, desc="%s inference" % (mode), smoothing=0, mininterval=1.0,
                                    disable=not self.show_progress):
                predicts.append(model(self._to_device(x_dict)).data)
        ops.check_errors(self.device)
        return torch.cat(predicts, dim=0)

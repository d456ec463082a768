"""MTLTrainer (API mirror of torch_rechub/trainers/mtl_trainer.py:15-260) on the HIP hot path.

Same constructor, ``train_one_epoch`` -> per-task mean losses, ``fit(train, val, mode, seed)`` -> per-epoch log and
``model_{mode}_{seed}.pth``, ``evaluate`` -> per-task scores, ``predict``.  The loss is the reference's
(mtl_trainer.py:118-133): BCE / MSE per task on column i of the (B, n_task) prediction; their mean, or for ``ESMM`` the
sum of the CTR and CTCVR losses, or with ``adaptive_params={"method": "uwl"}`` the uncertainty weighting
``sum_i 2 L_i exp(-w_i) + w_i`` with w_i = clamp(weight_i, 0) learned alongside (registered on the model under the
reference's module name ``"loss weight"``).  Everything else -- dense-exact Adam over the tables, the packed gradient
bucket, RCCL data parallelism / row-sharded tables, the hipGraph step -- is inherited from CTRTrainer.

Not carried over: ``"metabalance"`` and ``"gradnorm"`` (utils/mtl.py:40-140) take one backward pass PER TASK over the
shared parameters, tables included, and rescale the per-task gradients before they are summed; with table gradients kept
as one accumulated buffer per table that needs its own kernels and is not built -- they raise NotImplementedError.
"""
import os

import torch
import tqdm
from torch import nn

from .. import ops, sharding
from ..models.multi_task import ESMM
from ..utils.data import get_loss_func, get_metric_func
from .ctr_trainer import CTRTrainer


class MTLTrainer(CTRTrainer):

    def __init__(self, model, task_types, optimizer_fn=torch.optim.Adam, optimizer_params=None,
                 regularization_params=None, scheduler_fn=None, scheduler_params=None, adaptive_params=None, n_epoch=10,
                 earlystop_taskid=0, earlystop_patience=10, device="cpu", gpus=None, model_path="./", model_logger=None,
                 **kw):
        self.task_types = task_types
        self.n_task = len(task_types)
        self.loss_weight = None
        self.adaptive_method = None
        if adaptive_params is not None:
            method = adaptive_params["method"]
            if method == "uwl":
                self.adaptive_method = "uwl"
                self.loss_weight = nn.ParameterList(nn.Parameter(torch.zeros(1)) for _ in range(self.n_task))
                model.add_module("loss weight", self.loss_weight)  # before the optimizer is built: it trains them
            elif method in ("metabalance", "gradnorm"):
                raise NotImplementedError(f"adaptive method {method!r} (per-task backward passes with gradient "
                                          "rescaling, torch_rechub/utils/mtl.py) is not built on the HIP path; use "
                                          "'uwl' or the reference trainer")
        super().__init__(model, optimizer_fn=optimizer_fn, optimizer_params=optimizer_params,
                         regularization_params=regularization_params, scheduler_fn=scheduler_fn,
                         scheduler_params=scheduler_params, n_epoch=n_epoch, earlystop_patience=earlystop_patience,
                         device=device, gpus=gpus, model_path=model_path, model_logger=model_logger, **kw)
        self.loss_fns = [get_loss_func(t) for t in task_types]
        self.evaluate_fns = [get_metric_func(t) for t in task_types]
        self.earlystop_taskid = earlystop_taskid
        self._task_loss = torch.zeros(self.n_task, dtype=torch.float32, device=self.device)

    def _prepare_target(self, y):
        return y.float()

    def _compute_loss(self, x_dict, ys):
        y_preds = self.model(x_dict)
        losses = [self._criterion_of(i, y_preds[:, i], ys[:, i]) for i in range(self.n_task)]
        self._task_loss += torch.stack([l.detach() for l in losses])  # in place: also inside a captured step
        if isinstance(self.model, ESMM):
            loss = sum(losses[1:])  # CVR is supervised only through CTCVR = CTR * CVR (entire-space training)
        elif self.adaptive_method == "uwl":
            loss = 0
            for l, w in zip(losses, self.loss_weight):
                w = torch.clamp(w, min=0)
                loss = loss + 2 * l * torch.exp(-w) + w
            loss = loss.reshape(())
        else:
            loss = sum(losses) / self.n_task
        return self._add_reg(loss)

    def _criterion_of(self, i, y_pred, y):
        fn = self.loss_fns[i]
        if ops.bce_ok(fn, y_pred, y):
            return ops.bce_mean(y_pred, y)
        return fn(y_pred, y)

    def train_one_epoch(self, data_loader, log_interval=10):
        self._task_loss.zero_()
        super().train_one_epoch(data_loader, log_interval)
        per_task = (self._task_loss / max(self._epoch_batches, 1)).tolist()
        if self.rank == 0 and self.show_progress:
            print("train loss: ", {"task_%d:" % i: v for i, v in enumerate(per_task)})
            if self.loss_weight:
                print("loss weight: ", [w.item() for w in self.loss_weight])
        return per_task

    def fit(self, train_dataloader, val_dataloader, mode="base", seed=0):
        total_log = []
        for logger in self._iter_loggers():
            logger.log_hyperparams({"n_epoch": self.n_epoch, "learning_rate": self.optimizer.param_groups[0]["lr"],
                                    "adaptive_method": self.adaptive_method})
        for epoch_i in range(self.n_epoch):
            log = self.train_one_epoch(train_dataloader)
            logs = {f"train/task_{i}_loss": v for i, v in enumerate(log)}
            logs["learning_rate"] = self.optimizer.param_groups[0]["lr"]
            if self.scheduler is not None:
                if epoch_i % self.scheduler.step_size == 0 and self.rank == 0:
                    print("Current lr : {}".format(self.optimizer.state_dict()["param_groups"][0]["lr"]))
                self.scheduler.step()
            # mean over the ranks: one early-stopping decision and one restored epoch for the whole job
            scores = [self._global_metric(v) for v in self.evaluate(self.model, val_dataloader)]
            if self.rank == 0:
                print("epoch:", epoch_i, "validation scores: ", scores)
            for i, score in enumerate(scores):
                logs[f"val/task_{i}_score"] = score
                log.append(score)
            logs["auc"] = scores[self.earlystop_taskid]
            if self.loss_weight:
                for i, w in enumerate(self.loss_weight):
                    logs[f"loss_weight/task_{i}"] = w.item()
            total_log.append(log)
            for logger in self._iter_loggers():
                logger.log_metrics(logs, step=epoch_i)
            if self.early_stopper.stop_training(scores[self.earlystop_taskid], self.model.state_dict()):
                if self.rank == 0:
                    print("validation best auc of main task %d: %.6f" % (self.earlystop_taskid,
                                                                           self.early_stopper.best_auc))
                self.model.load_state_dict(self.early_stopper.best_weights)
                break
        weights = self._checkpoint_weights()
        if self.rank == 0:
            torch.save(weights, os.path.join(self.model_path, "model_{}_{}.pth".format(mode, seed)))
        for logger in self._iter_loggers():
            logger.finish()
        return total_log

    def evaluate(self, model, data_loader):
        self.flush()
        if self.tables == "shard":
            self._check_equal_batches(data_loader, "evaluate (row-sharded tables: each batch is a collective)")
        model.eval()
        targets, predicts = [], []
        with torch.no_grad():
            for x_dict, ys in tqdm.tqdm(data_loader, desc="validation", smoothing=0, mininterval=1.0,
                                        disable=not self.show_progress):
                y_preds = model(self._to_device(x_dict))
                targets.append(ys.detach().float().cpu())
                predicts.append(y_preds.detach().float().cpu())
        ops.check_errors(self.device)
        targets, predicts = torch.cat(targets).numpy(), torch.cat(predicts).numpy()
        return [self.evaluate_fns[i](targets[:, i], predicts[:, i]) for i in range(self.n_task)]

    def predict(self, model, data_loader):
        self.flush()
        model.eval()
        predicts = []
        with torch.no_grad():
            for x_dict in tqdm.tqdm(data_loader, desc="predict", smoothing=0, mininterval=1.0,
                                    disable=not self.show_progress):
                predicts.extend(model(self._to_device(x_dict)).tolist())
        ops.check_errors(self.device)
        return predicts

"""Adam with torch.optim.Adam semantics where every embedding table is stepped by ONE HIP launch.

Reference: CTRTrainer builds ``optimizer_fn(model.parameters(), lr=1e-3, weight_decay=1e-5)``
(trainers/ctr_trainer.py:59-61) and calls ``optimizer.step()`` per batch (:99).  With
``torch.optim.Adam`` that is a DENSE update with coupled L2: every row of every table moves every step
(SURVEY Q9).  ``TableAdam`` keeps exactly that arithmetic; it only changes how it is executed:

* table parameters (``nn.Embedding`` weights with a persistent grad buffer, ops.grad_buffer) go through
  ``rh_adam_dense``: a single multi-tensor streaming kernel that reads p, g, m, v, writes p, m, v and
  re-zeroes the touched gradient rows in the same pass (no separate ``zero_grad`` traffic);
* all other parameters take the stock ``torch.optim.Adam`` code path (fused / capturable on GPU).

It IS a ``torch.optim.Adam`` (schedulers, ``state_dict`` and ``param_groups`` work unchanged).
"""
import ctypes

import torch

from . import _lib, ops


class TableAdam(torch.optim.Adam):

    def __init__(self, params, table_params=(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, **kw):
        params = list(params)
        table_ids = {id(p) for p in table_params}
        if params and isinstance(params[0], dict):
            raise ValueError("TableAdam takes a flat parameter iterable (as CTRTrainer passes model.parameters())")
        tables = [p for p in params if id(p) in table_ids and p.requires_grad]
        others = [p for p in params if not (id(p) in table_ids and p.requires_grad)]
        for p in tables:
            if not p.is_cuda or p.dtype != torch.float32 or p.numel() % 4 != 0 or not p.is_contiguous():
                raise ValueError("TableAdam: table parameters must be contiguous float32 HIP tensors (numel % 4 == 0)")
        groups = []
        if others:
            groups.append({"params": others})
        if tables:
            groups.append({"params": tables, "rh_tables": True})
        if kw.get("amsgrad") or kw.get("maximize"):
            raise ValueError("TableAdam: amsgrad / maximize are not supported on the fused table path")
        if others and others[0].is_cuda:
            kw.setdefault("capturable", True)  # device-side step counter: the step is hipGraph-capturable
            kw.setdefault("foreach", True)
        super().__init__(groups, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kw)
        self._tables = tables
        self._t_hyper_host = None
        if tables:
            dev = tables[0].device
            self._t_m = [torch.zeros_like(p) for p in tables]
            self._t_v = [torch.zeros_like(p) for p in tables]
            self._t_step = torch.zeros(1, dtype=torch.int64, device=dev)
            self._t_hyper = torch.zeros(16, dtype=torch.float64, device=dev)
            self._t_numel = (ctypes.c_int64 * len(tables))(*[p.numel() for p in tables])
            self._t_desc = None
            self._t_desc_key = None
            for p, m, v in zip(tables, self._t_m, self._t_v):
                # same keys as torch.optim.Adam so state_dict() round-trips; 'step' is synced lazily
                self.state[p] = {"step": torch.tensor(0.0), "exp_avg": m, "exp_avg_sq": v}

    # ------------------------------------------------------------------------------------
    def _table_group(self):
        for g in self.param_groups:
            if g.get("rh_tables"):
                return g
        return None

    def sync_hyper(self):
        """Upload lr/betas/eps/weight_decay of the table group if they changed (call outside graph capture)."""
        g = self._table_group()
        if g is None:
            return
        lr = g["lr"]
        lr = float(lr.item()) if torch.is_tensor(lr) else float(lr)
        host = (lr, float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))
        if host != self._t_hyper_host:
            self._t_hyper[:5].copy_(torch.tensor(host, dtype=torch.float64))
            self._t_hyper_host = host

    def _desc(self):
        grads = [ops.grad_buffer(p) for p in self._tables]
        key = tuple([p.data_ptr() for p in self._tables] + [g.data_ptr() for g in grads])
        if key != self._t_desc_key:
            rows = ([p.data_ptr() for p in self._tables] + [g.data_ptr() for g in grads] +
                    [m.data_ptr() for m in self._t_m] + [v.data_ptr() for v in self._t_v] +
                    [p.numel() for p in self._tables])
            self._t_desc = torch.tensor(rows, dtype=torch.int64).to(self._tables[0].device)
            self._t_desc_key = key
        return self._t_desc

    def step_tables(self):
        """One dense Adam step over every table (+ in-pass re-zeroing of the gradient rows)."""
        if not self._tables:
            return
        stream = ops._stream()
        desc = self._desc()
        _lib.call("rh_adam_prepare", ops._p(self._t_hyper), ops._p(self._t_step), stream)
        _lib.call("rh_adam_dense", ops._p(desc), len(self._tables), ctypes.cast(self._t_numel, ctypes.c_void_p),
                  ops._p(self._t_hyper), 1, stream)
        for p in self._tables:
            p._rh_dirty = False  # the kernel zeroed every non-zero gradient row
            if p.grad is None:
                p.grad = p._rh_grad

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._tables:
            if not torch.cuda.is_current_stream_capturing():
                self.sync_hyper()
            self.step_tables()
        dense_groups = [g for g in self.param_groups if not g.get("rh_tables")]
        if dense_groups:
            all_groups = self.param_groups
            self.param_groups = dense_groups
            try:
                super().step()
            finally:
                self.param_groups = all_groups
        return loss

    def zero_grad(self, set_to_none=True):
        """Dense params: as torch.  Tables: their gradient rows were already re-zeroed by step()."""
        dense_groups = [g for g in self.param_groups if not g.get("rh_tables")]
        all_groups = self.param_groups
        self.param_groups = dense_groups
        try:
            super().zero_grad(set_to_none=set_to_none)
        finally:
            self.param_groups = all_groups
        for p in self._tables:
            if getattr(p, "_rh_dirty", False):
                ops.grad_buffer(p).zero_()
                p._rh_dirty = False

    def state_dict(self):
        if self._tables:
            t = float(self._t_step.item())
            for p in self._tables:
                self.state[p]["step"] = torch.tensor(t)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        if self._tables:
            for i, p in enumerate(self._tables):
                st = self.state[p]
                self._t_m[i].copy_(st["exp_avg"])
                self._t_v[i].copy_(st["exp_avg_sq"])
                st["exp_avg"], st["exp_avg_sq"] = self._t_m[i], self._t_v[i]
            self._t_step.fill_(int(float(self.state[self._tables[0]]["step"])))
            self._t_hyper_host = None

"""CTRTrainer with the reference's constructor and methods (torch_rechub/trainers/ctr_trainer.py:11-187).

What changes relative to the reference loop (train_one_epoch, :77-108):
* ``optimizer_fn=torch.optim.Adam`` (the default) becomes ``optim.TableAdam``: identical arithmetic, but all
  embedding tables are stepped by one streaming HIP kernel that also re-zeroes their gradient rows;
* ``model.zero_grad()`` is one memset of the flat dense-gradient bucket (tables are re-zeroed by the step);
* the loss is accumulated on the device and read back every ``log_interval`` batches, not twice per step;
* multi-GPU = one process per GPU (torchrun) over RCCL instead of ``nn.DataParallel`` (``gpus`` is accepted for
  signature compatibility; with an initialised process group of world size > 1 the replica on this rank is
  synchronised by ``distributed.DataParallelContext``);
* with a ``DeviceDataLoader`` and ``use_graph=True`` a full batch step (batch assembly, forward, backward,
  optimizer) is captured once into a hipGraph and replayed.
"""
import os

import torch
import torch.distributed as dist
import tqdm

from .. import _lib, graphs, ops, sharding
from ..basic.callback import EarlyStopper
from ..basic.loss_func import RegularizationLoss
from ..distributed import DataParallelContext, DenseGradBucket, table_parameters
from ..optim import TableAdam
from ..utils.data import DeviceDataLoader

DP_FUSED_HEAD = _lib.ab("dphead")  # False (RECHUB_AB=dphead=0): batch assembly and refresh as two launches under data parallelism


class CTRTrainer(object):

    def __init__(self, model, optimizer_fn=torch.optim.Adam, optimizer_params=None, regularization_params=None,
                 scheduler_fn=None, scheduler_params=None, n_epoch=10, earlystop_patience=10, device="cpu", gpus=None,
                 loss_mode=True, model_path="./", model_logger=None, use_graph=None, show_progress=True,
                 table_update=None, lazy_k=None, tables=None, shard_min_rows=0, lazy_small_rows=None):
        self.model = model
        self.gpus = [] if gpus is None else gpus
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("torch_rechub_amd.CTRTrainer drives the HIP hot path: device must be a HIP device "
                               f"('cuda:N'), got {device!r}. Use the reference trainer for CPU runs.")
        self.model.to(self.device)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        # RECHUB_FORCE_DP=1 runs the full data-parallel machinery (RCCL calls included) on a world of one
        force_dp = os.environ.get("RECHUB_FORCE_DP", "0") == "1" and dist.is_available() and dist.is_initialized()
        # tables: "replicate" = one replica per rank, gradient rows exchanged (DataParallel semantics, the default);
        # "shard" = one row-shard per rank (sharding.py): table memory, optimizer state and sweep traffic / world.
        # Either way the step computes the reference's global-batch update.
        if tables is None:
            tables = sharding.placement_from_env()
        if tables not in ("replicate", "shard"):
            raise ValueError("tables must be 'replicate' or 'shard'")
        if tables == "shard" and not (self.world > 1 or force_dp):
            raise RuntimeError("tables='shard' needs an initialised process group with more than one rank "
                               "(or RECHUB_FORCE_DP=1 for a one-rank dry run)")
        self.tables = tables
        self.dp = None
        if self.world > 1 or force_dp:
            self.dp = DataParallelContext(self.model, force=force_dp, shard_tables=(tables == "shard"),
                                          shard_min_rows=shard_min_rows)
        if optimizer_params is None:
            optimizer_params = {"lr": 1e-3, "weight_decay": 1e-5}
        tables = table_parameters(self.model)
        if regularization_params is None:
            regularization_params = {"embedding_l1": 0.0, "embedding_l2": 0.0, "dense_l1": 0.0, "dense_l2": 0.0}
        # table_update: "lazy" = blocked-lazy exact Adam (bit-identical to "dense", ~1/lazy_k of its HBM traffic),
        # "dense" = every row every step.  An embedding regulariser adds a dense gradient term -> dense mode.
        # lazy_k = 128 (round 4; 64 before): the window sweep replays the same number of element-steps per training step
        # whatever lazy_k is, but touches half as many rows twice as long -- half its HBM traffic beside the step's chain --
        # while the pre-gather refresh of the batch's rows replays twice as far.  Measured on the headline step (same box,
        # step-ahead form): lazy_k 32 / 64 / 96 / 128 / 160 / 192 / 256 / 384 = 0.282 / 0.255 / 0.247 / 0.244 / 0.250 / 0.251 /
        # 0.256 / 0.276 ms.  The refresh's share grows with the batch (B = 32768: 0.850 ms at 64, 0.901 at 128), so lazy_k = None
        # (the default) lets the optimizer take 128 for steps of up to 8192 samples and 64 beyond (decided at its first step,
        # while every row is still current); an explicit lazy_k is kept.
        if table_update is None:
            table_update = os.environ.get("RECHUB_TABLE_ADAM", "lazy")
        if table_update not in ("lazy", "dense"):
            raise ValueError("table_update must be 'lazy' or 'dense'")
        if regularization_params.get("embedding_l1", 0) > 0 or regularization_params.get("embedding_l2", 0) > 0:
            table_update = "dense"
        self.table_update = table_update
        if optimizer_fn is torch.optim.Adam and not optimizer_params.get("amsgrad", False):
            optimizer_params = dict(optimizer_params)
            if lazy_small_rows is not None:  # tables up to this many rows take the dense pass (K = 1) in lazy mode
                optimizer_params["lazy_small_rows"] = lazy_small_rows
            self.optimizer = TableAdam(self.model.parameters(), table_params=tables,
                                       lazy_k=((128 if lazy_k is None else lazy_k) if table_update == "lazy" else 0),
                                       lazy_k_auto=lazy_k is None, **optimizer_params)
        else:
            self.optimizer = optimizer_fn(self.model.parameters(), **optimizer_params)
        table_ids = {id(p) for p in tables}
        self.short_sweep_inline = False
        if (self.dp is not None and self.tables == "shard" and isinstance(self.optimizer, TableAdam) and
                not os.environ.get("RECHUB_STEP_FORM")):
            # a rank's shard of the tables may be small enough for the window sweep to go back in line (optim.py,
            # SHORT_SWEEP_ELEMENTS: from four ranks up on the Criteo-shape tables; -11 % / -14 % per-rank step at 4 / 8 ranks)
            self.short_sweep_inline = self.optimizer.prefer_inline_for_short_sweeps(auto_k=lazy_k is None)
        if self.dp is not None and self.tables == "shard" and isinstance(self.optimizer, TableAdam):
            # Row-sharded step: its head (batch assembly, index all-gather, localisation, refresh of the shard's rows) stays on the
            # chain's queue and only the sweep crosses to its own -- the sharded chain (lookup exchange + reduce-scatter on top of
            # the N = 1 chain) is the longer of the two paths, so the cross-queue edge belongs on the sweep's side: one rank
            # 0.3394 / 0.3407 / 0.3393 -> 0.3191 / 0.3196 / 0.3203 ms same box (round 6; the N = 1 DSSM / DCN-v2 steps, whose
            # sweep or library GEMMs are the bound, lose 2 % / 8 % by the same switch and keep the head on the sweep's queue)
            self.optimizer.head_on_side = False
            # ... and forks its sweep behind a gate instead of at a segment boundary (optim.TableAdam._cut_fork) when every table
            # is sharded (no replicated small table whose gathered rows would ask for a join inside the step)
            self.optimizer.gated_fork = shard_min_rows <= 0
        if self.dp is not None and (self.tables == "replicate" or shard_min_rows > 0) and getattr(self.optimizer, "lazy_k", 0) > 1:
            # the gradient-row exchange hands this rank the rows of every rank's batch: TableAdam._join_before_foreign_rows
            self.optimizer.foreign_rows = True
        if self.dp is not None:
            self.bucket = self.dp.bucket
        else:
            self.bucket = DenseGradBucket([p for p in self.model.parameters() if id(p) not in table_ids])
        self._bucket_attached = False
        self.scheduler = None
        if scheduler_fn is not None:
            self.scheduler = scheduler_fn(self.optimizer, **scheduler_params)
        self.loss_mode = loss_mode
        self.criterion = torch.nn.BCELoss()
        from sklearn.metrics import roc_auc_score
        self.evaluate_fn = roc_auc_score
        self.n_epoch = n_epoch
        self.early_stopper = EarlyStopper(patience=earlystop_patience)
        self.model_path = model_path
        self.reg_loss_fn = RegularizationLoss(**regularization_params)
        self.model_logger = model_logger
        if use_graph is None:
            use_graph = os.environ.get("RECHUB_HIPGRAPH", "0") == "1"
        self.use_graph = bool(use_graph)
        # data-parallel hipGraph layout: "split" = [graph A] -> eager RCCL -> [graph B]; "single" = collectives captured
        # (default; RCCL >= 2.9 launches are capturable; falls back to "split" if the capture raises)
        self.dp_graph = os.environ.get("RECHUB_DP_GRAPH", "single")
        self.show_progress = show_progress and self.rank == 0
        self._graph = None
        self._graph_b = None
        self._graph_loss = None
        self._deferred_static = None
        self._counters = []  # device counters of the running step (loader position ...): advanced by rh_step_scalars

    # -- one optimisation step ----------------------------------------------------------------
    def _zero_grad(self):
        self.bucket.zero()
        if not isinstance(self.optimizer, TableAdam):
            for p in table_parameters(self.model):
                if getattr(p, "_rh_dirty", False):
                    ops.grad_buffer(p).zero_()
                    p._rh_dirty = False

    def _prepare_target(self, y):
        return y.float()

    def _load(self, loader, B=None):
        """Next batch of a DeviceDataLoader; its position counter is advanced by this step's scalar launch."""
        opt = self.optimizer
        # (data parallel, replicated tables: the one-kernel head in its strict form; row-sharded tables localise the gathered
        # global indices first and keep the two launches)
        dp_ok = self.dp is None or (DP_FUSED_HEAD and not self.dp.sharded)
        fused = isinstance(opt, TableAdam) and dp_ok and opt.assemble_with_refresh(loader, B, strict=self.dp is not None)
        x, y = loader.load_next(B, advance=False, assemble=not fused)
        self._counters.append(loader.counter(loader.batch_size if B is None else B))
        return x, y

    def _forward_loss(self, x_dict, y, defer_scalars=False):
        """_compute_loss under an armed ops.StepFusion: head + BCE terms in one launch, the loss mean / Adam bias
        corrections / device counters in ONE scalar launch (ops.StepFusion).  Anything the fusion did not absorb is
        launched here."""
        fuse = (isinstance(self.optimizer, TableAdam) and self.loss_mode and
                type(self)._compute_loss is CTRTrainer._compute_loss and type(self)._criterion is CTRTrainer._criterion
                and type(self.criterion) is torch.nn.BCELoss)
        counters, self._counters = self._counters, []
        if isinstance(self.optimizer, TableAdam):
            self.optimizer.rollback_abandoned_prepare()
        ops.fusion_begin(target=y if fuse else None, optimizer=self.optimizer if fuse else None, counters=counters,
                         defer_scalars=defer_scalars and fuse)
        try:
            loss = self._compute_loss(x_dict, y)
        finally:
            left = ops.fusion_end()
        ops.advance_counters(left)
        return loss

    def _compute_loss(self, x_dict, y):
        """model forward + criterion + regularisation (trainers/ctr_trainer.py:86-95); overridden by MatchTrainer."""
        if self.loss_mode:
            loss = self._criterion(self.model(x_dict), y)
        else:
            y_pred, other_loss = self.model(x_dict)
            loss = self._criterion(y_pred, y) + other_loss
        return self._add_reg(loss)

    def _add_reg(self, loss):
        """loss + RegularizationLoss (ctr_trainer.py:92-95); the reference adds the python float 0.0 when every
        coefficient is zero -- which would be one add-scalar launch per step here."""
        reg = self.reg_loss_fn(self.model)
        return loss if (isinstance(reg, float) and reg == 0.0) else loss + reg

    def _criterion(self, y_pred, y):
        # torch.nn.BCELoss() (the reference default, ctr_trainer.py:62) runs as one HIP launch each way
        if ops.bce_ok(self.criterion, y_pred, y):
            return ops.bce_mean(y_pred, y)
        return self.criterion(y_pred, y)

    def _scale_for_world(self, loss):
        """Gradients of the data loss and of the dense regulariser are SUMMED over the ranks (row exchange / dense
        all-reduce), so the loss is scaled by 1/world: the global-batch mean, as DataParallel.  The embedding
        regulariser's gradient is a LOCAL dense term on each replica / shard and keeps its full strength."""
        self._root_scale = 1.0
        if self.world <= 1:
            return loss
        emb_reg = self.reg_loss_fn.embedding_term(self.model)
        if isinstance(emb_reg, float) and emb_reg == 0.0:
            # no embedding regulariser (the default): the factor 1/world rides on the ROOT of the backward (_grad_root) instead
            # of a division + an add-zero in the forward and their backward -- three launches of ~5 us per step at N > 1
            self._root_scale = 1.0 / self.world
            return loss
        return loss / self.world + (1.0 - 1.0 / self.world) * emb_reg

    def train_step(self, x_dict, y):
        """forward + loss + backward + optimizer step on device tensors; returns the detached loss tensor."""
        # (defer_scalars: the backward below follows at once and the loss value is read after it -- the fused MLP chain's
        # head backward may then carry the step's scalar launch, i.e. the loss buffer is WRITTEN during the backward.  Only
        # when nothing consumes the loss value in the forward: an active regulariser adds its penalty to it there
        # (_add_reg) and would read the buffer before it is filled -- round-4 advisor finding: wrong reported loss with
        # dense_l1 / dense_l2 > 0 on one GPU)
        loss = self._forward_loss(x_dict, y, defer_scalars=self.dp is None and not self.reg_loss_fn.active())
        report = loss.detach()
        loss = self._scale_for_world(loss)
        self._zero_grad()
        fast = isinstance(self.optimizer, TableAdam)
        packed = fast and self.optimizer._bucket is not None
        # single GPU, packed optimizer: the weight-gradient / head / LR backward kernels leave their partial slabs to the
        # step's ONE packing launch (ops.DeferredGrads) instead of reducing them one by one
        defer = packed and self.dp is None
        # data parallel (round 5): the same slabs, summed by the bucket's own packing launches (DenseGradBucket.flush: one at
        # the start of the embedding backward -- the all-reduce of the MLP's gradients starts on it -- one for the rest)
        defer_dp = packed and self.dp is not None and self.bucket.use_cuda
        if defer or defer_dp:
            # (data parallel: the bucket packs slabs in the middle of the backward -- count the uses of every parameter first)
            ops.deferred.arm(self.bucket.params, root=loss if defer_dp else None)
        try:
            loss.backward(self._grad_root(loss))
            if defer_dp:
                self.bucket.finish(assign_views=False)
        finally:
            items = ops.deferred.disarm() if (defer or defer_dp) else {}
        if fast and not self._bucket_attached:
            # first step: every dense parameter must receive a gradient for the packed one-launch optimizer path
            # (torch.optim.Adam skips parameters without a gradient; the packed path cannot)
            self._bucket_attached = True
            if self.bucket.all_present() and self.bucket.params:
                self.optimizer.attach_bucket(self.bucket)
            packed = self.optimizer._bucket is not None  # attached in THIS step: pack it the plain way below
        if packed and not defer_dp and not all(p.grad is not None or id(p) in items for p in self.bucket.params):
            raise RuntimeError("a dense parameter stopped receiving gradients; rebuild the trainer")
        late_gate = None
        if defer:
            # with this step's Adam scalars already on the device (the step's scalar launch computed them in the forward),
            # the packing launch also steps the dense parameters: rh_pack_grads + rh_adam_small as ONE launch
            adam = self.optimizer.small_adam_args()
            # step-ahead form being captured: the packing launch goes BEHIND the end-of-step table launch (the two are
            # independent) and opens the deferred sweep's gate when it starts, instead of a launch of its own doing that
            late_gate = self.optimizer.gate_for_late_pack() if adam is not None else None
            if late_gate is None:
                self.bucket.pack(items, adam=adam)
        elif (packed or self.dp is not None) and not defer_dp:
            self.bucket.finish(assign_views=not packed)
        self.optimizer.step()
        if late_gate is not None:
            self.bucket.pack(items, adam=adam, gate=late_gate)
        return report

    def _grad_root(self, loss):
        """d loss / d loss = 1 as a cached tensor: autograd's implicit ones_like is a fill launch per step.  With more than
        one rank (and no embedding regulariser) the root is 1/world: the loss scaling of _scale_for_world."""
        scale = float(getattr(self, "_root_scale", 1.0))
        one = getattr(self, "_one", None)
        capturing = torch.cuda.is_current_stream_capturing()
        if one is None or one.device != loss.device or one.shape != loss.shape:
            if capturing:
                # (not cached yet: never the case after the eager warm-up steps in front of a capture)
                return None if scale == 1.0 else torch.full_like(loss, scale)
            one = torch.full_like(loss, scale)
            self._one, self._one_scale = one, scale
        elif getattr(self, "_one_scale", 1.0) != scale:
            # ONE persistent root tensor per trainer: the fused head / BCE kernels of a captured step read it BY POINTER, so
            # it is refilled in place, never replaced (round-5 advisor finding) -- and a captured step baked the other
            # scale's whole forward (the embedding regulariser's terms), so it cannot be replayed after such a change
            if capturing or getattr(self, "_graph", None) is not None:
                raise RuntimeError("the loss scale of the data-parallel step changed (embedding regulariser switched on or "
                                   "off) under a captured hipGraph step; build a new trainer")
            one.fill_(scale)
            self._one_scale = scale
        return one

    # -- data-parallel step in three phases: [A: batch, forward, backward, pack] -> [X: RCCL collectives] ->
    #    [B: scatter gathered rows, optimizer].  Under hipGraph the three are captured as ONE graph (dp_graph =
    #    "single"), or A and B as two graphs with the collectives launched eagerly between them ("split"). ---------
    def _phase_a(self, x_dict, y):
        self.dp.deferred_mode, self.dp.deferred = True, []
        self.bucket.defer = True
        # (defer_scalars as in train_step: the loss value is read after the backward, which follows at once -- the chain's head
        # backward carries the step's scalar launch; not with an active regulariser, which reads the loss in the forward)
        loss = self._forward_loss(x_dict, y, defer_scalars=not self.reg_loss_fn.active())
        report = loss.detach()
        loss = self._scale_for_world(loss)
        self._zero_grad()
        defer_dp = self.optimizer._bucket is not None and self.bucket.use_cuda
        if defer_dp:
            ops.deferred.arm(self.bucket.params, root=loss)
        try:
            loss.backward(self._grad_root(loss))
            ops.deferred.backward_done()
            if not self._bucket_attached:
                self._bucket_attached = True
                if self.bucket.all_present() and self.bucket.params:
                    self.optimizer.attach_bucket(self.bucket)
            if self.optimizer._bucket is None or not self.bucket.all_present():
                raise RuntimeError("split-graph data parallel step needs every dense parameter to receive a gradient")
            self.bucket.flush()  # pack only (defer = True)
        finally:
            if defer_dp:
                ops.deferred.disarm()
        self.dp.deferred_mode = False
        return report, list(self.dp.deferred)

    def _phase_x(self, deferred):
        self.bucket.reduce_deferred()  # dense all-reduce on the side stream ...
        gathered = self.dp.exchange_deferred(deferred)  # ... overlapping the sparse all-gathers
        self.bucket.join()
        return gathered

    def _phase_b(self, gathered):
        for call, idx_all, rows_all in gathered:
            ops.scatter_rows(call, idx_all, rows_all)
            for w in {id(w): w for w in call.weights if w.requires_grad}.values():
                ops._publish_grad(w)
        self.bucket.defer = False
        self.optimizer.step()

    def _split_step(self, x_dict, y):
        report, deferred = self._phase_a(x_dict, y)
        self._phase_b(self._phase_x(deferred))
        return report

    GRAPH_WARMUP = 3

    # Candidates of the step's self-tuning: (form of the captured step, persistent workgroups of the side-stream sweep:
    # RH_TUNE_DEFERRED_GRID, 512 / 256 = 2 / 1 per CU).  Forms: "deferred" = join -> [assembly, refresh] -> fork sweep ->
    # [rest of the step] (two graph segments); "inline" = the merged end-of-step launch.  The in-line form is always a
    # candidate: a deferred sweep pays when the step's chain is latency-bound (DeepFM / DSSM at B = 4096), less when its
    # kernels are heavy themselves (DIN, B = 65536: the sweep slows them by what it hides).  Round 3 also built a
    # "pipelined" form (0.37-0.39 ms where "deferred" reaches 0.305) and a "branch" form (0.312 ms); both were removed
    # in round 4, DESIGN 4.3 keeps their numbers and timelines.
    # (form, residency cap of the deferred sweep in workgroups, hold-back of the sweep behind the end of the step's graph in ns
    # -- step-ahead form, rh_adam_sweep_gate: WHERE in the next step's chain the sweep's workgroups are dispatched decides
    # whether they spread evenly over the SIMDs; 22 us was a 305 us step where 28 us was a 245 us one, tools/period_hist.py)
    # (the opening is counted by the packing launch, ~5 us before the graph ends: 32 / 40 us here = 28 / 36 us behind a separate
    # opening launch, where the landscape was measured)
    # (round 6, steps WITHOUT a chain start to release the sweep -- DCN-v2, DSSM: a hold-back of 150 us moves the sweep off the
    # step's first library GEMMs, which run at the sweep's own wave priority and take 4 x their time beside it: DCN-v2
    # 0.6346 -> 0.6162 ms on one box, 280 us 0.6199, 400 us 0.6820)
    TUNE_CANDIDATES = (("deferred", 512, 32000), ("deferred", 512, 40000), ("deferred", 256, 32000), ("deferred", 256, 150000),
                       ("inline", 0, 0))
    TUNE_SETTLE, TUNE_STEPS = 6, 16

    def tune_budget_steps(self):
        """Upper bound of the replayed steps the self-tuning below takes once the optimizer is in its steady state (bench.py
        keeps its timed region behind it)."""
        return len(self.TUNE_CANDIDATES) * (self.TUNE_SETTLE + self.TUNE_STEPS)

    def _tune_step_form(self, loader):
        """Self-tuning of HOW the captured step ends, over real training steps (nothing is thrown away): once the lazy
        optimizer is in its steady state (lazy_k + 8 replays after the capture: the window sweeps have their full
        length), every candidate of TUNE_CANDIDATES runs TUNE_SETTLE + TUNE_STEPS steps bracketed by HIP events, then ONE
        event synchronisation picks the fastest.  The residency cap is a parameter of the EAGER side-stream launch and can
        change between replays of one graph; the other forms are further captures of the same step (their own graphs, same
        arithmetic: the optimizer's bit-equality tests cover all of them).  RECHUB_STEP_FORM=deferred|inline and
        RECHUB_SWEEP_GRID=workgroups pin the choice; data-parallel steps keep the configured form."""
        opt = self.optimizer
        st = getattr(self, "_tune", None)
        if st is None:
            from .. import _lib
            lazy = isinstance(opt, TableAdam) and getattr(opt, "lazy_k", 0) > 1 and bool(opt._tables)
            form = os.environ.get("RECHUB_STEP_FORM", "")
            if form not in ("", "deferred", "inline"):
                raise ValueError(f"RECHUB_STEP_FORM={form!r}: 'deferred' or 'inline' (the round-3 forms 'pipelined' and "
                                 "'branch' were removed)")
            grid = os.environ.get("RECHUB_SWEEP_GRID", "")
            if grid:
                _lib.call("rh_set_tuning", 8, int(grid))
            # (round 5 measured larger residency caps for the sweep-bound configs[4] step -- 1024 / 2048 workgroups: 0.98 / 1.04 ms
            # against 0.834 at 512 and 0.833 at 256 -- the chain loses more than the sweep gains there too; not candidates)
            cands = [c for c in self.TUNE_CANDIDATES if (not form or c[0] == form) and
                     (not grid or c[0] == "inline" or c[1] == int(grid))]
            if form and (grid or not cands):  # fully pinned (also forms / grids that are not tuning candidates)
                cands = [(form, int(grid or 512) if form != "inline" else 0, 0)]
            # a user who pinned the deferred sweep's grid / hold-back through RECHUB_TUNE (keys 8 / 13, exact match) keeps them
            pinned = {kv.split("=")[0].strip() for kv in os.environ.get("RECHUB_TUNE", "").split(",") if "=" in kv}
            if "13" in pinned or form and (grid or len(cands) == 1):
                seen, kept = set(), []
                for c in cands:  # one candidate per (form, grid): the hold-back stays what the user / the library set
                    if c[:2] not in seen:
                        seen.add(c[:2])
                        kept.append((c[0], c[1], 0))
                cands = kept
            if lazy and getattr(opt, "gate_by_chain", False):
                # the captured step counts its chain starts: the sweep is released by the next step's first GEMM, the hold-back
                # is no dimension of the step any more (round 4: 22 us = 305 us steps, 28 us = 245 us, box-dependent)
                seen, kept = set(), []
                for c in cands:
                    if c[:2] not in seen:
                        seen.add(c[:2])
                        kept.append((c[0], c[1], 0))
                cands = kept
            active = lazy and self.dp is None and len(cands) > 1 and "8" not in pinned
            st = self._tune = {"active": bool(active), "wait": (opt.lazy_k + 8) if lazy else 0, "i": 0, "n": 0, "ev": [],
                               "cands": cands}
            if lazy and self.dp is None and form and form != self._form and len(cands) == 1:
                self._apply_candidate(cands[0], loader)
        if not st["active"]:
            return
        if st["wait"] > 0:
            st["wait"] -= 1
            return
        per = self.TUNE_SETTLE + self.TUNE_STEPS
        i, n = st["i"], st["n"]
        if n == 0:
            self._apply_candidate(st["cands"][i], loader)
        if n == self.TUNE_SETTLE:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            st["ev"].append([e0, None])
        st["n"] = n + 1
        if st["n"] == per:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()  # in front of this step's replay: TUNE_STEPS - 1 whole steps since e0, for every candidate alike
            st["ev"][i][1] = e1
            st["i"], st["n"] = i + 1, 0
            if st["i"] == len(st["cands"]):
                times = []
                for e0, e1 in st["ev"]:
                    e1.synchronize()
                    times.append(e0.elapsed_time(e1))
                best = min(range(len(times)), key=times.__getitem__)
                st.update(active=False, chosen=st["cands"][best], ms=[round(t / (self.TUNE_STEPS - 1), 4) for t in times])
                st["pending"] = st["cands"][best]  # applied in front of the NEXT replay (this one still belongs to the last candidate)
        return

    def _apply_candidate(self, cand, loader):
        from .. import _lib
        form, grid, hold = cand
        if form != "inline":
            _lib.call("rh_set_tuning", 8, int(grid))
            if hold:
                _lib.call("rh_set_tuning", 13, int(hold))
        if self._form != form:
            self._switch_form(form, loader)

    def _switch_form(self, form, loader):
        """Continue with another form of the captured step, capturing it on first use.  The switch happens between two
        steps from a settled sweep state, so every graph sees the same invariants."""
        opt = self.optimizer
        opt.settle_sweep()
        self._counters = []
        forms = self.__dict__.setdefault("_graph_forms", {})
        forms[self._form] = (self._graph, self._graph_loss)
        self._form = form
        opt.overlap_sweep = form != "inline"
        if form in forms:
            self._graph, self._graph_loss = forms[form]
            return
        g = graphs.SegmentedGraph()

        def whole_step():
            x, y = self._load(loader)
            return self.train_step(x, y)

        self._graph_loss = g.capture(whole_step)
        self._graph = g

    def _graphed_step(self, loader):
        """Replay the captured (batch assembly + train_step); the first call warms up eagerly and captures.

        Returns (sum of losses, number of batches consumed).  Warm-up steps are real optimisation steps.
        """
        split = self.dp is not None
        if split and not isinstance(self.optimizer, TableAdam):
            raise RuntimeError("hipGraph + data parallel needs the default Adam optimizer (TableAdam)")
        if self._graph is not None and getattr(self, "_graph_loader", loader) is not loader:
            # the captured kernels read the static batch buffers of the loader they were captured with; the trainer
            # keeps that loader alive, but batches of another loader would silently never be seen
            raise RuntimeError("this trainer's hipGraph step was captured with another DeviceDataLoader; keep using that "
                               "loader (reshuffle() / new epochs are fine) or build a new trainer")
        if self._graph is None:
            self._graph_loader = loader
            if isinstance(self.optimizer, TableAdam):
                self.optimizer.sync_hyper()
            total = torch.zeros((), dtype=torch.float32, device=self.device)
            side = graphs.role_stream("warmup", self.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(self.GRAPH_WARMUP):  # allocator, descriptor caches, lazily created optimizer state
                    x, y = self._load(loader)
                    total += self._split_step(x, y) if split else self.train_step(x, y)
            torch.cuda.current_stream().wait_stream(side)
            # Segmented capture: the optimizer cuts the step where it launches the deferred table sweep eagerly on
            # its side stream (graphs.SegmentedGraph); without cuts this is one ordinary hipGraph.
            if hasattr(self.optimizer, "settle_sweep"):
                self.optimizer.settle_sweep()
            self._graph = graphs.SegmentedGraph()

            def whole_step():
                x, y = self._load(loader)
                return self._split_step(x, y) if split else self.train_step(x, y)

            self._form = "deferred" if getattr(self.optimizer, "overlap_sweep", False) else "inline"
            if not split:
                self._graph_loss = self._graph.capture(whole_step)
                return total, self.GRAPH_WARMUP
            if self.dp_graph == "single":
                # RCCL collectives captured with the rest of the step: no eager launches but the table sweep
                try:
                    self._graph_loss = self._graph.capture(whole_step)
                    self._graph_b = None
                    return total, self.GRAPH_WARMUP
                except RuntimeError as e:  # collective not capturable with this RCCL build: use the split layout
                    import warnings
                    warnings.warn(f"hipGraph capture of the RCCL collectives failed ({e}); using the split-graph step")
                    torch.cuda.synchronize()
                    self.dp_graph = "split"
                    self.bucket.defer = False
                    self.dp.deferred_mode = False
            if getattr(self.optimizer, "lazy_k", 0) > 1:
                self.optimizer._join_sweep()  # plain captures cannot hold the eager side-stream sweep:
                self.optimizer.overlap_sweep = False  # from here on the window sweep runs in line
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                x, y = self._load(loader)
                self._graph_loss, self._deferred_static = self._phase_a(x, y)
            self._graph.replay()  # capture does not execute: run A for real, exchange, then capture + run B
            gathered = self._phase_x(self._deferred_static)
            self._graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_b, pool=self._graph.pool(), capture_error_mode="thread_local"):
                self._phase_b(gathered)
            self._graph_b.replay()
            return total + self._graph_loss, self.GRAPH_WARMUP + 1
        self._tune_step_form(loader)
        pend = self._tune.pop("pending", None)
        if pend is not None:
            self._apply_candidate(pend, loader)
        self._graph.replay()
        if split and self._graph_b is not None:
            self.bucket._deferred_runs = [(0, len(self.bucket.params))]
            self._phase_x(self._deferred_static)
            self._graph_b.replay()
        return self._graph_loss, 1

    def _check_equal_batches(self, data_loader, what):
        """Every rank must run the same number of equally sized batches: each step (and, with row-sharded tables,
        each evaluation batch) is a collective.  Raises on EVERY rank alike when they disagree."""
        if self.world <= 1 or not hasattr(data_loader, "__len__"):
            return
        n = len(data_loader)
        bs = int(getattr(data_loader, "batch_size", 0) or 0)
        rows = getattr(data_loader, "N", None)
        if rows is None and hasattr(data_loader, "dataset") and hasattr(data_loader.dataset, "__len__"):
            rows = len(data_loader.dataset)
        mine = torch.tensor([n, -n, bs, -bs, int(rows or 0), -int(rows or 0)], dtype=torch.int64, device=self.device)
        dist.all_reduce(mine, op=dist.ReduceOp.MIN)
        lo_n, hi_n, lo_b, hi_b, lo_r, hi_r = (int(v) for v in mine.tolist())
        if lo_n != -hi_n or lo_b != -hi_b or lo_r != -hi_r:
            raise RuntimeError(f"{what}: the ranks disagree on the loader (batches {lo_n}..{-hi_n}, batch size "
                               f"{lo_b}..{-hi_b}, rows {lo_r}..{-hi_r}); every step is a collective -- give each rank "
                               "the same number of rows (DeviceDataLoader.from_parquet truncates to the minimum)")

    def _global_metric(self, value):
        """Mean of a per-rank validation metric over the ranks: every rank then takes the same early-stopping decision
        and restores the same epoch (a per-rank decision would leave some ranks inside the next collective)."""
        if self.world <= 1:
            return value
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item()) / self.world

    def train_one_epoch(self, data_loader, log_interval=10):
        self.model.train()
        self._check_equal_batches(data_loader, "train_one_epoch")
        device_loader = isinstance(data_loader, DeviceDataLoader)
        if isinstance(self.optimizer, TableAdam):
            self.optimizer.sync_hyper()
        run = torch.zeros((), dtype=torch.float32, device=self.device)
        epoch = torch.zeros((), dtype=torch.float32, device=self.device)
        batch_count = 0
        full = data_loader.N // data_loader.batch_size if device_loader else 0
        if device_loader and self.use_graph and (self._graph is not None or full > self.GRAPH_WARMUP):
            data_loader.reshuffle()
            rem = data_loader.N - full * data_loader.batch_size
            it = tqdm.tqdm(total=full, desc="train", smoothing=0, mininterval=1.0, disable=not self.show_progress)
            since_log = 0
            while batch_count < full:
                if self._graph is None and full - batch_count <= self.GRAPH_WARMUP:
                    x, y = self._load(data_loader)
                    loss, n = self.train_step(x, y), 1
                else:
                    loss, n = self._graphed_step(data_loader)
                run += loss
                epoch += loss
                batch_count += n
                since_log += n
                it.update(n)
                if since_log >= log_interval:
                    it.set_postfix(loss=run.item() / since_log)
                    run.zero_()
                    since_log = 0
            it.close()
            if rem and not data_loader.drop_last:
                x, y = self._load(data_loader, rem)
                epoch += self.train_step(x, y)
                batch_count += 1
        else:
            it = tqdm.tqdm(data_loader, desc="train", smoothing=0, mininterval=1.0, disable=not self.show_progress)
            for i, (x_dict, y) in enumerate(it):
                if not device_loader:
                    x_dict = {k: v.to(self.device, non_blocking=True) for k, v in x_dict.items()}
                    y = y.to(self.device, non_blocking=True)
                loss = self.train_step(x_dict, self._prepare_target(y))
                run += loss
                epoch += loss
                batch_count += 1
                if (i + 1) % log_interval == 0:
                    it.set_postfix(loss=run.item() / log_interval)
                    run.zero_()
        self.flush()
        ops.check_errors(self.device)
        self._epoch_batches = batch_count
        return epoch.item() / batch_count if batch_count > 0 else 0

    def flush(self):
        """Bring lazily updated table rows up to date (no-op otherwise); runs before anything reads the weights."""
        if isinstance(self.optimizer, TableAdam):
            self.optimizer.flush()

    # -- epochs -------------------------------------------------------------------------------
    def fit(self, train_dataloader, val_dataloader=None):
        for logger in self._iter_loggers():
            logger.log_hyperparams({"n_epoch": self.n_epoch, "learning_rate": self.optimizer.param_groups[0]["lr"],
                                    "loss_mode": self.loss_mode})
        for epoch_i in range(self.n_epoch):
            if self.rank == 0:
                print("epoch:", epoch_i)
            train_loss = self.train_one_epoch(train_dataloader)
            for logger in self._iter_loggers():
                logger.log_metrics({"train/loss": train_loss, "learning_rate": self.optimizer.param_groups[0]["lr"]},
                                   step=epoch_i)
            if self.scheduler is not None:
                if epoch_i % self.scheduler.step_size == 0 and self.rank == 0:
                    print("Current lr : {}".format(self.optimizer.state_dict()["param_groups"][0]["lr"]))
                self.scheduler.step()
            if val_dataloader:
                auc = self._global_metric(self.evaluate(self.model, val_dataloader))
                if self.rank == 0:
                    print("epoch:", epoch_i, "validation: auc:", auc)
                for logger in self._iter_loggers():
                    logger.log_metrics({"val/auc": auc}, step=epoch_i)
                if self.early_stopper.stop_training(auc, self.model.state_dict()):
                    if self.rank == 0:
                        print(f"validation: best auc: {self.early_stopper.best_auc}")
                    self.model.load_state_dict(self.early_stopper.best_weights)
                    break
        # row-sharded tables are reassembled (a collective): the file has the reference's layout whatever the placement
        weights = self._checkpoint_weights()
        if self.rank == 0:
            torch.save(weights, os.path.join(self.model_path, "model.pth"))
        for logger in self._iter_loggers():
            logger.finish()

    def _checkpoint_weights(self):
        """The state_dict every trainer's fit() writes: row-sharded tables reassembled (a collective) into the reference's
        layout, and compact tensors -- a padded-width table (PaddedEmbedding) exposes its (vocab, embed_dim) VIEW, and
        torch.save of a view would write the whole padded storage behind it."""
        weights = sharding.full_state_dict(self.model) if self.tables == "shard" else self.model.state_dict()
        return {k: (v if not torch.is_tensor(v) or v.is_contiguous() else v.contiguous()) for k, v in weights.items()}

    def _iter_loggers(self):
        if self.model_logger is None or self.rank != 0:
            return []
        if isinstance(self.model_logger, (list, tuple)):
            return list(self.model_logger)
        return [self.model_logger]

    def _to_device(self, x_dict):
        return {k: (v if v.is_cuda else v.to(self.device, non_blocking=True)) for k, v in x_dict.items()}

    def evaluate(self, model, data_loader):
        self.flush()
        if self.tables == "shard":
            self._check_equal_batches(data_loader, "evaluate (row-sharded tables: each batch is a collective)")
        model.eval()
        targets, predicts = [], []
        with torch.no_grad():
            it = tqdm.tqdm(data_loader, desc="validation", smoothing=0, mininterval=1.0,
                           disable=not self.show_progress)
            for x_dict, y in it:
                y_pred = model(self._to_device(x_dict)) if self.loss_mode else model(self._to_device(x_dict))[0]
                targets.append(y.detach().float().reshape(-1).cpu())
                predicts.append(y_pred.detach().float().reshape(-1).cpu())
        ops.check_errors(self.device)
        return self.evaluate_fn(torch.cat(targets).numpy(), torch.cat(predicts).numpy())

    def predict(self, model, data_loader):
        self.flush()
        model.eval()
        predicts = []
        with torch.no_grad():
            it = tqdm.tqdm(data_loader, desc="predict", smoothing=0, mininterval=1.0, disable=not self.show_progress)
            for x_dict, y in it:
                y_pred = model(self._to_device(x_dict)) if self.loss_mode else model(self._to_device(x_dict))[0]
                predicts.extend(y_pred.tolist())
        ops.check_errors(self.device)
        return predicts

    def export_onnx(self, *args, **kwargs):
        raise NotImplementedError("ONNX export is outside the HIP hot path; export with the reference "
                                  "torch_rechub.trainers.CTRTrainer after loading this model's state_dict "
                                  "(the checkpoint keys are identical).")

    def visualization(self, *args, **kwargs):
        raise NotImplementedError("model visualisation is outside the HIP hot path; use the reference trainer.")

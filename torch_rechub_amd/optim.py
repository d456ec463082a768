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
import os

import torch

from . import _lib, graphs, ops


SWEEP_WINDOW, SWEEP_FLUSH, SWEEP_LAZY_TABLES, SWEEP_DENSE_TABLES = 0, 1, 2, 3  # rh_adam_lazy_sweep modes
EAGER_HEAD = _lib.ab("eagerhead")  # False (RECHUB_AB=eagerhead=0): the one-kernel head stays a captured graph segment
ASSEMBLE_WITH_REFRESH = _lib.ab("assemble")  # False (RECHUB_AB=assemble=0): rh_batch_gather and the refresh as two launches
RELAXED_JOIN = _lib.ab("lookahead")  # False (RECHUB_AB=lookahead=0): the eager head on the sweep's queue, strict join (below)
TOUCH_GROUP = _lib.ab("touchgroup")  # False (RECHUB_AB=touchgroup=0): one touched-rows launch per gather of a step with several
MERGE_I32 = _lib.ab("mergei32")  # False (RECHUB_AB=mergei32=0): int32 index batches keep the touched pass and the dense tables' step apart
DP_MERGED_TAIL = _lib.ab("dptail")  # False (RECHUB_AB=dptail=0): head_behind's touched pass and next refresh as two launches
GATED_FORK = _lib.ab("gatedfork")  # False (RECHUB_AB=gatedfork=0): a head on the chain's queue forks its sweep at a segment boundary
DP_HEAD_BEHIND = _lib.ab("dpbehind")  # False (RECHUB_AB=dpbehind=0): the data-parallel strict head stays an eager launch in front of the graph
STEP_AHEAD = _lib.ab("ahead")  # False (RECHUB_AB=ahead=0): the head stays an eager launch in front of every replay
CHAIN_GATE = _lib.ab("chaingate")  # False (RECHUB_AB=chaingate=0): the sweep is released RH_TUNE_SWEEP_GATE_NS behind the opening (round 4)
HOST_DONE = _lib.ab("hostdone")  # False (RECHUB_AB=hostdone=0): an event record behind every deferred sweep instead of the gates' host-mapped count
WGRAD_RIDER = _lib.ab("wgradrider")  # False (RECHUB_AB=wgradrider=0): the chain's grouped weight gradients stay a launch of the backward
LATE_PACK = _lib.ab("latepack")  # False (RECHUB_AB=latepack=0): the gate is opened by a one-lane launch of its own
GATE_FALLBACK_NS = 50000  # step-ahead form with a chain-start count in the graph: release of a sweep no chain start follows (ns)
LOOK_DEPTH = 2  # step-ahead form: batches beyond the next one whose lookups in the coming sweep's window are refreshed early


class TableAdam(torch.optim.Adam):

    RING = 1024  # per-step (A, E) history for the lazy replay; lazy_k must be < RING
    _rider = None  # weight-gradient group the coming end-of-step launch carries (_ride_wgrad)
    _dp_tail_head = None  # strict head at the END of the step's graph (replicated tables under data parallelism): what _lazy_step launches

    def __init__(self, params, table_params=(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, lazy_k=0,
                 lazy_small_rows=None, lazy_dense_ratio=None, lazy_k_auto=False, **kw):
        """lazy_k <= 1: dense pass over every table row each step (rh_adam_dense).
        lazy_k  > 1: blocked-lazy EXACT mode — rows are refreshed when the batch touches them and at least every
        lazy_k steps (rh_adam_lazy_*); bit-identical to the dense pass after ``flush()``.  Valid only while the table
        gradients come from embedding lookups (fused gather / sequence pooling), not from dense terms such as an
        embedding L2 regulariser; ``flush()`` must run before anything else reads the tables."""
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
            if "foreach" not in kw and "fused" not in kw:
                kw["fused"] = True  # one multi-tensor kernel for the ~15 small dense tensors
        try:
            super().__init__(groups, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kw)
        except RuntimeError:
            if not kw.pop("fused", False):
                raise
            kw["foreach"] = True
            super().__init__(groups, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kw)
        self._tables = tables
        self._others = others
        self._bucket = None  # distributed.DenseGradBucket: when attached, dense params step through rh_adam_small
        self._small_done = False  # this step's rh_adam_small was folded into the packing launch (small_adam_args)
        self._t_hyper_host = None
        self.lazy_k = int(lazy_k) if tables else 0
        # lazy_k_auto: at the first lazy step (every row still current) a step of more than 8192 samples takes min(lazy_k, 64):
        # the pre-gather refresh replays ~ lazy_k / 2 steps per looked-up row, the window sweep's work does not depend on it
        self._lazy_k_auto = bool(lazy_k_auto)
        if self.lazy_k >= self.RING:
            raise ValueError(f"lazy_k must be < {self.RING}")
        # tables of <= lazy_small_rows rows are stepped densely (K = 1).  Left at its default (4096) the optimizer also
        # moves tables to dense stepping by their lookup volume at the first step (lazy_dense_ratio, _decide_dense_by_volume);
        # an explicit lazy_small_rows is taken as the whole placement rule unless lazy_dense_ratio is given too
        if lazy_dense_ratio is None:
            lazy_dense_ratio = 4.0 if lazy_small_rows is None else 0.0
        self.lazy_small_rows = 4096 if lazy_small_rows is None else int(lazy_small_rows)
        self.lazy_dense_ratio = float(lazy_dense_ratio)
        self._dense_by_volume = set()  # ids of tables switched to dense stepping by their lookup volume
        self._k_decided = False
        # step number (device counter _t_step) every row was last known to be at: flush() compares it with the counter
        # ITSELF, not with a host-side flag -- hipGraph replays advance the tables without running any host code
        self._flushed_at = 0
        self._prepared = False  # the bias corrections of the coming step were already computed by rh_step_scalars
        if tables or (others and others[0].is_cuda):
            dev = (tables or others)[0].device
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
            self._t_ring = torch.zeros(2 * self.RING, dtype=torch.float32, device=dev)
            if self.lazy_k > 1:
                self._t_last = [torch.zeros(p.shape[0], dtype=torch.int32, device=dev) for p in tables]
                self._lazy_groups = None
                self._ft_cache = {}
                self._touch_log = []
                self._table_ids = {id(p) for p in tables}
                # Deferred sweep (default; RECHUB_STEP_FORM=inline sweeps in line): the window sweep of step s (lazy tables)
                # is launched on a side stream at the start of step s + 1, right after that step's rows were refreshed,
                # with its step number BY VALUE; it then runs under the whole of step s + 1 (forward, backward, exchange,
                # optimizer: none of them touches a row that is behind step s) and is joined before step s + 2
                # refreshes its rows.  The launch is eager, so inside a hipGraph capture it needs a
                # graphs.SegmentedGraph (two segments per step; plain captures sweep in line).  Exact (bit-identical
                # after flush(): tests/test_gpu_properties.py, tests/test_gpu_models.py::test_graph_mode_flush_*).
                # Round 2 measured no gain (the VALU-saturating sweep starved the step's small dependent kernels: they
                # ran 2.6x slower under it).  What makes it pay (round 3): the sweep's residency capped at
                # two workgroups per CU (RH_TUNE_DEFERRED_GRID), the chain's kernels at wave priority 3
                # (RH_CHAIN_PRIO), the step's scalar / packing fusions kept, and only two graph segments:
                # DeepFM 0.365 -> 0.310 ms, DSSM 1.43 -> 1.22 ms per step.
                # Forms measured in round 3 and removed in round 4 (DESIGN 4.3 keeps the numbers): "pipelined" (one segment
                # per step, the refresh of the next batch beside the sweep: 0.37 ms), "branch" (the sweep as a captured
                # branch of the step's graph: 0.312 ms), external-event nodes (1.66 ms), CU-masked streams (0.350 ms), the
                # row-list table gradient (backward 159 vs 103 us at B = 65536).
                self.overlap_sweep = os.environ.get("RECHUB_STEP_FORM", "deferred") != "inline"
                self.head_on_side = _lib.ab("headside")
                self._head_event = None
                self._sweep_events = None   # relaxed join: the ends of the sweeps launched by the last two heads
                self._look_token = None     # relaxed join: (graph, loader generation, step) the last head looked ahead for
                self._step_ahead = None     # step-ahead form: what _merged_step launches while the graph `seg` is captured
                self._rider = None          # ... and the weight-gradient group that launch carries (_ride_wgrad)
                # step-ahead form: the sweep's release is a device word the LAST launch of the step's graph counts up
                # (rh_adam_sweep_gate_open): [openings, wall clock of the last one]
                self._gate = torch.zeros(16, dtype=torch.int64, device=dev)  # RH_GATE_WORDS (include/rechub_hip.h)
                self._gate_seen = 0  # openings issued so far (one per replay of a step-ahead graph)
                self._gate_by_pack = False
                self.gate_by_chain = False  # the captured step-ahead graph holds a chain-start count (rh_linear_fwd_gate)
                self._pre_refreshed = None  # the record rh_adam_lazy_refresh_assemble refreshed for the coming gather
                self._head_forks = False    # capture: the eager head function forks the sweep, on_gather must not cut
                self._step_recs, self._last_recs = [], []
                self.foreign_rows = False     # set by the trainers: the touched pass sees rows other ranks looked up
                self._sweep_pending = False   # sweep of the last completed step not launched yet
                self._sweep_inflight = False  # ... launched on the side stream, not joined yet
                self._side = None
                self._host_step = 0           # completed steps (host mirror of _t_step)
                self._gathers = 0             # training-mode gathers seen since the last step
                self._gathers_per_step = None  # learned from the previous step: the sweep forks at the LAST gather
                self._join_seg = None
                ops.add_lazy_listener(self)  # on_gather: refresh rows before they are read; on_touch: log lookups

    # ------------------------------------------------------------------------------------
    def _table_group(self):
        for g in self.param_groups:
            if g.get("rh_tables"):
                return g
        return None

    def attach_bucket(self, bucket):
        """Step the dense parameters with ONE rh_adam_small launch reading the packed gradient bucket."""
        if [id(p) for p in bucket.params] != [id(p) for p in self._others if p.requires_grad]:
            raise ValueError("TableAdam.attach_bucket: the bucket must hold exactly the non-table parameters, in order")
        if not bucket.params or len(bucket.params) > 128:
            return
        self._bucket = bucket
        dev = bucket.flat.device
        self._s_m = [torch.zeros_like(p) for p in bucket.params]
        self._s_v = [torch.zeros_like(p) for p in bucket.params]
        rows = ([p.data_ptr() for p in bucket.params] + [m.data_ptr() for m in self._s_m] +
                [v.data_ptr() for v in self._s_v] + [p.numel() for p in bucket.params] + list(bucket.offsets[:-1]))
        self._s_desc = torch.tensor(rows, dtype=torch.int64).to(dev)
        self._s_numel = (ctypes.c_int64 * len(bucket.params))(*[p.numel() for p in bucket.params])
        for p, m, v in zip(bucket.params, self._s_m, self._s_v):
            old = self.state.get(p)
            if old and "exp_avg" in old:  # load_state_dict() ran before the first step (resume): keep the moments
                m.copy_(old["exp_avg"])
                v.copy_(old["exp_avg_sq"])
                if not self._tables:  # with tables the step counter was restored from them
                    self._t_step.fill_(int(float(old["step"])))
                    self._t_hyper_host = None
            self.state[p] = {"step": torch.tensor(0.0), "exp_avg": m, "exp_avg_sq": v}

    def _groups_agree(self):
        keys = ("lr", "betas", "eps", "weight_decay")
        first = self.param_groups[0]
        return all(all(g[k] == first[k] for k in keys) for g in self.param_groups[1:])

    def sync_hyper(self):
        """Upload lr/betas/eps/weight_decay of the table group if they changed (call outside graph capture)."""
        g = self._table_group() or (self.param_groups[0] if self._bucket is not None else None)
        if g is None:
            return
        lr = g["lr"]
        lr = float(lr.item()) if torch.is_tensor(lr) else float(lr)
        host = (lr, float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))
        if host != self._t_hyper_host:
            if self._t_hyper_host is not None and host[1:] != self._t_hyper_host[1:]:
                self.flush()  # betas / eps / weight_decay are assumed constant inside a replay window
            self._t_hyper[:5].copy_(torch.tensor(host, dtype=torch.float64))
            self._t_hyper_host = host

    MAX_TENSORS = 128  # kMaxTensors of csrc/optim.hip: tensors per multi-tensor launch

    def _desc(self):
        """[(device descriptor, n tensors, host numel array)] -- one entry per launch of <= MAX_TENSORS tables."""
        grads = [ops.grad_buffer(p) for p in self._tables]
        key = tuple([p.data_ptr() for p in self._tables] + [g.data_ptr() for g in grads])
        if key != self._t_desc_key:
            out = []
            for c0 in range(0, len(self._tables), self.MAX_TENSORS):
                sl = slice(c0, c0 + self.MAX_TENSORS)
                tabs = self._tables[sl]
                rows = ([p.data_ptr() for p in tabs] + [g.data_ptr() for g in grads[sl]] +
                        [m.data_ptr() for m in self._t_m[sl]] + [v.data_ptr() for v in self._t_v[sl]] +
                        [p.numel() for p in tabs])
                out.append((torch.tensor(rows, dtype=torch.int64).to(tabs[0].device), len(tabs),
                            (ctypes.c_int64 * len(tabs))(*[p.numel() for p in tabs])))
            self._t_desc = out
            self._t_desc_key = key
        return self._t_desc

    # -- blocked-lazy exact mode ---------------------------------------------------------
    def _lazy_setup(self):
        """Group the tables by embed_dim (one launch each) and build their device descriptors."""
        grads = [ops.grad_buffer(p) for p in self._tables]
        key = tuple([p.data_ptr() for p in self._tables] + [g.data_ptr() for g in grads])
        if self._lazy_groups is not None and self._lazy_key == key:
            return self._lazy_groups
        by_dim = {}
        for i, p in enumerate(self._tables):
            by_dim.setdefault(int(p.shape[1]), []).append(i)
        chunks = [(D, m[c0:c0 + self.MAX_TENSORS]) for D, m in by_dim.items() for c0 in range(0, len(m), self.MAX_TENSORS)]
        out = []
        for D, members in chunks:
            rows = [int(self._tables[i].shape[0]) for i in members]
            ks = [self.table_k(self._tables[i]) for i in members]
            win = [-(-r // k) for r, k in zip(rows, ks)]
            desc = ([self._tables[i].data_ptr() for i in members] + [grads[i].data_ptr() for i in members] +
                    [self._t_m[i].data_ptr() for i in members] + [self._t_v[i].data_ptr() for i in members] +
                    [self._t_last[i].data_ptr() for i in members] + rows + ks + win)
            out.append(dict(D=D, members=members, local={id(self._tables[i]): j for j, i in enumerate(members)},
                          ldesc=torch.tensor(desc, dtype=torch.int64).to(self._tables[0].device),
                          h_rows=(ctypes.c_int64 * len(rows))(*rows), h_win=(ctypes.c_int64 * len(rows))(*win)))
        self._lazy_groups, self._lazy_key = out, key
        self._ft_cache = {}
        return out

    def _field_table(self, rec, grp):
        ids = (id(grp),) + tuple(id(w) for w in rec["weights"]) + tuple(rec["pads"])
        ft = self._ft_cache.get(ids)
        if ft is None:
            tab = [grp["local"].get(id(w), -1) for w in rec["weights"]]
            pads = [(-1 if q is None else int(q)) for q in rec["pads"]]
            ft = torch.tensor(tab + pads, dtype=torch.int64).to(self._tables[0].device)
            self._ft_cache[ids] = ft
        return ft

    def _touch(self, rec, groups, stream, refresh=False):
        for grp in groups:
            if grp["D"] != rec["D"] or not any(id(w) in grp["local"] for w in rec["weights"]):
                continue
            _lib.call("rh_adam_lazy_touched", ops._p(grp["ldesc"]), len(grp["members"]), ops._p(self._field_table(rec, grp)),
                      ops._p(rec["idesc"]), rec["idx_is_i64"], rec["B"], rec["F"], rec["D"], ops._p(self._t_hyper),
                      ops._p(self._t_ring), self.RING, 64, int(refresh),
                      ops._p(ops.err_flag(self._tables[0].device)), stream)

    def _touch_many(self, recs, groups, stream, refresh=False):
        """``_touch`` for several gathers of one step: ONE launch (rh_adam_lazy_touched_group) when they all belong to the same
        table group, one launch each otherwise."""
        recs = list(recs)
        if TOUCH_GROUP and 2 <= len(recs) <= 4 and len(groups) >= 1:
            owner = []
            for rec in recs:
                mine = [g for g in groups if g["D"] == rec["D"] and any(id(w) in g["local"] for w in rec["weights"])]
                owner.append(mine[0] if len(mine) == 1 else None)
            grp = owner[0]
            if grp is not None and all(o is grp for o in owner):
                n = len(recs)
                fts = [self._field_table(rec, grp) for rec in recs]
                ft = (ctypes.c_void_p * n)(*[t.data_ptr() for t in fts])
                idesc = (ctypes.c_void_p * n)(*[rec["idesc"].data_ptr() for rec in recs])
                i64 = (ctypes.c_int * n)(*[int(bool(rec["idx_is_i64"])) for rec in recs])
                Bs = (ctypes.c_int * n)(*[int(rec["B"]) for rec in recs])
                Fs = (ctypes.c_int * n)(*[int(rec["F"]) for rec in recs])
                cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
                _lib.call("rh_adam_lazy_touched_group", ops._p(grp["ldesc"]), len(grp["members"]), n, cast(ft), cast(idesc),
                          cast(i64), cast(Bs), cast(Fs), grp["D"], ops._p(self._t_hyper), ops._p(self._t_ring), self.RING, 64,
                          int(refresh), ops._p(ops.err_flag(self._tables[0].device)), stream)
                del fts
                return
        for rec in recs:
            self._touch(rec, groups, stream, refresh=refresh)

    def on_gather(self, rec):
        """Pre-gather event: replay the rows of this index batch up to the last completed step (their gradient rows are
        zero at this point, so this is the pure wd*p replay); the forward then reads exactly what a dense optimizer
        would have left in the table.  Unconditional, so that a captured hipGraph always contains the launch."""
        if not any(id(w) in self._table_ids for w in rec["weights"]):
            return
        seg = graphs.active()
        capturing = torch.cuda.is_current_stream_capturing()
        training = rec.get("training", torch.is_grad_enabled())
        if training:
            # "keep": the index tensors behind idesc stay alive as long as the record does (refresh-ahead replays it)
            self._step_recs.append({k: rec.get(k) for k in ("weights", "pads", "idesc", "idx_is_i64", "B", "F", "D", "keep")})
        if capturing and seg is not None and self.overlap_sweep:
            # segmented replay: EVERY replay joins the sweep forked by the previous one before its first segment (the
            # refresh below must not meet a row the sweep is still writing), whatever the state at capture time was
            if self._join_seg is not seg:
                seg.at_start(self._head_begin if self.head_on_side else self._join_sweep)
                self._join_seg = seg
            self._sweep_inflight = False
            if training and self._refresh_ahead(rec, seg):
                return
            if training and getattr(self, "_ahead", 0) > 0:
                # this step's first gather already refreshed the recorded gathers and FORKED the sweep, and this gather is
                # not the one its record announced (other index tensor, order or count): its ordinary refresh below would
                # run while the sweep -- which claims no rows -- may be writing the same rows.  Every replay joins the
                # sweep here first (it is not forked a second time in this step); the refresh-ahead records are dropped,
                # so the next capture of this model starts from the fork-after-the-last-refresh form.
                seg.cut(self._join_forked_sweep)
                self._ahead_broken = True
        elif self._sweep_inflight:  # the previous step's sweep must be done before rows are refreshed again
            if capturing:
                raise RuntimeError("TableAdam: a deferred table sweep is in flight on the side stream; call "
                                   "optimizer.flush() before capturing a training step into a plain hipGraph "
                                   "(or capture with torch_rechub_amd.graphs.SegmentedGraph)")
            else:
                self._join_sweep()
        pre, self._pre_refreshed = self._pre_refreshed, None
        if not (pre is not None and training and self._same_gather(rec, pre)):
            # (pre: rh_adam_lazy_refresh_assemble already refreshed exactly these lookups when the batch was assembled)
            if self._head_forks and capturing and seg is not None:
                # ... but this is not the gather it announced, and the eager head forks the sweep in front of the graph:
                # every replay joins that sweep before this refresh runs (as a refresh-ahead mismatch does)
                seg.cut(self._join_forked_sweep)
            self._touch(rec, self._lazy_setup(), ops._stream(), refresh=True)
        if rec.get("training", torch.is_grad_enabled()):
            self._gathers += 1
            if getattr(self, "_ahead_broken", False) and capturing:
                return  # the sweep of this step ran (and was joined) already
            if self._head_forks and capturing:
                self._head_forks = False
                # the eager head of every replay launches this step's sweep (assemble_with_refresh): it IS in flight from here
                # on in every replay -- what _join_before_foreign_rows (replicated tables under data parallelism) must see
                self._sweep_inflight = True
                return
            if self._sweep_pending and self._gathers >= (self._gathers_per_step or 1):
                if not capturing:
                    self._fork_sweep()
                elif seg is not None:
                    self._cut_fork(seg)  # every replay: eager side-stream launch after the refresh above
                    self._sweep_pending, self._sweep_inflight = False, True
            elif capturing and seg is not None and self.overlap_sweep and self._gathers >= (self._gathers_per_step or 1):
                # captured from a settled state (nothing pending at capture time): the replays still fork one sweep each
                self._cut_fork(seg)
                self._sweep_inflight = True
                # plain capture: leave it pending, step_tables() sweeps in line (device-side step number)

    @staticmethod
    def _same_gather(a, b):
        return a["idesc"] is b["idesc"] and (a["B"], a["F"], a["D"], a["idx_is_i64"]) == (b["B"], b["F"], b["D"], b["idx_is_i64"]) \
            and len(a["weights"]) == len(b["weights"]) and all(x is y for x, y in zip(a["weights"], b["weights"])) \
            and list(a["pads"]) == list(b["pads"])

    def assemble_with_refresh(self, loader, B=None, strict=False):
        """Called by the trainers INSTEAD of the loader's batch assembly when the coming step is known to gather ONE index
        batch that lives in the loader's static buffer (the previous step's record): the assembly and the pre-gather refresh
        of that batch run as ONE launch (rh_adam_lazy_refresh_assemble), the refresh reading its indices from the dataset.
        Returns False (nothing launched: the caller assembles the ordinary way) whenever that is not the situation.
        ``strict``: only the strict eager head (round 6, replicated tables under data parallelism: the sweep is joined in front
        of the touched pass over the gathered rows in every step, so neither the relaxed join's preview nor the step-ahead
        launch -- both protect the LOCAL batch's rows only -- has anything to win or the right to run)."""
        if self.lazy_k <= 1 or not self._tables or not ASSEMBLE_WITH_REFRESH or not self._k_decided:
            return False
        if ops.chain_gate is self._gate:
            ops.chain_gate = None  # (a capture that was abandoned between its head and its last launch)
        self._drop_rider()
        self._dp_tail_head = None
        recs = self._last_recs
        if len(recs) != 1 or (self._gathers_per_step or 0) != 1 or self._gathers != 0:
            return False
        args = loader.assembly_args(B)
        rec = recs[0]
        if args is None or not rec["idx_is_i64"] or rec["B"] != args["B"]:
            return False
        sp = args["sparse_out"]
        lo, hi = sp.data_ptr(), sp.data_ptr() + 8 * sp.shape[1]
        cols = rec.get("keep")
        if not cols or any((not torch.is_tensor(c)) or c.dim() != 1 or c.stride(0) != sp.shape[1] or
                           not lo <= c.data_ptr() < hi for c in cols):
            return False  # the gather's index columns are not columns of this loader's batch buffer
        groups = [g for g in self._lazy_setup() if g["D"] == rec["D"] and any(id(w) in g["local"] for w in rec["weights"])]
        if len(groups) != 1:
            return False
        grp = groups[0]
        capturing = torch.cuda.is_current_stream_capturing()
        seg = graphs.active()
        if capturing and seg is None and self.overlap_sweep:
            return False  # (a plain capture with a pending deferred sweep: on_gather raises with the explanation)
        a = args
        ft = self._field_table(rec, grp)
        keep = (grp["ldesc"], ft, rec["idesc"], a)  # (the tensors behind the pointers below)
        cargs = (ops._p(grp["ldesc"]), len(grp["members"]), ops._p(ft), ops._p(rec["idesc"]), rec["B"], rec["F"], rec["D"],
                 ops._p(self._t_hyper), ops._p(self._t_ring), self.RING, 64, ops._p(ops.err_flag(self._tables[0].device)),
                 ops._p(a["perm"]), ops._p(a["pos"]), a["N"], ops._p(a["sparse"]), a["F"], ops._p(a["dense"]), a["ND"],
                 ops._p(a["label"]), ops._p(a["sparse_out"]), ops._p(a["dense_out"]), ops._p(a["label_out"]))
        if capturing and seg is not None and self.overlap_sweep and self.head_on_side and EAGER_HEAD and \
                len(seg.segments) == 1 and self._join_seg is not seg:
            # The head of the step is now ONE kernel: it is not captured at all.  Every replay launches it eagerly on the
            # sweep's queue -- [wait for the previous chain] head -> event -> sweep -- in front of the ONE graph that holds
            # the chain (which waits for the event).  One graph launch per step less than the two-segment form.
            def head(cargs=cargs, keep=keep):
                side = self._side_stream()
                side.wait_stream(torch.cuda.current_stream())
                if self._head_event is None:
                    self._head_event = torch.cuda.Event()
                with torch.cuda.stream(side):
                    _lib.call("rh_adam_lazy_refresh_assemble", *cargs, 0, ops._stream())
                    self._head_event.record()
                    if self._host_step > 0:
                        self._sweep(SWEEP_LAZY_TABLES, ops._stream(), t_value=self._host_step)
                torch.cuda.current_stream().wait_event(self._head_event)
                self._sweep_pending, self._sweep_inflight = False, True

            def head_relaxed(cargs=cargs, keep=keep, seg=seg, loader=loader):
                # Relaxed join (round 4).  The strict form above has two cross-queue waits on the step's critical cycle
                # (chain -> head on the sweep's queue -> chain: 12 + 20 us of the 273 us period in profiles/r04_step_timeline.txt)
                # because the head must follow the previous sweep AND the previous chain.  Here the head stays on the chain's
                # queue and also refreshes the lookups of the NEXT batch that fall into the window of the sweep launched behind
                # it (lookahead = 1): that sweep meets no row the next batch reads, so the next head does not wait for it --
                # only for the sweep before it, which has had a whole step.  The sweep leaves the critical cycle altogether.
                # The preview holds when the next head belongs to the same captured step, the loader's perm / pos moved only
                # by this step's own advance and exactly one step was completed in between; anything else joins everything.
                main = torch.cuda.current_stream()
                side = self._side_stream()
                if self._sweep_events is None:
                    self._sweep_events = [torch.cuda.Event(), torch.cuda.Event()]
                    self._head_event = self._head_event or torch.cuda.Event()
                ev = self._sweep_events[self._host_step & 1]  # last recorded two heads ago
                if self._look_token == (id(seg), loader.generation, self._host_step):
                    main.wait_event(ev)
                else:
                    main.wait_stream(side)
                _lib.call("rh_adam_lazy_refresh_assemble", *cargs, 1, ops._stream())
                self._head_event.record()
                with torch.cuda.stream(side):
                    side.wait_event(self._head_event)
                    if self._host_step > 0:
                        _lib.call("rh_adam_sweep_stagger", ops._stream())  # not in the same microsecond as the chain's GEMM
                        self._sweep(SWEEP_LAZY_TABLES, ops._stream(), t_value=self._host_step)
                    ev.record()
                self._look_token = (id(seg), loader.generation, self._host_step + 1)
                self._sweep_pending, self._sweep_inflight = False, True

            def head_ahead(cargs=cargs, keep=keep, seg=seg, loader=loader):
                # Step ahead (round 4): the previous replay's LAST launch (rh_adam_lazy_step_ahead) has already assembled this
                # batch and refreshed its rows, and looked LOOK_DEPTH batches further ahead for the window of the sweep launched
                # behind it -- that sweep may run under this replay and the next; only the one launched LOOK_DEPTH + 1 tails ago
                # has to be done.  Nothing is launched here then.  Whenever that is not the situation (first replay, another
                # graph, the loader reshuffled or moved, a step in between): join everything and prepare the batch the ordinary way.
                main = torch.cuda.current_stream()
                if self._sweep_events is None or len(self._sweep_events) != LOOK_DEPTH + 1:
                    self._sweep_events = [torch.cuda.Event() for _ in range(LOOK_DEPTH + 1)]
                    self._head_event = self._head_event or torch.cuda.Event()
                if self._look_token == (id(seg), loader.generation, self._host_step):
                    done = self._done_word()
                    if done is not None:
                        # round 6: the sweeps' progress is a word of host-mapped memory the gate launches write (the gate of
                        # sweep s starts when sweep s - 1 has finished and says so) -- no event record between two kernels of
                        # the sweep's stream (7.5 us of idle queue per step on the step's longer path), no wait packet in front
                        # of the graph.  The host simply does not enqueue this replay before the sweep launched LOOK_DEPTH + 1
                        # tails ago has finished: it stays at most ~2 steps ahead of the device (it needs ~60 us per step).
                        self._wait_sweeps_done(done[0], self._host_step - LOOK_DEPTH, main)
                    else:
                        ev = self._sweep_events[(self._host_step + 1) % (LOOK_DEPTH + 1)]
                        if not ev.query():  # (every packet between two graphs costs the chain ~7 us of idle queue)
                            main.wait_event(ev)
                else:
                    main.wait_stream(self._side_stream())
                    _lib.call("rh_adam_lazy_refresh_assemble", *cargs, 0, ops._stream())
                self._gate_seen += 1  # the last launch of this replay's graph opens the gate of this step's sweep
                self._sweep_pending, self._sweep_inflight = False, True

            def tail_ahead(seg=seg, loader=loader):
                # after the step's graph (its last launch prepared the next batch): the sweep of the step just completed, by
                # value, on the side stream behind an event; it is not joined before LOOK_DEPTH + 1 further replays
                self._host_step += 1
                h = self._host_step
                side = self._side_stream()
                with torch.cuda.stream(side):
                    # released by the graph's own last launch (no event record between two graph launches of the chain) AND the
                    # next replay's chain start (its first own GEMM has placed its workgroups: rh_linear_fwd_gate) -- into that
                    # GEMM, not beside a launch of the chain; without a chain start GATE_FALLBACK_NS behind the opening (round 4:
                    # a wall-clock hold-back of RH_TUNE_SWEEP_GATE_NS, which remains the release for graphs without an own GEMM)
                    done = self._done_word()
                    if done is not None:  # (this gate starts when everything before it on this stream -- sweep h - 1 -- is done)
                        _lib.call("rh_adam_sweep_gate_done", ops._p(self._gate), self._gate_seen,
                                  GATE_FALLBACK_NS if self.gate_by_chain else 0,
                                  ops._p(ops.err_flag(self._tables[0].device)), done[1], h - 1, ops._stream())
                        self._sweep(SWEEP_LAZY_TABLES, ops._stream(), t_value=h)
                    else:
                        _lib.call("rh_adam_sweep_gate", ops._p(self._gate), self._gate_seen,
                                  GATE_FALLBACK_NS if self.gate_by_chain else 0,
                                  ops._p(ops.err_flag(self._tables[0].device)), ops._stream())
                        self._sweep(SWEEP_LAZY_TABLES, ops._stream(), t_value=h)
                        self._sweep_events[h % (LOOK_DEPTH + 1)].record()
                self._look_token = (id(seg), loader.generation, h)
                self._sweep_pending, self._sweep_inflight = False, True

            def head_behind(cargs=cargs, keep=keep, seg=seg, loader=loader):
                # Strict head whose kernel is the LAST launch of the previous replay's graph (_lazy_step captured it behind the
                # touched pass): this batch is assembled and its rows are refreshed already, on the chain's own queue -- no
                # cross-queue hop in front of the chain, the sweep goes to its queue at once.  Replicated tables under data
                # parallelism only: there every step joins its sweep in front of the touched pass (foreign rows), so the
                # refresh behind that pass follows the sweep AND the chain by stream order, which is all the strict head asks.
                # Whenever the previous replay was not this graph's, the loader moved or a step went in between: prepare here.
                main = torch.cuda.current_stream()
                if self._look_token != (id(seg), loader.generation, self._host_step):
                    main.wait_stream(self._side_stream())
                    _lib.call("rh_adam_lazy_refresh_assemble", *cargs, 0, ops._stream())
                self._gate_seen += 1  # the last launch of this replay's graph opens the gate of this step's sweep
                self._sweep_pending, self._sweep_inflight = False, True

            def tail_behind(seg=seg, loader=loader):
                # after the step's graph: the sweep of the step just completed goes to its queue behind a gate -- opened by the
                # graph's last launch and released by the NEXT replay's chain start (its first own GEMM has placed its
                # workgroups), as in the step-ahead form: dispatched a few microseconds into the chain instead, the sweep's
                # workgroups spread unevenly and the statistics-prologue GEMM beside them took 93 us for 27 (traced)
                self._host_step += 1
                h = self._host_step
                with torch.cuda.stream(self._side_stream()):
                    _lib.call("rh_adam_sweep_gate", ops._p(self._gate), self._gate_seen,
                              GATE_FALLBACK_NS if self.gate_by_chain else 0, ops._p(ops.err_flag(self._tables[0].device)),
                              ops._stream())
                    self._sweep(SWEEP_LAZY_TABLES, ops._stream(), t_value=h)
                self._look_token = (id(seg), loader.generation, h)
                self._sweep_pending, self._sweep_inflight = False, True

            if strict and DP_HEAD_BEHIND and self.foreign_rows:
                seg.at_start(head_behind)
                seg.after(tail_behind)
                self._advance_seg = seg  # (step_tables: tail_behind counts the replayed steps)
                self._dp_tail_head = dict(seg=seg, cargs=cargs, keep=keep, rec=rec, grp=grp, ft=ft, a=a)
                ops.chain_gate = self._gate if CHAIN_GATE else None  # (the first own GEMM captured into this graph counts the chain start)
                del ops.chain_gate_used[:]
            elif strict:
                seg.at_start(head)
            elif RELAXED_JOIN and STEP_AHEAD and self._merge_ahead_ok(rec, grp):
                seg.at_start(head_ahead)
                seg.after(tail_ahead)
                self._advance_seg = seg  # (step_tables: tail_ahead counts the replayed steps)
                self._step_ahead = dict(seg=seg, rec=rec, grp=grp, ft=ft, a=a)
                # the first own GEMM captured into this graph counts the chain start that releases the sweep (round 5: a
                # dependency instead of round 4's wall-clock hold-back behind the opening; ops._MlpChainFn, csrc/gemm.hip)
                ops.chain_gate = self._gate if CHAIN_GATE else None
                del ops.chain_gate_used[:]
                # ... and the chain's grouped weight gradients ride in this graph's last table launch (round 6)
                ops.wgrad_rider = self._ride_wgrad if WGRAD_RIDER else None
                ops.wgrad_rider_flush = self._flush_rider if WGRAD_RIDER else None
            else:
                seg.at_start(head_relaxed if RELAXED_JOIN else head)
            self._join_seg = seg
            self._head_forks = True  # on_gather: the sweep of this step is forked by head(), no cut
            self._sweep_pending, self._sweep_inflight = False, True
            self._pre_refreshed = rec
            return True
        if capturing and seg is not None and self.overlap_sweep:
            if self._join_seg is not seg:  # what on_gather does in front of the first refresh of a segmented capture
                seg.at_start(self._head_begin if self.head_on_side else self._join_sweep)
                self._join_seg = seg
            self._sweep_inflight = False
        elif self._sweep_inflight:
            self._join_sweep()
        _lib.call("rh_adam_lazy_refresh_assemble", *cargs, 0, ops._stream())
        self._pre_refreshed = rec
        return True

    def _refresh_ahead(self, rec, seg):
        """Deferred form, a step with SEVERAL gathers (two-tower / sequence models): the sweep may only start once the rows
        of every gather of the step are current, i.e. after the LAST refresh -- in the configs[4] step that is a third of
        the way in, and the sweep (0.65 ms there) then outlasts the chain.  The index buffers of all the step's gathers are
        static and assembled before the first one, so the first gather refreshes the rows of ALL of them (the records of
        the previous step's gathers, checked one by one against this step's) and forks the sweep at once; the later gathers
        find their rows done.  Returns True when this gather needs no refresh of its own any more."""
        def same(a, b):
            return a["idesc"] is b["idesc"] and (a["B"], a["F"], a["D"], a["idx_is_i64"]) == (b["B"], b["F"], b["D"], b["idx_is_i64"]) \
                and len(a["weights"]) == len(b["weights"]) and all(x is y for x, y in zip(a["weights"], b["weights"])) \
                and list(a["pads"]) == list(b["pads"])
        k = self._gathers
        recs = self._last_recs
        if k == 0:
            self._ahead = 0
            self._ahead_broken = False
            if len(recs) < 2 or (self._gathers_per_step or 0) != len(recs) or not same(rec, recs[0]):
                return False
            groups = self._lazy_setup()
            self._touch_many([dict(r, training=True) for r in recs], groups, ops._stream(), refresh=True)
            self._ahead = len(recs)
            self._gathers = 1
            self._cut_fork(seg, head_only=True)  # every replay: the side-stream launch, after the refreshes above
            self._sweep_pending, self._sweep_inflight = False, True
            return True
        if k < getattr(self, "_ahead", 0) and same(rec, recs[k]):
            self._gathers += 1
            return True
        return False  # not the gather the record announced: on_gather joins the forked sweep, then refreshes it

    def _join_forked_sweep(self):
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        self._sweep_inflight = False

    def release_gate(self):
        """No further step follows the ones enqueued so far (the caller is about to synchronise, or the epoch is over): the
        last step-ahead sweep, which rh_adam_sweep_gate holds back for the NEXT step's chain start, may go at once instead of
        after GATE_FALLBACK_NS.  Harmless when nothing is held back (a chain start counted where none was needed: every
        later gate takes its base count at its own opening)."""
        if getattr(self, "gate_by_chain", False) and getattr(self, "_gate", None) is not None and \
                not torch.cuda.is_current_stream_capturing():
            _lib.call("rh_adam_sweep_release", ops._p(self._gate), ops._stream())

    def settle_sweep(self):
        """Bring the deferred-sweep state to rest (nothing in flight, nothing pending) without a full flush: what a
        switch between the forms of the captured step starts from."""
        if self.lazy_k > 1 and self._tables and self.overlap_sweep:
            self.release_gate()
            self._join_sweep()
            self._finish_sweep()

    def _sweep(self, mode, stream, t_value=-1):
        for grp in self._lazy_setup():
            _lib.call("rh_adam_lazy_sweep", ops._p(grp["ldesc"]), len(grp["members"]),
                      ctypes.cast(grp["h_rows"], ctypes.c_void_p), ctypes.cast(grp["h_win"], ctypes.c_void_p), grp["D"],
                      ops._p(self._t_hyper), ops._p(self._t_ring), self.RING, mode, t_value, stream)

    # -- the head of the step on the sweep's queue (round 4) --------------------------------------------------------------
    # Two-segment replay:  [batch assembly, refresh] | fork sweep | [forward ... touched rows].  Forked from the MAIN stream,
    # the sweep waited ~20 us for the cross-queue event behind the refresh before it started -- and the sweep's path
    # (head + that wait + ~240 us) is the longer of the step's two.  With head_on_side the FIRST segment is replayed on the
    # sweep's own stream: batch assembly -> refresh -> sweep are consecutive kernels of one queue (no event in between; the
    # previous sweep is in front of them on that queue, which IS the join), and it is the chain on the main stream that waits
    # for the event recorded behind the refresh.  The cross-queue latency moves from the longer path to the shorter one.
    # (A/B: RECHUB_AB=headside=0.)
    def _head_begin(self):
        """at_start of a segmented replay: the head segment (on the side stream) follows the previous step's chain."""
        if self._side is None:
            self._side = self._make_side_stream()
        self._side.wait_stream(torch.cuda.current_stream())
        self._sweep_inflight = False  # (whatever was in flight is in front of the head on the same queue)

    def _side_stream(self):
        if self._side is None:
            self._side = self._make_side_stream()
        return self._side

    def _cut_fork(self, seg, head_only=False):
        """Close the head segment: the sweep is forked here on every replay.  The head may run on the sweep's stream when
        everything captured so far IS the head (batch assembly + the refreshes): a step with one gather, or refresh-ahead."""
        head = self.head_on_side and len(seg.segments) == 1 and (head_only or (self._gathers_per_step or 1) == 1)
        if head:
            seg.replay_segment_on(0, self._side_stream)
            seg.cut(self._fork_sweep_behind_head)
        else:
            if self.head_on_side and self._join_seg is seg and self._head_begin in seg.before:
                seg.before[seg.before.index(self._head_begin)] = self._join_sweep  # the head stays on the main stream
            if GATED_FORK and self.gated_fork and not self.head_on_side and not self.foreign_rows and \
                    len(seg.segments) == 1 and (self._gathers_per_step or 1) == 1:
                # Head on the chain's queue, one gather per step, no join inside the step (the trainers set ``gated_fork`` for
                # row-sharded tables under data parallelism: a join at a LATER segment boundary -- foreign rows, a refresh-ahead
                # mismatch -- would run before the sweep launched behind the graph and find nothing to wait for): NO segment
                # boundary for the fork (15 us of idle queue in front of the chain, profiles/r06_dp_one_rank_shard_step.txt).
                # The graph counts an opening behind the head's refresh instead, and the sweep is launched AFTER the whole
                # graph has been enqueued, behind a gate that waits for that opening and then for the chain start -- the
                # step's first own GEMM has placed its workgroups -- like the step-ahead form's (stream_gate_kernel; a gate
                # nobody opens gives up after its timeout).  Same order on the device: refresh -> sweep -> (join in front
                # of the next replay).
                _lib.call("rh_adam_sweep_gate_open", ops._p(self._gate), ops._stream())
                ops.chain_gate = self._gate if CHAIN_GATE else None
                del ops.chain_gate_used[:]
                self._gated_fork_seg = seg
                if self._fork_sweep_gated not in seg.after_fns:
                    seg.after(self._fork_sweep_gated)  # (registered before step_tables' host-step advance: runs first)
            else:
                seg.cut(self._fork_sweep)

    _gated_fork_seg = None
    gated_fork = False  # set by the trainers (row-sharded tables under data parallelism)

    def _fork_sweep_gated(self):
        """after() of a graph captured by the gated fork above: this replay's opening is enqueued -- its sweep may wait for it."""
        self._gate_seen += 1
        if self._host_step > 0:
            with torch.cuda.stream(self._side_stream()):
                _lib.call("rh_adam_sweep_gate", ops._p(self._gate), self._gate_seen,
                          GATE_FALLBACK_NS if self.gate_by_chain else 0, ops._p(ops.err_flag(self._tables[0].device)),
                          ops._stream())
                self._sweep(SWEEP_LAZY_TABLES, ops._stream(), t_value=self._host_step)
        self._sweep_pending, self._sweep_inflight = False, True

    def _fork_sweep_behind_head(self):
        """After the head segment was enqueued on the side stream: mark the end of the refresh there (the chain waits for
        it), then the sweep of the last completed step right behind it on the same queue."""
        side = self._side_stream()
        if self._head_event is None:
            self._head_event = torch.cuda.Event()
        with torch.cuda.stream(side):
            self._head_event.record()
            if self._host_step > 0:
                self._sweep(SWEEP_LAZY_TABLES, ops._stream(), t_value=self._host_step)
        torch.cuda.current_stream().wait_event(self._head_event)
        self._sweep_pending, self._sweep_inflight = False, True

    def _fork_sweep(self):
        """Launch the sweep of the last completed step on the side stream, ordered after everything queued so far."""
        if self._side is None:
            self._side = self._make_side_stream()
        self._side.wait_stream(torch.cuda.current_stream())
        if self._host_step > 0:
            with torch.cuda.stream(self._side):
                self._sweep(SWEEP_LAZY_TABLES, ops._stream(), t_value=self._host_step)
        self._sweep_pending, self._sweep_inflight = False, True

    def _make_side_stream(self):
        """The sweep's stream (a plain stream: CU-masked and priority streams were measured in round 3 and bought nothing)."""
        return graphs.role_stream("sweep", self._tables[0].device)  # one per process: graphs.role_stream says why

    def _join_sweep(self):
        if self._sweep_inflight:
            torch.cuda.current_stream().wait_stream(self._side)
            self._sweep_inflight = False

    def _advance_host_step(self):
        """Replay-time mirror of what step_tables() does on the host in an eager step."""
        self._host_step += 1
        self._sweep_pending = self.overlap_sweep

    def on_touch(self, rec):
        if any(id(w) in self._table_ids for w in rec["weights"]):
            self._touch_log.append(rec)

    def _merge_ok(self, groups):
        if len(self._touch_log) != 1 or len(groups) != 1:
            return False
        rec, grp = self._touch_log[0], groups[0]
        # (int32 index columns -- the row-sharded step's localised indices -- take the merged launch too since round 6,
        # rh_adam_lazy_step_mode_idx; the step-ahead launches read int64 indices from the dataset and check for themselves)
        return not (grp["D"] != rec["D"] or not (rec["idx_is_i64"] or MERGE_I32) or rec["B"] < 1 or
                    not any(id(w) in grp["local"] for w in rec["weights"]))

    def gate_for_late_pack(self):
        """Called by the trainer where it would launch the dense gradients' packing + Adam: while the step-ahead graph is
        being captured, returns the gate the packing launch shall open -- the trainer then launches it BEHIND step() -- else None."""
        ah = getattr(self, "_step_ahead", None)
        if ah is None or graphs.active() is not ah["seg"] or not torch.cuda.is_current_stream_capturing() or not LATE_PACK:
            return None
        self._gate_by_pack = True
        return self._gate

    _sweep_done = None  # (pinned host int64 word, its device address) | False: no host-mapped memory (events instead)

    def _done_word(self):
        """The host-mapped word in which the sweep gates of the step-ahead form count the finished sweeps, or None."""
        if self._sweep_done is None:
            self._sweep_done = False
            if HOST_DONE:
                try:
                    host = torch.zeros(1, dtype=torch.int64).pin_memory()
                    dev = ctypes.c_void_p()
                    _lib.call("rh_host_device_pointer", ctypes.c_void_p(host.data_ptr()), ctypes.byref(dev))
                    self._sweep_done = (host, dev)
                except RuntimeError:
                    self._sweep_done = False
        return self._sweep_done or None

    @staticmethod
    def _wait_sweeps_done(host_word, need, main, timeout_s=0.05):
        """Host-side wait until the device has counted ``need`` finished sweeps (normally true on the first look)."""
        if need <= 0 or int(host_word[0]) >= need:
            return
        import time
        t_end = time.perf_counter() + timeout_s
        while int(host_word[0]) < need:
            if time.perf_counter() > t_end:  # (a stalled device: fall back to the stream dependency and go on)
                main.wait_stream(graphs.role_stream("sweep", main.device))
                return

    def _ride_wgrad(self, problems, B):
        """ops.wgrad_rider while this optimizer captures a step-ahead graph: take the MLP chain's grouped weight gradients
        into the end-of-step launch (_merged_step).  False = not this step (the backward launches them itself)."""
        ah = self._step_ahead
        if ah is None or graphs.active() is not ah["seg"] or not torch.cuda.is_current_stream_capturing() or not problems:
            return False
        have = self._rider
        if have is not None and (have[1] != int(B) or len(have[0]) + len(problems) > 8):
            return False  # (one launch carries at most eight problems of one batch size: the rest launch themselves)
        if len(problems) > 8:
            return False
        self._rider = ((have[0] if have is not None else []) + list(problems), int(B))
        return True

    def _flush_rider(self):
        """Weight gradients taken by _ride_wgrad that no end-of-step launch carried (the step ended in another form): their own
        grouped launch, now -- the packing launch that sums their slabs comes after step()."""
        rider, self._rider = self._rider, None
        if rider is not None:
            ops.linear_wgrad_partial_group(*rider)

    def _drop_rider(self):
        self._rider = None  # (of a capture that was abandoned: its tensors belong to a dead graph)
        if ops.wgrad_rider == self._ride_wgrad:
            ops.wgrad_rider = ops.wgrad_rider_flush = None

    def _merge_ahead_ok(self, rec, grp):
        """Will _merged_step of the step being captured see exactly this gather over exactly this table group?"""
        groups = self._lazy_setup()
        # lazy_k >= LOOK_DEPTH + 2: the sweeps of the last LOOK_DEPTH steps may be in flight while this step's last launch
        # claims rows of ITS window -- the windows of LOOK_DEPTH + 1 consecutive steps must be disjoint (at lazy_k = 2,
        # window(t) is window(t - 2): tests/test_lazy_protocol_model.py); smaller lazy_k takes the relaxed / strict head
        return len(groups) == 1 and groups[0] is grp and rec["idx_is_i64"] and self.overlap_sweep and \
            self.lazy_k >= LOOK_DEPTH + 2

    def _merged_step(self, groups, stream):
        """The touched-rows step of the batch and the window sweep as ONE launch (rh_adam_lazy_step) when the step has a
        single index batch over a single table group with int64 indices -- the DeepFM / DCN / WideDeep step.  The short,
        latency-bound touched pass then runs under the ALU-bound sweep instead of in front of it."""
        if not self._merge_ok(groups):
            return False
        rec, grp = self._touch_log[0], groups[0]
        ah = self._step_ahead
        if ah is not None and graphs.active() is ah["seg"] and torch.cuda.is_current_stream_capturing():
            self._step_ahead = None
            if not (self._same_gather(rec, ah["rec"]) and grp is ah["grp"] and self.overlap_sweep):
                raise RuntimeError("TableAdam: the captured step does not end with the gather its head announced "
                                   "(step-ahead form; RECHUB_AB=ahead=0 captures the eager-head form)")
            a = ah["a"]
            cargs = (ops._p(grp["ldesc"]), len(grp["members"]),
                     ctypes.cast(grp["h_rows"], ctypes.c_void_p), ctypes.cast(grp["h_win"], ctypes.c_void_p), grp["D"],
                     ops._p(self._t_hyper), ops._p(self._t_ring), self.RING, ops._p(ah["ft"]), ops._p(rec["idesc"]), rec["B"],
                     rec["F"], ops._p(ops.err_flag(self._tables[0].device)), ops._p(a["perm"]), ops._p(a["pos"]), a["N"],
                     ops._p(a["sparse"]), a["F"], ops._p(a["dense"]), a["ND"], ops._p(a["label"]), ops._p(a["sparse_out"]),
                     ops._p(a["dense_out"]), ops._p(a["label_out"]), LOOK_DEPTH)
            rider, self._rider = self._rider, None
            if ops.wgrad_rider == self._ride_wgrad:
                ops.wgrad_rider = ops.wgrad_rider_flush = None
            if rider is not None:
                # the chain's weight gradients as the first workgroups of this launch (ops._MlpChainFn.backward handed them over)
                wargs, keep = ops.wgrad_group_args(*rider)
                _lib.call("rh_adam_lazy_step_ahead_wgrad", *cargs, *wargs, stream)
                del keep
            else:
                _lib.call("rh_adam_lazy_step_ahead", *cargs, stream)
            if not self._gate_by_pack:  # (else the packing launch behind this one opens it: gate_for_late_pack)
                _lib.call("rh_adam_sweep_gate_open", ops._p(self._gate), stream)
            self._gate_by_pack = False
            # (a graph without an own GEMM in front -- no fused MLP chain -- never counts a chain start: its sweeps are
            # released by the gate's fallback, RH_TUNE_SWEEP_GATE_NS behind the opening, as in round 4)
            self.gate_by_chain = bool(ops.chain_gate_used)
            ops.chain_gate = None
            self._sweep_pending = True
            return True
        # deferred sweep: only the dense (K = 1) tables ride along here, the lazy tables' window goes to the side stream
        mode = SWEEP_DENSE_TABLES if self.overlap_sweep else SWEEP_WINDOW
        _lib.call("rh_adam_lazy_step_mode_idx", ops._p(grp["ldesc"]), len(grp["members"]),
                  ctypes.cast(grp["h_rows"], ctypes.c_void_p), ctypes.cast(grp["h_win"], ctypes.c_void_p), grp["D"],
                  ops._p(self._t_hyper), ops._p(self._t_ring), self.RING, ops._p(self._field_table(rec, grp)),
                  ops._p(rec["idesc"]), int(bool(rec["idx_is_i64"])), rec["B"], rec["F"], 64,
                  ops._p(ops.err_flag(self._tables[0].device)), mode, stream)
        if self.overlap_sweep:
            self._sweep_pending = True
        return True

    SWEEP_BOUND_ROWS = 64_000_000  # lazy rows beyond which a step is bound by the window sweep, not by its chain (lazy_k, tuner)

    # Row-sharded tables under data parallelism (the step does not self-tune there): a rank's shard sweeps rows / world rows
    # per step.  Below ~200 M lazy elements the deferred sweep no longer pays for its segment boundary and cross-queue edge --
    # the window sweep goes back in line (the merged end-of-step launch) and an automatic lazy_k drops to 64 (a shorter
    # replay per refreshed row; the short sweep's traffic does not matter).  Measured per-rank step of the DeepFM headline on a
    # one-rank group with every table at 1/2, 1/4, 1/8 of its rows (tools/r05_session{3,4}.sh, profiles/r05_dp_shard_form.txt):
    #   1/2 (270 M elements): deferred K = 128 / 64 0.326 / 0.322 ms, in line K = 128 / 64 0.332 / 0.324  -> left deferred
    #   1/4 (135 M):          deferred 0.325 / 0.315,                 in line 0.300 / 0.289
    #   1/8 ( 68 M):          deferred 0.313,                          in line 0.282 / 0.269 (K = 32: 0.267, K = 16: 0.272)
    SHORT_SWEEP_ELEMENTS = 200_000_000

    def prefer_inline_for_short_sweeps(self, auto_k):
        """Called by the trainers once, before the first step, for row-sharded tables (RECHUB_STEP_FORM pins the form
        instead).  Returns True when the in-line form was chosen."""
        if self.lazy_k <= 1 or not self._tables or self._host_step or self._sweep_pending or self._sweep_inflight:
            return False
        n = sum(int(p.numel()) for p in self._tables if self.table_k(p) != 1)
        if n > self.SHORT_SWEEP_ELEMENTS:
            return False
        self.overlap_sweep = False
        if auto_k and self.lazy_k > 64:
            self.lazy_k = 64
            self._lazy_groups = None
        return True

    def lazy_rows(self):
        """Rows of the tables that are stepped lazily (window sweep + touched passes)."""
        return sum(int(p.shape[0]) for p in self._tables if self.table_k(p) != 1)

    def table_k(self, p):
        """Window divisor of table ``p``: 1 = stepped densely (with its gradient) by every sweep, lazy_k = blocked-lazy."""
        if self.lazy_k <= 1 or int(p.shape[0]) <= self.lazy_small_rows or id(p) in self._dense_by_volume:
            return 1
        return self.lazy_k

    def _decide_dense_by_volume(self):
        """Once, at the first lazy step (every row is current then, so a table's mode may still change): a table whose
        rows are looked up so often that the touched-row passes cost more than streaming it -- rows <= lazy_dense_ratio x
        lookups per step, e.g. the item table under DIN's 100-position histories: 63 k rows, 409 600 lookups -- is
        stepped densely like the small tables.  Measured cost model: ~0.5 ns per lookup for the two touched passes
        against ~0.09 ns per row for the dense pass."""
        self._k_decided = True
        if torch.cuda.is_current_stream_capturing():
            return  # (a capture must not change the table grouping under itself)
        changed = False
        # (before the placement rule's early return: an explicit lazy_small_rows switches the volume rule off, not this one)
        if self._lazy_k_auto and self.lazy_k > 64 and self._touch_log and (
                max(int(r["B"]) for r in self._touch_log) > 8192 or self.lazy_rows() > self.SWEEP_BOUND_ROWS):
            # (second condition, round 5: with > 64 M lazy rows -- configs[4], 110 M -- the window sweep is far longer than the
            # step's chain and the pre-gather refresh sits IN FRONT of it on the same path: what counts is refresh + sweep, and the
            # refresh replays lazy_k / 2 steps per looked-up row.  DSSM: 0.831 ms at 128, 0.788 at 64 -- round 4 had made 128 the
            # global default for the DeepFM step, where the sweep runs beside a chain of its own length)
            self.lazy_k = 64
            changed = True
        if self.lazy_dense_ratio <= 0:
            if changed:
                self._lazy_groups = None
            return
        lookups = {}
        for rec in self._touch_log:
            for w in rec["weights"]:
                lookups[id(w)] = lookups.get(id(w), 0) + int(rec["B"])
        for p in self._tables:
            n = lookups.get(id(p), 0)
            if n and self.table_k(p) != 1 and int(p.shape[0]) <= self.lazy_dense_ratio * n:
                self._dense_by_volume.add(id(p))
                changed = True
        if changed:
            self._lazy_groups = None

    def _join_before_foreign_rows(self):
        """Replicated tables under data parallelism: the touched-rows step below runs over the GATHERED index matrix, i.e. also
        over the rows the OTHER ranks' batches looked up -- rows this rank's pre-gather refresh never stamped.  The deferred
        sweep of the previous step (side stream, no claims) may be replaying exactly such a row of its window at this moment:
        it read the row's last-step word when it fetched it and would store its state over the step applied here (the step's
        gradient lost).  Nothing of the kind can happen to the rows of the LOCAL batch (refreshed = stamped before the sweep
        was launched), nor with row-sharded tables (the shard's gather refreshes the rows of the whole global batch that live
        on this rank).  So with foreign rows the sweep is joined before the touched pass: eagerly, or -- in a segmented
        capture -- by one more cut (every replay waits for the side stream there).  Found in round 5 by reading the protocol
        against the data-parallel step (tests/test_lazy_protocol_model.py::test_foreign_rows_*); the world-2 GPU tests run
        eager steps on tables small enough for every sweep to be over long before."""
        if not self.foreign_rows or not self.overlap_sweep:
            return
        if torch.cuda.is_current_stream_capturing():
            seg = graphs.active()
            if seg is not None and self._sweep_inflight:
                seg.cut(self._join_forked_sweep)
        else:
            self._join_sweep()

    def _lazy_step(self, stream):
        if not self._k_decided:
            self._decide_dense_by_volume()
        self._join_before_foreign_rows()
        groups = self._lazy_setup()
        if self._dp_merged_tail(groups, stream):
            del self._touch_log[:]
            return
        if self._merged_step(groups, stream):
            del self._touch_log[:]
        else:
            # rows of the batches are at step t-1 (refreshed before the forward): one step each
            self._touch_many(self._touch_log, groups, stream)
            del self._touch_log[:]
            if self.overlap_sweep:
                self._sweep(SWEEP_DENSE_TABLES, stream)  # small tables take their gradient now; the rest is deferred
                self._sweep_pending = True
            else:
                self._sweep(SWEEP_WINDOW, stream)
        if self._gated_fork_seg is not None:
            if graphs.active() is self._gated_fork_seg:
                self.gate_by_chain = bool(ops.chain_gate_used)  # (else: the gate's wall-clock hold-back behind the opening)
                ops.chain_gate = None
            self._gated_fork_seg = None
        th, self._dp_tail_head = self._dp_tail_head, None
        if th is not None and graphs.active() is th["seg"] and torch.cuda.is_current_stream_capturing():
            # (head_behind: the NEXT batch's assembly + refresh as the last launch of this step's graph; the loader's position
            # has been advanced by this step's scalar launch)
            _lib.call("rh_adam_lazy_refresh_assemble", *th["cargs"], 0, stream)
            _lib.call("rh_adam_sweep_gate_open", ops._p(self._gate), stream)
            self.gate_by_chain = bool(ops.chain_gate_used)
            ops.chain_gate = None

    def _dp_merged_tail(self, groups, stream):
        """Replicated tables under data parallelism, head_behind form: the touched-rows step over the GATHERED index matrix, the
        dense tables' step, the next local batch's assembly and the refresh of its rows as ONE launch
        (rh_adam_lazy_step_ahead_touched, look-ahead depth 0: the sweep is joined in every step) instead of two; then the
        opening of this step's sweep gate.  False whenever the step is not that shape (the two launches follow)."""
        th = self._dp_tail_head
        if th is None or not DP_MERGED_TAIL or not self._merge_ok(groups) or graphs.active() is not th["seg"] or \
                not torch.cuda.is_current_stream_capturing() or groups[0] is not th["grp"]:
            return False
        rec, grp, a, lrec = self._touch_log[0], th["grp"], th["a"], th["rec"]
        if rec["F"] != lrec["F"] or rec["D"] != lrec["D"] or not lrec["idx_is_i64"] or not rec["idx_is_i64"]:
            return False
        self._dp_tail_head = None
        _lib.call("rh_adam_lazy_step_ahead_touched", ops._p(grp["ldesc"]), len(grp["members"]),
                  ctypes.cast(grp["h_rows"], ctypes.c_void_p), ctypes.cast(grp["h_win"], ctypes.c_void_p), grp["D"],
                  ops._p(self._t_hyper), ops._p(self._t_ring), self.RING, ops._p(th["ft"]), ops._p(lrec["idesc"]), lrec["B"],
                  lrec["F"], ops._p(ops.err_flag(self._tables[0].device)), ops._p(a["perm"]), ops._p(a["pos"]), a["N"],
                  ops._p(a["sparse"]), a["F"], ops._p(a["dense"]), a["ND"], ops._p(a["label"]), ops._p(a["sparse_out"]),
                  ops._p(a["dense_out"]), ops._p(a["label_out"]), 0, ops._p(rec["idesc"]), rec["B"], stream)
        _lib.call("rh_adam_sweep_gate_open", ops._p(self._gate), stream)
        self.gate_by_chain = bool(ops.chain_gate_used)
        ops.chain_gate = None
        self._gated_fork_seg = None
        self._sweep_pending = True
        return True

    def _finish_sweep(self):
        """A sweep that was not forked (no training-mode gather since the last step, or a plain hipGraph capture)
        runs in line, before the step counter moves."""
        if self._sweep_pending:
            # the step BY VALUE wherever the host knows it: with the step's scalar launch fused into the forward
            # (fuse_prepare) the device-side step number may already belong to the coming step when this runs.  Only a
            # plain hipGraph capture (no host code per replay) takes the device-side number -- can_fuse_prepare() keeps the
            # early advance out of that case.
            plain_capture = torch.cuda.is_current_stream_capturing() and graphs.active() is None
            self._sweep(SWEEP_LAZY_TABLES, ops._stream(), t_value=-1 if plain_capture else self._host_step)
            self._sweep_pending = False

    def flush(self):
        """Bring every table row up to the current step (no-op in dense mode).  Must run before the weights are read
        by anything but the training step: evaluation, state_dict, checkpointing.

        Whether rows are behind is decided from the DEVICE step counter (one scalar read-back, i.e. a stream sync --
        every caller is an epoch / evaluation / checkpoint boundary): steps replayed from a hipGraph run no host code,
        so a host-side dirty flag would miss them (round-1 bug: stale rows in state_dict() after graph replays)."""
        if self.lazy_k > 1 and self._tables:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("TableAdam.flush() inside a hipGraph capture")
            self.rollback_abandoned_prepare()  # a forward that never reached its optimizer step advanced the counter
            self.release_gate()
            self._join_sweep()
            self._sweep_pending = False  # subsumed: the flush visits every row
            t = int(self._t_step.item())
            # plain hipGraph replays run no host code: the host mirror of the step (by-value argument of deferred sweeps)
            # is re-read from the device counter wherever the host synchronises anyway
            self._host_step = t
            if t != self._flushed_at:
                self._sweep(SWEEP_FLUSH, ops._stream())
                self._flushed_at = t

    # -- the step's scalar launch (ops.StepFusion): rh_step_scalars may do this optimizer's rh_adam_prepare --------
    def can_fuse_prepare(self):
        # (with the deferred sweep too: it takes its step number by value and its (A, E) from the ring entry of that
        # step, so the early advance of hyper[12..14] by this step's scalar launch does not reach it)
        if getattr(self, "overlap_sweep", False) and torch.cuda.is_current_stream_capturing() and graphs.active() is None:
            return False  # plain capture: a pending deferred sweep is finished in line with the DEVICE-side step number
        return (self._tables or self._bucket is not None) and not self._prepared

    def fuse_prepare(self):
        """Arguments of the prepare part of rh_step_scalars; the next step_tables() then skips rh_adam_prepare.  Must
        be launched AFTER the last gather of the step's forward (the pre-gather refresh reads the previous step's
        corrections) and before the optimizer kernels -- i.e. where the loss is computed."""
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()  # lr / betas / eps / weight_decay must be on the device BEFORE the corrections are formed
        self._prepared = True
        return self._t_hyper, self._t_step, self._t_ring, self.RING

    def rollback_abandoned_prepare(self):
        """Called by the trainers at the start of a step.  ``fuse_prepare()`` advances the device step counter and the
        bias corrections during the FORWARD (they ride in the step's scalar launch); if that step was then abandoned
        (an exception in the backward, a caller that never stepped), ``_prepared`` is still set here and the counter is one
        ahead of the last COMPLETED step -- the pre-gather refresh and ``flush()`` would replay a step that never
        happened.  Take the early advance back; the coming forward prepares the same step number again."""
        if self._prepared and not torch.cuda.is_current_stream_capturing():
            if int(self._t_step.item()) <= 1:
                # the very first step was the abandoned one: "last completed step" is step 0, for which no corrections exist
                # (1 - b1^0 = 0).  Restore the initial state instead of running prepare with t = 0.
                self._t_step.zero_()
                self._t_hyper[8:15].zero_()
                self._t_ring[:4].zero_()
            else:
                # counter back to (last completed step - 1), then the ordinary prepare launch: it re-forms the corrections
                # hyper[8..12] and the ring entry of the last completed step, bit for bit what they were
                self._t_step.sub_(2)
                _lib.call("rh_adam_prepare", ops._p(self._t_hyper), ops._p(self._t_step), ops._p(self._t_ring), self.RING,
                          ops._stream())
            self._prepared = False
            self._small_done = False
            if self.lazy_k > 1 and self._tables:
                self._gathers = 0  # the abandoned forward's gathers do not count towards the step's gather count
                self._pre_refreshed = None
                if self._touch_log:
                    # the abandoned backward scattered gradient rows that no optimizer step will consume (and re-zero)
                    for p in self._tables:
                        ops.grad_buffer(p).zero_()
                        p._rh_dirty = False
                del self._touch_log[:]
                del self._step_recs[:]

    def small_adam_args(self):
        """(sdesc, hyper) for rh_pack_grads_adam when the dense parameters' step may ride on the packing launch of this
        step -- i.e. when this step's Adam scalars are already on the device (fuse_prepare ran in the forward, so
        step_tables() would not launch rh_adam_prepare) -- else None.  The following step_tables() then skips rh_adam_small."""
        if self._bucket is None or not self._prepared or not self._groups_agree():
            return None
        self._small_done = True
        return self._s_desc, self._t_hyper

    def step_tables(self):
        """One Adam step over every table (+ in-pass re-zeroing of the gradient rows)."""
        if not self._tables and self._bucket is None:
            return
        stream = ops._stream()
        if self.lazy_k > 1 and self._tables:
            self._finish_sweep()
            if self._gathers:
                self._gathers_per_step, self._gathers = self._gathers, 0
            if self._step_recs:
                self._last_recs, self._step_recs = self._step_recs, []
                if getattr(self, "_ahead_broken", False):
                    self._last_recs = []
            seg = graphs.active()
            if seg is not None:
                if getattr(self, "_advance_seg", None) is not seg:  # replays count their steps on the host too (the
                    seg.after(self._advance_host_step)              # deferred sweep takes its step by value), after the
                    self._advance_seg = seg                         # last segment -- in-line captures as well: a trainer
                                                                    # may alternate between the two forms of the step
            elif not torch.cuda.is_current_stream_capturing():
                self._host_step += 1
        if self._prepared:
            self._prepared = False
        else:
            _lib.call("rh_adam_prepare", ops._p(self._t_hyper), ops._p(self._t_step), ops._p(self._t_ring), self.RING,
                      stream)
        if self._bucket is not None and self._small_done:
            self._small_done = False  # the packing launch of this step already stepped the dense parameters
        elif self._bucket is not None:
            b = self._bucket
            if not all(b.packed):
                raise RuntimeError("TableAdam: the dense gradient bucket was not packed (call bucket.finish() first)")
            _lib.call("rh_adam_small", ops._p(self._s_desc), len(b.params), ctypes.cast(self._s_numel, ctypes.c_void_p),
                      ops._p(b.flat), ops._p(self._t_hyper), stream)
        if not self._tables:
            return
        if self.lazy_k > 1:
            self._lazy_step(stream)
        else:
            for desc, n, numel in self._desc():
                _lib.call("rh_adam_dense", ops._p(desc), n, ctypes.cast(numel, ctypes.c_void_p),
                          ops._p(self._t_hyper), 1, stream)
        for p in self._tables:
            p._rh_dirty = False  # the kernels zeroed every non-zero gradient row
            if p.grad is None:
                p.grad = p._rh_grad

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._bucket is not None and not self._groups_agree():
            raise RuntimeError("TableAdam: per-group hyper-parameters differ; detach the bucket to use them")
        if self._tables or self._bucket is not None:
            if not torch.cuda.is_current_stream_capturing():
                self.sync_hyper()
            self.step_tables()
            self._flush_rider()
        dense_groups = [] if self._bucket is not None else [g for g in self.param_groups if not g.get("rh_tables")]
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
        self.rollback_abandoned_prepare()
        self.flush()
        if self._tables or self._bucket is not None:
            t = float(self._t_step.item())
            for p in self._tables + (self._bucket.params if self._bucket is not None else []):
                self.state[p]["step"] = torch.tensor(t)
        sd = super().state_dict()
        # padded-width tables (basic.initializers.PaddedEmbedding): the moments leave with the table's LOGICAL width, like
        # its weight in the model's state_dict -- the file is exchangeable with the reference optimizer's checkpoint
        for idx, width in self._padded_state_indices().items():
            st = sd["state"].get(idx)
            if st is not None:
                st = dict(st)
                for k in ("exp_avg", "exp_avg_sq"):
                    if torch.is_tensor(st.get(k)) and st[k].dim() == 2 and st[k].shape[1] > width:
                        st[k] = st[k][:, :width].contiguous()
                sd["state"][idx] = st
        return sd

    def _padded_state_indices(self):
        """{index of the parameter in state_dict()['state']: logical width} for padded-width tables."""
        out, i = {}, 0
        for g in self.param_groups:
            for p in g["params"]:
                w = getattr(p, "_rh_logical_dim", None)
                if w is not None and p.dim() == 2 and int(w) < int(p.shape[1]):
                    out[i] = int(w)
                i += 1
        return out

    def load_state_dict(self, state_dict):
        self.rollback_abandoned_prepare()
        if self.lazy_k > 1 and self._tables:
            self._join_sweep()
        padded = self._padded_state_indices()
        if padded:  # moments saved at the logical width (by this class or by the reference's optimizer): pad with zeros
            flat = [p for g in self.param_groups for p in g["params"]]
            state = dict(state_dict["state"])
            for idx, width in padded.items():
                st = state.get(idx)
                if st is None:
                    continue
                st = dict(st)
                for k in ("exp_avg", "exp_avg_sq"):
                    t = st.get(k)
                    if torch.is_tensor(t) and t.dim() == 2 and t.shape[1] == width:
                        full = t.new_zeros((t.shape[0], flat[idx].shape[1]))
                        full[:, :width] = t
                        st[k] = full
                state[idx] = st
            state_dict = dict(state_dict, state=state)
        super().load_state_dict(state_dict)
        if self._tables:
            for i, p in enumerate(self._tables):
                st = self.state[p]
                self._t_m[i].copy_(st["exp_avg"])
                self._t_v[i].copy_(st["exp_avg_sq"])
                st["exp_avg"], st["exp_avg_sq"] = self._t_m[i], self._t_v[i]
            t = int(float(self.state[self._tables[0]]["step"]))
            self._t_step.fill_(t)
            self._t_hyper_host = None
            if self.lazy_k > 1:  # a checkpoint is a flushed state: every row is at step t
                self._join_sweep()
                self._sweep_pending, self._flushed_at, self._host_step = False, t, t
                for last in self._t_last:
                    last.fill_(t)
        if self._bucket is not None:
            for i, p in enumerate(self._bucket.params):
                st = self.state[p]
                self._s_m[i].copy_(st["exp_avg"])
                self._s_v[i].copy_(st["exp_avg_sq"])
                st["exp_avg"], st["exp_avg_sq"] = self._s_m[i], self._s_v[i]
            self._t_step.fill_(int(float(self.state[self._bucket.params[0]]["step"])))
            self._t_hyper_host = None

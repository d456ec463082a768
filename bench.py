#!/usr/bin/env python
"""bench.py — CTR train samples/sec, DeepFM on Criteo-shape synthetic data, N x MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 100 --warmup 10
    python bench.py --gpus N --steps K --warmup W          (N > 1 without a torchrun environment: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (default, --model deepfm) = BASELINE.json configs[1]: DeepFM (tutorials/00 wiring: MLP on 13 dense + 26x16
sparse, FM + LR on the sparse block), 26 sparse fields with the Criteo cardinalities (33 762 577 rows, D=16, fp32,
2.01 GiB of tables), synthetic rows resident in HBM, B=4096 per GPU, the reference trainer's defaults (Adam lr 1e-3,
coupled weight_decay 1e-5, dropout 0.2) — i.e. a DENSE-exact optimizer step over every table row, as torch.optim.Adam
does in the reference.  A step = batch assembly + forward + BCE + backward + optimizer step.  Weak scaling:
per-GPU batch fixed, one process per GPU over RCCL.  --model dcnv2 | din | dssm time configs[2..4] on the same harness.

Timed region: EXACTLY --steps steps of the STEADY STATE of the blocked-lazy exact optimizer (at least lazy_k + 8 replayed
steps since the last flush, so every window sweep replays lazy_k steps and the lag distribution of the table rows is
the same at the start and at the end of the region: no deferred work enters or leaves it).  The end-of-epoch flush (all
rows brought to the last step) is timed separately (`flush_ms`) and checked afterwards (`min(last) == t`).

One JSON line on rank 0; `roofline` / `kernels` are measured live with HIP events around the kernels of the step (same
stream, same steady-state regime), `cpu_baseline` is oracle/cpu_port.py (the reference's op chain + its DataLoader
contract on eager torch CPU) timed on the host cores.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Criteo-Kaggle cardinalities as used by public DLRM configs; sum = 33 762 577
CRITEO_VOCABS = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306,
                 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572]
N_DENSE = 13
EMBED_DIM = 16
DEFERRED_SWEEP_PMC_TRAFFIC_K64 = 208.9e6  # bytes per launch of adam_lazy_sweep_kernel<4, false>: (2 * 51 561.3 + 100 850.6) KiB, profiles/r04_k64_pmc_sweep_{FETCH,WRITE}_SIZE.txt (round 3's passes gave 208.8e6: profiles/r03_pmc_sweep_*)
# per lazy_k: (bytes per launch of adam_lazy_sweep_kernel<4, false>, files under profiles/)
DEFERRED_SWEEP_PMC_TRAFFIC = {64: (DEFERRED_SWEEP_PMC_TRAFFIC_K64, "r04_k64_pmc_sweep"),
                              # lazy_k = 128: (2 * 25 830.0 + 50 508.5) KiB = 1.013 x the 103.25 MB of the window's rows
                              128: (104.6e6, "r04_pmc_sweep")}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable by a float4 copy

# algorithmic bytes: F=26, D=16, fp32, int64 indices as the loader holds them
FWD_BYTES_PER_SAMPLE = 26 * 8 + 26 * 16 * 4 + 26 * 16 * 4 + 8  # 3544
BWD_BYTES_PER_SAMPLE = 26 * 8 + 26 * 16 * 4 + 26 * 16 * 4 + 4 + 26 * 16 * 4  # 5204
ADAM_BYTES_PER_ELEM = 28  # read p,g,m,v + write p,m,v
GATHER_BYTES_PER_SAMPLE = 2 * (26 * 8 + N_DENSE * 4 + 4)
EPOCH_ROWS = 45_000_000  # BASELINE.json configs[1]: one epoch of the Criteo-shape dataset


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="deepfm", choices=["deepfm", "dcnv2", "din", "dssm"],
                    help="deepfm = BASELINE.json configs[1] (the headline); dcnv2 / din / dssm = configs[2] / [3] / [4]")
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch size")
    ap.add_argument("--rows", type=int, default=0, help="synthetic dataset rows resident per GPU (0 = per model)")
    ap.add_argument("--graph", default="auto", choices=["auto", "0", "1"], help="replay the step from a hipGraph")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "zipf"], help="index distribution")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=24.0, help="seconds of CPU baseline work (both legs together)")
    ap.add_argument("--vocab-scale", type=float, default=1.0, help="debug: shrink every table")
    ap.add_argument("--table-adam", default="lazy", choices=["lazy", "dense"],
                    help="how the dense-exact Adam over the tables is executed (results are bit-identical)")
    ap.add_argument("--lazy-k", type=int, default=None,
                    help="window divisor of the blocked-lazy exact Adam; default: 128 for steps of <= 8192 samples, 64 beyond (the trainer's own rule)")
    ap.add_argument("--tables", default="auto", choices=["auto", "replicate", "shard", "both"],
                    help="N > 1 placement of the embedding tables; both compute the reference's global-batch update. "
                         "replicate: one replica per rank, gradient rows all-gathered (nn.DataParallel's layout, SURVEY "
                         "8e); shard: one row-shard per rank, indices all-gathered and rows reduce-scattered (table "
                         "memory, optimizer state and sweep traffic / N).  auto / both (N > 1): time BOTH, report each "
                         "under `scaling_modes`, headline = the faster")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the data-parallel machinery (RCCL collectives) even on one GPU")
    ap.add_argument("--no-kernel-sweep", action="store_true", help="skip the gather-kernel batch sweep")
    ap.add_argument("--no-twin-repeat", dest="twin_repeat", action="store_false",
                    help="dense_twin_check: do not train the SECOND dense twin (twin-vs-twin = the comparison's noise floor)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the nested rocprofv3 --pmc passes (roofline.traffic)")
    ap.add_argument("--no-twin-check", action="store_true",
                    help="skip the dense-Adam twin that re-trains the same steps after the flush (dense_twin_check)")
    ap.add_argument("--brief", action="store_true",
                    help="headline only: skip batch_sweep / zipf / secondary_configs / step_accounting (SURVEY 8d extras)")
    ap.add_argument("--no-step-accounting", action="store_true", help="skip the nested rocprofv3 pass (step_accounting)")
    ap.add_argument("--acct-only", action="store_true",
                    help="of the extra lines (batch sweep, Zipf, secondary configs, step accounting) only the step accounting")
    ap.add_argument("--cpu-protocol", default="auto", choices=["auto", "full", "bounded"],
                    help="cpu_baseline: full = SURVEY 8(d)'s 3 warm-up + 10 timed steps per leg (~90 s at the Criteo shape); "
                         "bounded = about --cpu-budget seconds of steps; auto = full unless a step is too slow on this host")
    ap.add_argument("--trace-inner", action="store_true", help=argparse.SUPPRESS)  # child of the step_accounting pass
    ap.add_argument("--pmc-inner", action="store_true", help=argparse.SUPPRESS)  # child of the pmc_traffic passes
    ap.add_argument("--launch-dry-run", action="store_true",
                    help="exercise ONLY the rank launcher + rendezvous on CPU (gloo): every rank joins the group, one "
                         "all-reduce, rank 0 prints a JSON line with n_gpus = N and dry_run = true (tests/test_host_logic.py)")
    args = ap.parse_args()
    args.lazy_k_explicit = args.lazy_k is not None
    if args.lazy_k is None:
        args.lazy_k = 128 if args.batch <= 8192 else 64
    return args


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks here (one process per GPU,
    replacement of the reference's single-process nn.DataParallel, trainers/ctr_trainer.py:53-55) by re-executing this
    file under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1 and a free port.  The
    children's stdout (rank 0's ONE JSON line) is this process's stdout; the exit code is theirs.  Fewer than N visible
    devices is an error, never a silent one-rank measurement."""
    import socket
    import subprocess
    if not args.launch_dry_run:
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit(f"[bench] --gpus {args.gpus} but only {have} HIP device(s) are visible: refusing to measure fewer "
                     "ranks than asked for")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without a torchrun environment: launching {args.gpus} ranks: {' '.join(cmd)}",
          file=sys.stderr)
    sys.stdout.flush()
    sys.exit(subprocess.run(cmd, env=env).returncode)


def launch_dry_run(args, world, rank, result_fd):
    """The launcher path on CPU: gloo rendezvous of the ranks `self_launch` (or torchrun) started, one all-reduce."""
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    assert t.item() == world * (world + 1) / 2, (t.item(), world)
    if rank == 0:
        os.write(result_fd, (json.dumps({"dry_run": True, "n_gpus": world, "rccl_ranks": 0, "gloo_ranks": dist.get_world_size(),
                                         "gpus_asked": args.gpus, "steps": args.steps, "warmup": args.warmup}) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()


def criteo_columns(rows, vocabs, device, seed, dist_kind):
    g = torch.Generator(device=device).manual_seed(seed)
    sparse = torch.empty((rows, len(vocabs)), dtype=torch.int64, device=device)
    for j, v in enumerate(vocabs):
        if dist_kind == "uniform":
            sparse[:, j] = torch.randint(0, v, (rows,), generator=g, device=device)
        else:  # Zipf(1.05)-like via inverse CDF of a bounded power law
            u = torch.rand(rows, generator=g, device=device, dtype=torch.float64)
            a = 1.05
            x = ((v**(1 - a) - 1) * u + 1)**(1 / (1 - a))
            sparse[:, j] = (x - 1).clamp_(0, v - 1).long()
    dense = torch.rand((rows, N_DENSE), generator=g, device=device, dtype=torch.float32)
    label = (torch.rand(rows, generator=g, device=device) < 0.25).float()
    return sparse, dense, label


def padded_history(rows, L, vocab, g, device):
    lens = torch.randint(1, L + 1, (rows,), device=device, generator=g)
    h = torch.randint(1, vocab, (rows, L), device=device, generator=g)
    return h.masked_fill_(torch.arange(L, device=device)[None, :] >= lens[:, None], 0)


class Workload(object):
    """Feature lists, synthetic HBM-resident dataset and the model / trainer factory of one --model."""

    def __init__(self, args, device, rank):
        from torch_rechub_amd.basic.features import DenseFeature, SequenceFeature, SparseFeature
        self.args, self.device, self.name = args, device, args.model
        scale = args.vocab_scale
        g = torch.Generator(device=device).manual_seed(2022 + rank)
        self.match = False
        if self.name in ("deepfm", "dcnv2"):
            self.vocabs = [max(3, int(v * scale)) for v in CRITEO_VOCABS]
            self.dense_feas = [DenseFeature(f"I{i + 1}") for i in range(N_DENSE)]
            self.sparse_feas = [SparseFeature(f"C{i + 1}", vocab_size=v, embed_dim=EMBED_DIM)
                                for i, v in enumerate(self.vocabs)]
            rows = args.rows or 45_000_000
            self.sparse, self.dense, self.label = criteo_columns(rows, self.vocabs, device, 2022 + rank, args.dist)
            self.sparse_names = [f.name for f in self.sparse_feas]
            self.dense_names = [f.name for f in self.dense_feas]
            self.table_feas = self.sparse_feas
            self.desc = ("BASELINE.json configs[1]: DeepFM Criteo-shape synthetic, 26 sparse fields (33.76M rows total, "
                         "D=16) + 13 dense, MLP 429-256-128-1, fp32, dataset resident in HBM") if self.name == "deepfm" \
                else ("BASELINE.json configs[2]: DCN-v2 (CrossNetMix: 3 layers, rank 32, 4 experts; parallel DNN "
                      "429-256-128) on the same Criteo-shape synthetic data, fp32, dataset resident in HBM")
        elif self.name == "din":
            nu, ni, nc, L = int(200000 * scale) + 10, int(63001 * scale) + 10, 801, 100
            rows = args.rows or 1_000_000
            self.L = L
            self.feats = [SparseFeature("user_id", nu, 16)]
            self.hist = [SequenceFeature("hist_item", ni, 16, pooling="concat", shared_with="target_item", padding_idx=0),
                         SequenceFeature("hist_cate", nc, 16, pooling="concat", shared_with="target_cate", padding_idx=0)]
            self.tgt = [SparseFeature("target_item", ni, 16, padding_idx=0),
                        SparseFeature("target_cate", nc, 16, padding_idx=0)]
            self.sparse = torch.cat([torch.randint(0, nu, (rows, 1), device=device, generator=g),
                                     torch.randint(1, ni, (rows, 1), device=device, generator=g),
                                     torch.randint(1, nc, (rows, 1), device=device, generator=g),
                                     padded_history(rows, L, ni, g, device), padded_history(rows, L, nc, g, device)],
                                    dim=1).contiguous()
            self.sparse_names = ["user_id", "target_item", "target_cate", ("hist_item", L), ("hist_cate", L)]
            self.dense, self.dense_names = None, []
            self.label = (torch.rand(rows, device=device, generator=g) < 0.25).float()
            self.table_feas = self.feats + self.tgt
            self.desc = ("BASELINE.json configs[3]: DIN on Amazon-Electronics-shape synthetic (2 history fields x L<=100 "
                         "post-padded, 2 targets + user_id, D=16, attention MLP [256,128] Dice, MLP [256,128]), fp32")
        else:  # dssm
            nu, ni, L = int(10_000_000 * scale) + 10, int(100_000_000 * scale) + 10, 50
            rows = args.rows or 2_000_000
            self.L = L
            self.user = [SparseFeature("user_id", nu, 16),
                         SequenceFeature("hist_item", ni, 16, pooling="mean", shared_with="item_id", padding_idx=0)]
            self.item = [SparseFeature("item_id", ni, 16, padding_idx=0), SparseFeature("cate_id", 1000, 16)]
            self.sparse = torch.cat([torch.randint(0, nu, (rows, 1), device=device, generator=g),
                                     torch.randint(1, ni, (rows, 1), device=device, generator=g),
                                     torch.randint(0, 1000, (rows, 1), device=device, generator=g),
                                     padded_history(rows, L, ni, g, device)], dim=1).contiguous()
            self.sparse_names = ["user_id", "item_id", "cate_id", ("hist_item", L)]
            self.dense, self.dense_names = None, []
            self.label = torch.zeros(rows, device=device)
            self.table_feas = [self.user[0]] + self.item
            self.match = True
            self.desc = ("BASELINE.json configs[4]: DSSM two-tower, in-batch negatives (20 per row), 100M-item + "
                         "10M-user tables (D=16) resident in HBM, history L<=50 mean-pooled, towers [256,128,64] prelu")
        self.rows = rows

    def build(self, placement, use_graph, batch=None):
        """(model, trainer, loader); every call starts from fresh tables (Feature objects cache their nn.Embedding, Q2)."""
        from torch_rechub_amd.trainers import CTRTrainer, MatchTrainer
        from torch_rechub_amd.utils.data import DeviceDataLoader
        a, device = self.args, self.device
        feas = {"deepfm": lambda: self.sparse_feas, "dcnv2": lambda: self.sparse_feas,
                "din": lambda: self.feats + self.hist + self.tgt, "dssm": lambda: self.user + self.item}[self.name]()
        for f in feas:
            if hasattr(f, "embed"):
                del f.embed
        torch.manual_seed(2022)  # identical initial replica on every rank (and broadcast from rank 0 anyway)
        mlp = {"dims": [256, 128], "dropout": 0.2, "activation": "relu"}
        with torch.device(device):  # tables are created directly in HBM (never staged through the host)
            if self.name == "deepfm":
                from torch_rechub_amd.models.ranking import DeepFM
                m = DeepFM(self.dense_feas + self.sparse_feas, self.sparse_feas, mlp)
            elif self.name == "dcnv2":
                from torch_rechub_amd.models.ranking import DCNv2
                m = DCNv2(self.dense_feas + self.sparse_feas, 3, mlp)
                if os.environ.get("PROBE_DCN_BRANCHES") == "0":  # (A/B of the cross stack beside the MLP)
                    m.parallel_branches = False
            elif self.name == "din":
                from torch_rechub_amd.models.ranking import DIN
                m = DIN(self.feats, self.hist, self.tgt, mlp_params={"dims": [256, 128], "dropout": 0.2},
                        attention_mlp_params={"dims": [256, 128]})
                if os.environ.get("PROBE_DIN_BRANCHES") == "0":  # (A/B of the side-by-side activation units)
                    m.attention_branches = False
            else:
                from torch_rechub_amd.models.matching import DSSM
                tower = {"dims": [256, 128, 64], "activation": "prelu"}
                m = DSSM(self.user, self.item, user_params=dict(tower), item_params=dict(tower), temperature=0.02)
                if os.environ.get("PROBE_DSSM_BRANCHES") == "0":  # (A/B of the two towers side by side)
                    m.tower_branches = False
        # (a.lazy_k is the headline's value; other batch sizes of the sweep follow the trainer's own rule unless --lazy-k was given)
        # (None: the trainer's own rule -- 128, or 64 for steps of more than 8192 samples and for sweep-bound table sets)
        k = a.lazy_k if a.lazy_k_explicit else None
        kw = dict(device=str(device), show_progress=False, use_graph=use_graph, table_update=a.table_adam,
                  lazy_k=k, tables=placement)
        if self.match:
            t = MatchTrainer(m, mode=0, in_batch_neg=True, in_batch_neg_ratio=20, **kw)
        else:
            t = CTRTrainer(m, **kw)
        ld = DeviceDataLoader(self.sparse, self.sparse_names, self.dense, self.dense_names, self.label,
                              batch or a.batch, shuffle=True)
        ld.reshuffle()
        m.train()
        t.optimizer.sync_hyper()
        return m, t, ld


class KernelTimer(object):
    """HIP-event timing of individual C-ABI launches on the stream they run on (patched into _lib.call)."""

    def __init__(self, names):
        self.names = set(names)
        self.events = {n: [] for n in names}

    def install(self):
        from torch_rechub_amd import _lib
        self._orig = _lib.call
        timer = self

        def timed(name, *a):
            if name in timer.names:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = timer._orig(name, *a)
                e1.record()
                timer.events[name].append((e0, e1))
                return rc
            return timer._orig(name, *a)

        for mod in self._modules():
            mod._lib.call = timed
        _lib.call = timed

    def _modules(self):
        from torch_rechub_amd import ops, optim, sharding
        from torch_rechub_amd.utils import data
        return [m for m in (ops, optim, data, sharding) if hasattr(m, "_lib")]

    def remove(self):
        from torch_rechub_amd import _lib
        _lib.call = self._orig
        for mod in self._modules():
            mod._lib.call = self._orig

    def mean_ms(self):
        torch.cuda.synchronize()
        return {n: (sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else None) for n, ev in self.events.items()}

    def calls(self):
        return {n: len(ev) for n, ev in self.events.items()}


class CommTimer(object):
    """HIP-event timing of the collectives a step issues (eager profiling pass only): every torch.distributed collective
    is bracketed by events on the stream it is ISSUED on (main stream = exposed; the dense all-reduce runs on the
    bucket's side stream = overlapped), and the main stream's wait for the side stream (DenseGradBucket.join) is
    bracketed too (exposed tail of the overlapped all-reduce)."""

    FNS = ["all_reduce", "all_gather_into_tensor", "reduce_scatter_tensor", "all_gather", "broadcast"]

    def __init__(self, bucket):
        self.bucket, self.ev, self._orig = bucket, [], {}

    def _wrap(self, name, fn):
        timer = self

        def timed(*a, **k):
            cur = torch.cuda.current_stream()
            side = timer.bucket is not None and timer.bucket.side is not None and \
                cur.cuda_stream == timer.bucket.side.cuda_stream
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            if side:  # async all-reduce on the side stream: its bracket is closed when the main stream joins
                if timer.side_open is None:
                    timer.side_open = e0
                return out
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            timer.ev.append(("main_stream", e0, e1))
            return out

        return timed

    def install(self):
        self.side_open = None
        for n in self.FNS:
            if hasattr(dist, n):
                self._orig[n] = getattr(dist, n)
                setattr(dist, n, self._wrap(n, self._orig[n]))
        if self.bucket is not None:
            self._join = self.bucket.join
            timer = self

            def join():
                if timer.side_open is not None and timer.bucket.side is not None:
                    e1 = torch.cuda.Event(enable_timing=True)
                    with torch.cuda.stream(timer.bucket.side):
                        e1.record()  # after every collective queued on the side stream so far
                    timer.ev.append(("side_stream", timer.side_open, e1))
                    timer.side_open = None
                j0, j1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                j0.record()
                timer._join()
                j1.record()
                timer.ev.append(("join_wait", j0, j1))

            self.bucket.join = join

    def remove(self):
        for n, fn in self._orig.items():
            setattr(dist, n, fn)
        if self.bucket is not None and hasattr(self, "_join"):
            self.bucket.join = self._join

    def per_step_us(self, steps):
        torch.cuda.synchronize()
        out = {"main_stream": 0.0, "side_stream": 0.0, "join_wait": 0.0}
        for kind, e0, e1 in self.ev:
            try:
                out[kind] += e0.elapsed_time(e1) * 1e3
            except RuntimeError:
                pass
        return {k: round(v / max(steps, 1), 2) for k, v in out.items()}


def gather_sweep(model, wl, device, batches=(4096, 16384, 65536), rounds=8):
    """rh_embed_fwd (gather + FM + LR + dense concat) and rh_embed_bwd (gradient rows -> scatter-add + LR weight partials)
    alone at several batch sizes, on the real tables, packed (B, F) index layout as the loader hands it over.  The C-ABI
    launches (exactly the argument lists of ops._EmbedFused) are captured into ONE hipGraph per kernel -- `rounds` passes
    over several index sets -- and the graph replay is bracketed by HIP events on its stream: the figure is the kernel's
    launch-to-launch time as it runs inside the training step's graph, and it is what `rocprofv3 --kernel-trace` reports
    for these launches (profiles/).  Achieved GB/s is against the algorithmic bytes of SURVEY 8(d)."""
    from torch_rechub_amd import _lib, ops
    out = {}
    g = torch.Generator(device=device).manual_seed(7)
    w, b = model.linear.fc.weight.detach(), model.linear.fc.bias.detach()
    err = ops.err_flag(device)
    for B in batches:
        # several index sets, cycled: one set re-gathered in a loop would sit in the 256 MiB Infinity Cache (65536 x 26
        # rows = 109 MB) and flatter the kernel; training never looks up the same rows twice in a row
        nsets = max(2, min(16, -(-400_000_000 // (B * len(wl.vocabs) * 64))))
        calls = []
        for _ in range(nsets):
            idx = torch.stack([torch.randint(0, v, (B,), device=device, generator=g) for v in wl.vocabs], 1).contiguous()
            den = torch.rand(B, len(wl.dense_feas), device=device, generator=g)
            x = {f.name: idx[:, j] for j, f in enumerate(wl.sparse_feas)}
            x.update({f.name: den[:, j] for j, f in enumerate(wl.dense_feas)})
            call = model.embedding.make_call(x, wl.sparse_feas, wl.dense_feas, want_fm=True, want_lr=True)
            calls.append((call, call.fdesc(False), call.fdesc(True), call.idesc(), call.ddesc(), idx, den))
        c0 = calls[0][0]
        F, D, width = c0.F, c0.D, c0.width
        pitch = (width + 15) // 16 * 16
        emb = torch.empty((B, pitch), dtype=torch.float32, device=device)[:, :width]
        fm, lr = torch.empty(B, device=device), torch.empty(B, device=device)
        s_sum = torch.empty((B, D), dtype=torch.float32, device=device)
        g_out = torch.randn(B, width, device=device, generator=g)
        g_fm, g_lr = torch.randn(B, device=device, generator=g), torch.randn(B, device=device, generator=g)
        nch = _lib.call("rh_embed_bwd_nchunks", B, c0.samples_per_block)
        partial = torch.empty((nch, F * D), dtype=torch.float32, device=device)

        def fwd(c):
            _lib.call("rh_embed_fwd", ops._p(c[1]), ops._p(c[3]), c[0].idx_is_i64, B, F, D, ops._p(c[4]), len(c[0].dense),
                      c[0].dense_col, ops._p(emb), emb.stride(0), ops._p(w), ops._p(b), ops._p(lr), ops._p(fm),
                      ops._p(s_sum), c[0].field_split, ops._p(err), ops._stream())

        def bwd(c):
            _lib.call("rh_embed_bwd", ops._p(c[2]), ops._p(c[3]), c[0].idx_is_i64, B, F, D, ops._p(g_out), g_out.stride(0),
                      ops._p(emb), emb.stride(0), ops._p(s_sum), ops._p(g_fm), ops._p(g_lr), ops._p(w), ops._p(partial),
                      1.0, 0, ops._p(None), c[0].samples_per_block, ops._p(err), ops._stream())

        for w_ in c0.weights:  # the table-gradient buffers the backward scatters into
            ops.grad_buffer(w_)
        ent = {}
        side = torch.cuda.Stream(device=device)
        variants = [("rh_embed_fwd", fwd, FWD_BYTES_PER_SAMPLE), ("rh_embed_bwd", bwd, BWD_BYTES_PER_SAMPLE)]
        for name, fn, nbytes in variants:
            with torch.cuda.stream(side):
                for c in calls:
                    fn(c)
                side.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    for _ in range(rounds):
                        for c in calls:
                            fn(c)
                graph.replay()
                side.synchronize()
                best = None
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side)
                    graph.replay()
                    e1.record(side)
                    side.synchronize()
                    ms = e0.elapsed_time(e1) / (rounds * nsets)
                    best = ms if best is None else min(best, ms)
                del graph
            gbs = nbytes * B / (best * 1e-3) / 1e9
            ent[name] = {"avg_ms": round(best, 5), "achieved_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                         "launches_timed": rounds * nsets}
        for w_ in c0.weights:
            ops.grad_buffer(w_).zero_()
        ops.check_errors()
        # r01-compatible top-level fields = the forward
        ent.update({"avg_ms": ent["rh_embed_fwd"]["avg_ms"], "achieved_GBps": ent["rh_embed_fwd"]["achieved_GBps"],
                    "frac_of_hbm_peak": ent["rh_embed_fwd"]["frac"], "frac": ent["rh_embed_fwd"]["frac"],
                    "index_sets_cycled": nsets, "timing": "hipGraph of launches, HIP events around the replay"})
        out[str(B)] = ent
    return out


def cpu_model_string():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def _dense_twin(args, wl, loader, device, rng0, T):
    """A table_update="dense" trainer (rh_adam_dense over every row every step) advanced T steps over the batches the timed
    trainer consumed: same initial weights (wl.build reseeds), same permutation, same dropout seed and call counter."""
    import copy

    from torch_rechub_amd import ops
    a2 = copy.copy(args)
    a2.table_adam = "dense"
    keep = wl.args
    wl.args = a2
    try:
        ops._dropout_rng(device).copy_(rng0)
        m2, t2, ld2 = wl.build(None, True, batch=loader.batch_size)
    finally:
        wl.args = keep
    ld2.perm.copy_(loader.perm)
    ld2.pos.zero_()
    done = 0
    while done < T:
        if T - done < 3 and t2._graph is None:  # (fewer steps than the capture's eager warm-up)
            x, y = ld2.load_next()
            t2.train_step(x, y)
            done += 1
        else:
            done += t2._graphed_step(ld2)[1]
    t2.flush()
    torch.cuda.synchronize()
    if int(t2.optimizer._t_step.item()) != T:
        raise RuntimeError(f"dense twin ran {int(t2.optimizer._t_step.item())} steps, wanted {T}")
    return m2, t2, ld2


def _compare_tables(o1, o2, names, col_of, sparse, looked, device, n_sample, what1="timed trainer", raise_on_mismatch=True):
    """n_sample table rows (drawn over all tables in proportion to their size): rows no batch looked up -> weight, exp_avg,
    exp_avg_sq bit-equal; looked-up rows -> distribution of the weights' differences."""
    g = torch.Generator(device=device).manual_seed(77)
    total = sum(int(p.shape[0]) for p in o1._tables)
    out = {"rows_sampled": 0, "untouched_rows": 0, "touched_rows": 0, "untouched_bitwise_equal": True}
    diffs, scales = [], []
    for p1, p2 in zip(o1._tables, o2._tables):
        rows = int(p1.shape[0])
        n = min(rows, max(16, int(round(n_sample * rows / total))))
        idx = torch.randperm(rows, device=device, generator=g)[:n] if rows <= 4 * n else \
            torch.randint(0, rows, (n,), device=device, generator=g).unique()
        touched = torch.isin(idx, sparse[looked, col_of[names[id(p1)]]])
        un, to = idx[~touched], idx[touched]
        out["rows_sampled"] += int(idx.numel())
        out["untouched_rows"] += int(un.numel())
        out["touched_rows"] += int(to.numel())
        for what, x1, x2 in (("weight", p1.detach(), p2.detach()), ("exp_avg", o1.state[p1]["exp_avg"], o2.state[p2]["exp_avg"]),
                             ("exp_avg_sq", o1.state[p1]["exp_avg_sq"], o2.state[p2]["exp_avg_sq"])):
            if un.numel() and not torch.equal(x1[un], x2[un]):
                out["untouched_bitwise_equal"] = False
                d = (x1[un] - x2[un]).abs()
                msg = (f"dense twin: {what} of table {names[id(p1)]} ({rows} rows): {int((d > 0).sum())} elements of "
                       f"{int(un.numel())} never-looked-up rows of the {what1} differ from dense Adam (max {float(d.max()):.3e})")
                if raise_on_mismatch:
                    raise RuntimeError(msg)
                out.setdefault("mismatch", msg)
        if to.numel():
            diffs.append((p1.detach()[to] - p2.detach()[to]).abs().reshape(-1))
            scales.append(p2.detach()[to].abs().reshape(-1))
    if diffs:
        d, sc = torch.cat(diffs).double(), torch.cat(scales).double()
        qs = torch.tensor([0.5, 0.9, 0.99], dtype=torch.float64, device=device)
        out["touched_elements"] = int(d.numel())
        out["touched_abs_diff_p50_p90_p99_max"] = [float(f"{v:.3e}") for v in torch.quantile(d, qs).tolist() + [float(d.max())]]
        out["touched_abs_value_p50"] = float(f"{float(sc.median()):.3e}")
        out["touched_within_2e-5_plus_1e-3_frac"] = round(float((d <= 2e-5 + 1e-3 * sc).double().mean()), 5)
        out["touched_within_10pct_of_value_frac"] = round(float((d <= 1e-5 + 0.1 * sc).double().mean()), 5)
    return out


def dense_twin_check(args, wl, trainer, loader, device, rng0, n_sample=65536):
    """After the flush (outside every timed region): a table_update="dense" twin -- same initial weights, same batches in
    the same order, same dropout masks, every table row stepped by rh_adam_dense every step, which IS what the reference's
    torch.optim.Adam does (trainers/ctr_trainer.py:59-61, 99) -- is advanced exactly as many steps as the timed
    trainer has taken, and n_sample table rows (drawn over all tables in proportion to their size) are compared, weights and
    both Adam moments:
    * rows no batch looked up (the bulk of a 10 M-row table): their whole history is `g = wd * p` steps, which the lazy
      optimizer replayed in registers up to K steps late -- in the window sweeps beside the chain, in the step-ahead launch's
      refresh / lookahead parts and in the final flush.  They must equal the twin's BIT FOR BIT (raises otherwise);
    * rows some batch looked up: their gradient is a sum of float atomics in hardware order on both sides, the small tables'
      noise reaches every dense weight and comes back through every gradient, and Adam divides by sqrt(v): two trainings of
      the SAME code drift apart over hundreds of steps.  Their agreement is REPORTED as a distribution next to the same
      distribution between two dense twins (`twin_vs_twin`, --twin-repeat: the noise floor of the comparison).
    Returns the dict that goes into the JSON line."""
    opt = trainer.optimizer
    T = int(opt._t_step.item())
    B = loader.batch_size
    if int(loader.pos.item()) != (T * B) % loader.N or T * B > loader.N:
        return {"skipped": f"loader position {int(loader.pos.item())} is not steps x batch = {T} x {B} (wrapped or moved)"}
    t0 = time.perf_counter()
    m2, t2, ld2 = _dense_twin(args, wl, loader, device, rng0, T)
    looked = loader.perm[:T * B]
    names = {id(m_.weight): n for n, m_ in trainer.model.embedding.embed_dict.items()}
    col_of = {n: i for i, n in enumerate(wl.sparse_names)}
    out = {"steps": T}
    out.update(_compare_tables(opt, t2.optimizer, names, col_of, wl.sparse, looked, device, n_sample))
    if args.twin_repeat:
        m3, t3, ld3 = _dense_twin(args, wl, loader, device, rng0, T)
        names3 = {id(m_.weight): n for n, m_ in m2.embedding.embed_dict.items()}
        out["twin_vs_twin"] = _compare_tables(t2.optimizer, t3.optimizer, names3, col_of, wl.sparse, looked, device, n_sample,
                                              what1="first dense twin", raise_on_mismatch=False)
        del m3, t3, ld3
    out["twin_seconds"] = round(time.perf_counter() - t0, 2)
    out["what"] = ("table_update='dense' twin (rh_adam_dense over every row every step = torch.optim.Adam semantics) advanced the "
                   "same steps on the same batches after the timed trainer's flush; sampled rows no batch looked up: weight, "
                   "exp_avg, exp_avg_sq bit-equal (asserted); looked-up rows: float-atomic gradient sums on both sides, "
                   "distribution of |difference| reported")
    del m2, t2, ld2
    torch.cuda.empty_cache()
    return out


def run_mode(args, wl, placement, use_graph, world, rank, device, profile):
    """Build, warm up into the steady state, time exactly --steps steps; optionally the per-kernel eager pass; then the
    flush (timed separately) and the no-row-behind check.  Returns a dict."""
    from torch_rechub_amd import ops
    model, trainer, loader = wl.build(placement, use_graph, batch=args.batch)
    return _run_mode(args, wl, placement, use_graph, world, rank, device, profile, model, trainer, loader)


def _run_mode(args, wl, placement, use_graph, world, rank, device, profile, model, trainer, loader):
    from torch_rechub_amd import ops
    B = args.batch
    opt = trainer.optimizer
    lazy = getattr(opt, "lazy_k", 0) > 1
    res = {"tables": placement}
    graph_ok = False
    rng0 = ops._dropout_rng(device).clone()  # (seed, call counter) of the fused dropout before the first step: dense_twin_check

    def eager_step():
        x, y = loader.load_next()
        return trainer.train_step(x, y)

    if use_graph:
        try:
            trainer._graphed_step(loader)  # 3 eager warm-up steps + capture
            graph_ok = True
        except Exception as e:  # capture unsupported -> eager; say so in the JSON
            if rank == 0:
                print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); falling back to eager",
                      file=sys.stderr)
            torch.cuda.synchronize()
            trainer._graph = None
            trainer.use_graph = False

    def step():
        if graph_ok:
            trainer._graphed_step(loader)
        else:
            eager_step()

    def fence():
        if hasattr(opt, "release_gate"):
            opt.release_gate()  # nothing is enqueued behind this point: the last deferred sweep need not sit out its hold-back
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up INTO the steady state: after a flush the first lazy_k window sweeps replay 1, 2, ... steps only
    k_eff = int(getattr(opt, "lazy_k", 0))  # (decided by the optimizer at its first step unless --lazy-k was given)
    res["lazy_k"] = k_eff
    warm = max(args.warmup, (k_eff + 8) if lazy else 0)
    if lazy and graph_ok and trainer.dp is None:
        # + the trainer's self-tuning of the step's form (deferred / in-line sweep, residency cap): it starts once the
        # optimizer is in its steady state and must be over before the timed region
        warm += trainer.tune_budget_steps() + 4
    for _ in range(warm):
        step()
    if getattr(trainer, "_tune", None) and trainer._tune.get("active"):
        raise RuntimeError("the trainer's step-form tuning is still running at the start of the timed region")
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0  # host time to ENQUEUE the steps (diagnostic: close to dt = launch-bound host)
    fence()
    dt = time.perf_counter() - t0
    res["host_enqueue_ms_per_step"] = 1e3 * t_enq / args.steps
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    res.update(dt=dt, warmup_effective=warm, hipgraph=graph_ok, ms_per_step=1e3 * dt / args.steps,
               value=world * B * args.steps / dt)
    if profile and world == 1 and args.steps < 200 and not args.acct_only:
        # The closing fence of a K-step region also waits for the LAST step's deferred sweep (~0.23 ms of side-stream work
        # that overlaps the NEXT step in a longer run): at K = 20 that is ~13 us per step of the figure above.  The same
        # loop over 300 steps, reported beside it (never as `value`).
        n_steady = 300
        fence()
        s0 = time.perf_counter()
        for _ in range(n_steady):
            step()
        fence()
        sdt = time.perf_counter() - s0
        res["steady"] = {"steps": n_steady, "ms_per_step": round(1e3 * sdt / n_steady, 4),
                         "value": round(world * B * n_steady / sdt, 1), "unit": "samples/s",
                         "note": f"the same replayed step timed over {n_steady} steps right after the {args.steps}-step region: the "
                                 "closing fence's wait for the last deferred sweep is amortised (the K-step figure carries it once)"}

    # ---- the deferred window sweep in the GRAPH regime: it is launched eagerly on the side stream after every replay,
    # so HIP events on that stream bracket it while the captured chain runs beside it (what rocprofv3 shows for it) ----
    if profile and graph_ok and lazy and getattr(opt, "overlap_sweep", False) and trainer.dp is None and world == 1:
        # (single process only: with a process group every step is a collective, and only rank 0 profiles)
        t2 = KernelTimer(["rh_adam_lazy_sweep"])
        t2.install()
        for _ in range(30):
            step()
        res["deferred_sweep_ms"] = t2.mean_ms().get("rh_adam_lazy_sweep")
        t2.remove()
    # ---- per-kernel HIP-event timing, eager launches of the same step in the SAME regime (no flush in between) ----
    n_prof = max(8, min(args.steps, 30))
    if profile:
        names = ["rh_embed_fwd", "rh_embed_bwd", "rh_adam_dense", "rh_adam_lazy_touched", "rh_adam_lazy_sweep", "rh_adam_lazy_step",
                 "rh_adam_lazy_step_mode",
                 "rh_adam_lazy_step_mode",
                 "rh_batch_gather", "rh_embed_scatter_rows", "rh_shard_localize", "rh_seq_pool_fwd", "rh_seq_pool_bwd"]
        timer = KernelTimer(names)
        comm = CommTimer(trainer.bucket) if trainer.dp is not None else None
        overlap = None  # the eager pass launches what the step launches (deferred sweep on its side stream included)
        timer.install()
        if comm is not None:
            try:
                comm.install()
            except Exception as e:  # noqa: BLE001
                print(f"[bench] comm timer not installed: {e}", file=sys.stderr)
                comm = None
        for _ in range(n_prof):
            eager_step()
        res["kernel_ms"] = timer.mean_ms()
        res["kernel_calls_per_step"] = {n: round(c / n_prof, 2) for n, c in timer.calls().items() if c}
        for d_ in (res["kernel_ms"], res["kernel_calls_per_step"]):  # the merged end-of-step launch under its round-2 name
            if d_.get("rh_adam_lazy_step_mode") is not None:
                d_["rh_adam_lazy_step"] = d_.pop("rh_adam_lazy_step_mode")
            else:
                d_.pop("rh_adam_lazy_step_mode", None)
        timer.remove()
        if comm is not None:
            comm.remove()
            res["comm_us_per_step"] = comm.per_step_us(n_prof)
            res["comm_us_per_step"]["note"] = ("eager pass, HIP events on the issuing stream: main_stream = exposed "
                                               "collectives (row / index exchange), side_stream = dense all-reduce under "
                                               "the embedding backward, join_wait = its exposed tail")
        if overlap:
            opt.overlap_sweep = overlap
    elif world > 1:
        for _ in range(n_prof):  # keep the collectives of the profiling pass matched on every rank
            eager_step()

    # ---- the end-of-epoch flush, timed on its own, then: no row may be behind the step counter ----
    torch.cuda.synchronize()
    f0 = time.perf_counter()
    trainer.flush()
    torch.cuda.synchronize()
    res["flush_ms"] = round(1e3 * (time.perf_counter() - f0), 4)
    if lazy and opt._tables:
        t_now = int(opt._t_step.item())
        behind = min(int(last.min().item()) for last in opt._t_last)
        res["rows_behind_after_flush"] = t_now - behind
        if behind != t_now:
            raise RuntimeError(f"flush left rows behind: step counter {t_now}, min(last) {behind}")
    ops.check_errors(device)
    if profile and lazy and graph_ok and trainer.dp is None and world == 1 and wl.name in ("deepfm", "dcnv2") and \
            not args.no_twin_check and not args.acct_only:
        res["dense_twin_check"] = dense_twin_check(args, wl, trainer, loader, device, rng0)
        print(f"[bench] dense twin check: {res['dense_twin_check']}", file=sys.stderr)
    res["total_elems"] = sum(p.numel() for p in opt._tables) if hasattr(opt, "_tables") else 0
    # lazy sweep: bytes one launch must move = its 1/K window of every table (read + write p, m, v; 4 B/row of `last`
    # both ways) + the K=1 (small) tables in full incl. their gradient.  The other 1 - 1/K of the dense pass's traffic
    # is replaced by replay arithmetic, which is what bounds this kernel (VALU).
    sweep_bytes = step_bytes = 0
    for p_ in (opt._tables if hasattr(opt, "_tables") else []):
        rows, d_ = p_.shape
        k_ = opt.table_k(p_)
        win = -(-rows // k_)
        sweep_bytes += win * (d_ * 4 * (7 if k_ == 1 else 6) + 8)
        if k_ != 1:  # the deferred sweep visits the lazy tables' window only: p, m, v both ways + the last-step word
            res["deferred_sweep_bytes"] = res.get("deferred_sweep_bytes", 0) + win * (d_ * 4 * 6 + 8)
        # merged launch (rh_adam_lazy_step): the sweep also reads the gradient row of every window row, and the rows
        # the batch touched (one lookup per field and sample, counted once each: an upper bound under duplicates) are
        # read and written with their gradient: p, m, v, g both ways + index + last-step word
        # ... in the DEFERRED form (overlap_sweep) the end-of-step launch (rh_adam_lazy_step_mode) walks only the dense (K = 1)
        # tables in full plus the touched rows of the lazy ones; the lazy tables' window belongs to the side-stream sweep.
        # (Round 4 counted the window here too: 0.8455 of the HBM peak for a launch that moved a third of those bytes.)
        deferred_form = bool(getattr(opt, "overlap_sweep", False))
        step_bytes += (win * (d_ * 4 * 7 + 8) if (k_ == 1 or not deferred_form) else 0) + \
            (0 if k_ == 1 else min(B, rows) * (d_ * 4 * 8 + 16))
    res["sweep_bytes"] = sweep_bytes
    res["step_bytes"] = step_bytes
    res["overlap_sweep"] = bool(getattr(opt, "overlap_sweep", False))
    tune = getattr(trainer, "_tune", None) or {}
    res["step_form"] = {"chosen": {"form": tune["chosen"][0], "deferred_sweep_workgroups": tune["chosen"][1],
                                   "sweep_hold_back_ns": tune["chosen"][2]},
                        "candidates": [{"form": c[0], "deferred_sweep_workgroups": c[1], "sweep_hold_back_ns": c[2]}
                                       for c in tune["cands"]],
                        "ms_per_step_during_tuning": tune.get("ms")} if tune.get("chosen") else \
        {"chosen": {"form": getattr(trainer, "_form", None), "pinned": True}}
    if getattr(trainer, "short_sweep_inline", False):
        # row-sharded tables: the rank's shard is small enough for the window sweep to run in line (optim.TableAdam.
        # prefer_inline_for_short_sweeps), chosen by the trainer before its first step -- nothing was pinned by the caller
        res["step_form"]["chosen"].update(pinned=False, rule="short per-rank sweep: in line", lazy_k=int(getattr(opt, "lazy_k", 0)))
    # the north-star kernels over batch sizes (same tables, same stream, HIP events): their bandwidth regime starts
    # where the launch is no longer three dependent memory round trips long.  Last: it leaves junk gradient rows behind.
    if profile and wl.name == "deepfm" and trainer.dp is None and not args.no_kernel_sweep:
        try:
            res["gsweep"] = gather_sweep(model, wl, device)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] gather sweep failed: {type(e).__name__}: {e}", file=sys.stderr)
        if hasattr(opt, "_touch_log"):
            del opt._touch_log[:]
    if trainer.dp is not None:
        trainer.dp.close()
    return res


def _short_run(args, device, rank, use_graph, model=None, batch=None, dist_kind=None, rows=None, steps=30, wl=None):
    """One short steady-state measurement of another shape (same code path as the headline: run_mode with its own
    warm-up into the steady state, exactly `steps` timed graph replays, flush verified).  Returns (dict, workload)."""
    import copy
    a = copy.copy(args)
    a.steps, a.no_kernel_sweep = steps, True
    if model is not None:
        a.model = model
    if batch is not None:
        a.batch = batch
    if dist_kind is not None:
        a.dist = dist_kind
    if rows is not None:
        a.rows = rows
    if wl is None:
        wl = Workload(a, device, rank)
    else:
        wl.args = a
    r = run_mode(a, wl, None, use_graph, 1, rank, device, profile=False)
    out = {"batch": a.batch, "ms_per_step": round(r["ms_per_step"], 4), "value": round(r["value"], 1), "unit": "samples/s",
           "step_form": (r.get("step_form") or {}).get("chosen"),
           "steps": steps, "warmup_effective": r["warmup_effective"], "hipgraph": r["hipgraph"], "flush_ms": r["flush_ms"],
           "rows_behind_after_flush": r.get("rows_behind_after_flush")}
    torch.cuda.empty_cache()
    return out, wl


ACCOUNT_GROUPS = (  # first match wins; kernel-name substring -> row of SURVEY 8(d)'s whole-step accounting
    ("batch_gather_kernel", "batch_assembly"), ("batch_advance", "batch_assembly"),
    ("refresh_assemble", "batch_assembly_and_table_refresh"),
    ("adam_lazy_step_ahead", "touched_rows_step_and_next_batch_assembly_refresh"),
    ("adam_lazy_touched", "table_refresh_before_gather"), ("embed_fwd", "gather_fwd"), ("embed_bwd", "gather_bwd"),
    ("embed_scatter", "gather_bwd"), ("Cijk_", "mlp_gemm_library"), ("gemm_f32", "mlp_gemm_own"),
    ("linear_fwd", "mlp_gemm_own"), ("linear_dgrad", "mlp_gemm_own"),
    ("linear_wgrad", "mlp_weight_gradient"), ("wgrad", "mlp_weight_gradient"), ("bn_", "mlp_batchnorm_relu_dropout"),
    ("head_", "head_and_loss"), ("step_scalars", "step_scalars"), ("bce_", "head_and_loss"),
    ("pack_grads", "pack_and_dense_param_adam"), ("adam_small", "pack_and_dense_param_adam"),
    ("adam_lazy_sweep", "table_optimizer"), ("adam_lazy_step", "table_optimizer"), ("adam_dense", "table_optimizer"),
    ("adam_prepare", "step_scalars"), ("cross", "cross_network"), ("moe", "cross_network"), ("dice", "attention_mlp"),
    ("din_", "attention_mlp"), ("seq_pool", "sequence_pooling"), ("inbatch", "inbatch_negatives"), ("prelu", "towers"),
    ("nccl", "rccl"), ("rccl", "rccl"), ("copyBuffer", "memcpy"),
)


def _account_group(name):
    for key, grp in ACCOUNT_GROUPS:
        if key in name:
            return grp
    return "other"


def parse_step_trace(trace_csv, steps_wanted):
    """Per-step accounting from a rocprofv3 kernel trace: the last `steps_wanted` steps, a step = everything from one
    batch_gather_kernel launch (the first kernel of a replayed step) to the next."""
    import csv
    rows = []
    with open(trace_csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    rows = [r for r in rows if "stream_delay" not in r[2]]  # (the one-lane hold-back in front of the deferred sweep)
    marks = [i for i, r in enumerate(rows) if "batch_gather_kernel" in r[2] or "refresh_assemble" in r[2]]
    ahead = sum(1 for r in rows if "step_ahead" in r[2])
    if ahead > len(marks):  # step-ahead form: the head of a step is the LAST launch of the one before; a step starts at its gather
        marks = [i for i, r in enumerate(rows) if "embed_fwd_kernel" in r[2]]
    if len(marks) < 3:
        return None
    marks = marks[-(min(steps_wanted, len(marks) - 1) + 1):]
    n = len(marks) - 1
    wall = (rows[marks[-1]][0] - rows[marks[0]][0]) / n / 1e3
    per_kernel, order = {}, []
    # the step's chain runs on the queue of its scalar launch / gather; the optimizer's queue (the deferred sweep -- and,
    # since round 4, the head segment in front of it: batch assembly + refresh) is the other one
    main_q = next((r[3] for r in rows[marks[0]:marks[-1]] if "step_scalars" in r[2] or "embed_fwd" in r[2]), rows[marks[0]][3])
    for i in range(marks[0], marks[-1]):
        st, en, name, q = rows[i]
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("rechub::", "").split("(")[0][:60]
        head_seg = q != main_q and ("batch_gather" in name or "adam_lazy_touched" in name or "refresh_assemble" in name)
        if head_seg:
            short += " [head segment, optimizer queue]"
        elif q != main_q:
            short += " [side stream]"
        if short not in per_kernel:
            per_kernel[short] = [0, 0.0, name]
            order.append(short)
        per_kernel[short][0] += 1
        per_kernel[short][1] += (en - st) / 1e3
    groups = {}
    kernels = []
    busy = 0.0
    for k in order:
        cnt, tot, full = per_kernel[k]
        us = tot / n
        side = k.endswith("[side stream]")
        if not side:  # (the head segment gates the chain: the chain waits for the event behind the refresh)
            busy += us
        grp = _account_group(full) + ("_deferred_on_side_stream_overlapped" if side else "")
        groups[grp] = round(groups.get(grp, 0.0) + us, 2)
        kernels.append({"kernel": k, "launches_per_step": round(cnt / n, 2), "us_per_step": round(us, 2),
                        "avg_us": round(tot / cnt, 2), "group": grp})
    groups["idle_between_kernels"] = round(wall - busy, 2)
    return {"steps_averaged": n, "wall_us_per_step": round(wall, 2), "kernel_launches_per_step": round(sum(
        k["launches_per_step"] for k in kernels), 1), "groups_us_per_step": groups, "kernels": kernels}


def step_accounting(args):
    """SURVEY 8(d) whole-step accounting from ONE regime: a nested `rocprofv3 --kernel-trace` run of this file's
    --trace-inner mode (same model, batch, optimizer and hipGraph replay as the headline; a smaller resident dataset),
    parsed into in-graph kernel durations per step.  H2D is 0 by construction (dataset resident in HBM) and there is no
    RCCL at N = 1.  Returns None (with a note on stderr) when rocprofv3 cannot be nested."""
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    if any(k in os.environ for k in ("ROCPROFILER_LIBRARY_CTOR", "ROCP_TOOL_LIBRARIES", "ROCPROF_OUTPUT_PATH")) or \
            "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return {"skipped": "this process already runs under rocprofv3: see profiles/ for the kernel trace of this command"}
    tmp = tempfile.mkdtemp(prefix="rh_acct_", dir="/tmp")
    here = os.path.abspath(__file__)
    steps = 40
    cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "acct", "--", sys.executable, here,
           "--trace-inner", "--steps", str(steps), "--warmup", str(args.warmup), "--model", args.model, "--batch",
           str(args.batch), "--rows", str(min(args.rows or 4_000_000, 4_000_000)), "--lazy-k", str(args.lazy_k),
           "--table-adam", args.table_adam, "--dist", args.dist, "--vocab-scale", str(args.vocab_scale), "--graph", args.graph]
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
        files = glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True)
        if p.returncode != 0 or not files:
            return {"error": f"nested rocprofv3 run failed (rc {p.returncode}): {p.stderr.decode(errors='replace')[-300:]}"}
        acct = parse_step_trace(max(files, key=os.path.getsize), steps - 2)
        if acct is None:
            return {"error": "no steps found in the kernel trace"}
        acct.update({"source": "nested `rocprofv3 --kernel-trace -- python bench.py --trace-inner` (in-graph kernel durations "
                               "of hipGraph-replayed steady-state steps; the step under the tracer is a few us longer than "
                               "ms_per_step of the untraced headline)", "h2d_us_per_step": 0.0, "rccl_us_per_step": 0.0,
                     "optimizer_mode": f"table_adam={args.table_adam} lazy_k={args.lazy_k}",
                     "pass_seconds": round(time.perf_counter() - t0, 1)})
        return acct
    except subprocess.TimeoutExpired:
        return {"error": "nested rocprofv3 run timed out"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_traffic(args, kernel_key="adam_lazy_sweep_wide_kernel", tail=25):
    """HBM traffic per launch of the dominant kernel, RE-COLLECTED (round 4 reported a constant from profiles/): two nested
    `rocprofv3 --pmc <C> --kernel-trace` passes -- FETCH_SIZE and WRITE_SIZE need 3 + 2 of the 4 TCC slots, so one pass each;
    kernel trace only, no other trace domain -- over this file's --pmc-inner mode (the deferred sweep of the workload's tables in
    its steady state, launched eagerly with nothing beside it: see pmc_inner), mean over the last `tail` dispatches.  Bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE counts 64 B per
    128-B request of a 16 B / lane streaming read (MI355X_MICROARCH.md, HBM section; calibrated on rh_adam_dense in round 1:
    profiles/r01_pmc_traffic.md).  Returns a dict with `bytes` or `error`."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    if any(k in os.environ for k in ("ROCPROFILER_LIBRARY_CTOR", "ROCP_TOOL_LIBRARIES", "ROCPROF_OUTPUT_PATH")) or \
            "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return {"skipped": "this process already runs under rocprofv3"}
    here = os.path.abspath(__file__)
    out = {"kernel": kernel_key, "dispatches_averaged": tail}
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="rh_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "pmc", "--", sys.executable,
               here, "--pmc-inner", "--lazy-k", str(args.lazy_k), "--vocab-scale", str(args.vocab_scale)]
        try:
            p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=75)
            files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return {"error": f"nested rocprofv3 --pmc {counter} failed (rc {p.returncode}): {p.stderr.decode(errors='replace')[-200:]}"}
            vals = []
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter and kernel_key in row["Kernel_Name"]:
                        vals.append(float(row["Counter_Value"]))
            if len(vals) < tail:
                return {"error": f"only {len(vals)} dispatches of {kernel_key} in the {counter} pass"}
            out[counter + "_KiB"] = round(sum(vals[-tail:]) / tail, 1)
        except subprocess.TimeoutExpired:
            return {"error": f"nested rocprofv3 --pmc {counter} timed out"}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    out["bytes"] = round((2.0 * out["FETCH_SIZE_KiB"] + out["WRITE_SIZE_KiB"]) * 1024.0, 0)
    out["formula"] = "(2 x FETCH_SIZE + WRITE_SIZE) KiB, gfx950 correction of the streaming read"
    out["pass_seconds"] = round(time.perf_counter() - t0, 1)
    return out


def dp_one_rank(args):
    """The data-parallel step (RCCL collectives, gradient-row exchange or row-sharded lookup) measured on ONE rank: this file
    re-executed with --force-dp (a one-rank `nccl` group; every collective is launched, none is skipped) once per table
    placement, 100 replayed steps each.  NOT a scaling figure -- no scaling curve has been measured (the driver launches
    N = 2 / 4 / 8 when it has the node) -- but the per-rank cost of the data-parallel machinery next to the N = 1 step:
    `ratio_to_single` = its ms/step over this run's ms_per_step.  Reference: nn.DataParallel (trainers/ctr_trainer.py:53-55)."""
    import subprocess
    here = os.path.abspath(__file__)
    out = {}
    for placement in ("replicate", "shard"):
        cmd = [sys.executable, here, "--force-dp", "--tables", placement, "--steps", "100", "--warmup", str(args.warmup),
               "--no-cpu-baseline", "--brief", "--no-kernel-sweep", "--model", args.model, "--batch", str(args.batch),
               "--rows", str(min(args.rows or 8_000_000, 8_000_000)), "--dist", args.dist, "--graph", args.graph]
        # (a child of a torchrun-launched rank must not inherit the launcher's rendezvous: with TORCHELASTIC_USE_AGENT_STORE it
        # would wait for an agent store at ITS master port until the time-out -- 2 x 240 s of a `torchrun --nproc-per-node 1` run)
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_") and k not in (
            "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE", "TORCH_NCCL_ASYNC_ERROR_HANDLING")}
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 200), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        try:
            p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            line = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not line:
                out[placement] = {"error": f"rc {p.returncode}: {p.stderr.decode(errors='replace')[-300:]}"}
                continue
            d = json.loads(line[-1])
            out[placement] = {"ms_per_step": d["ms_per_step"], "value": d["value"], "hipgraph": d["config"]["hipgraph"],
                              "comm_us_per_step": d.get("comm_us_per_step"), "flush_ms": d.get("flush_ms"),
                              "rows_behind_after_flush": d.get("rows_behind_after_flush")}
        except subprocess.TimeoutExpired:
            out[placement] = {"error": "timed out"}
    return out


def pmc_inner(args, device):
    """Child of pmc_traffic(): the deferred window sweep of the workload's tables and NOTHING else -- the tables of
    configs[1] under TableAdam(lazy_k), lazy_k + 8 + 40 eager launches of rh_adam_prepare + rh_adam_lazy_sweep(RH_SWEEP_LAZY_TABLES,
    step by value): the steady state in which every window row is lazy_k steps behind.  No hipGraph, no side stream, no gate:
    under `rocprofv3 --pmc` dispatches are serialised, and a launch that spin-waits for another queue's progress (the sweep's
    gate in the replayed step) turns every step into a time-out (round 5: the passes over --trace-inner did not finish)."""
    from torch_rechub_amd import _lib, ops
    from torch_rechub_amd.optim import SWEEP_LAZY_TABLES, TableAdam
    g = torch.Generator(device=device).manual_seed(0)
    vocabs = [max(3, int(v * args.vocab_scale)) for v in CRITEO_VOCABS]
    tables = [torch.nn.Parameter(torch.randn(v, EMBED_DIM, device=device, generator=g) * 1e-2) for v in vocabs]
    opt = TableAdam(tables, table_params=tables, lr=1e-3, weight_decay=1e-5, lazy_k=args.lazy_k)
    opt.sync_hyper()
    opt._lazy_setup()
    for t in range(1, args.lazy_k + 8 + 40 + 1):
        _lib.call("rh_adam_prepare", ops._p(opt._t_hyper), ops._p(opt._t_step), ops._p(opt._t_ring), opt.RING, ops._stream())
        opt._sweep(SWEEP_LAZY_TABLES, ops._stream(), t_value=t)
    torch.cuda.synchronize()


def trace_inner(args, device, rank):
    """Child of step_accounting(): steady-state hipGraph steps and nothing else (no flush, no eager passes)."""
    wl = Workload(args, device, rank)
    model, trainer, loader = wl.build(args.tables if args.force_dp else None, True, batch=args.batch)
    trainer._graphed_step(loader)
    lazy = getattr(trainer.optimizer, "lazy_k", 0) > 1
    warm = max(args.warmup, (int(trainer.optimizer.lazy_k) + 8) if lazy else 0)
    if lazy:  # past the trainer's self-tuning of the step's form, as the headline's timed region is
        warm += trainer.tune_budget_steps() + 4
    for _ in range(warm + args.steps):
        trainer._graphed_step(loader)
    torch.cuda.synchronize()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # does not return
    # stdout carries exactly ONE line, the result JSON: RCCL / the runtime print banners on fd 1 (seen: "RCCL version :
    # ..." after the collectives), so everything else is sent to stderr for the lifetime of the process.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a "
                 "line whose n_gpus differs from what was asked for")
    if args.launch_dry_run:
        launch_dry_run(args, world, rank, result_fd)
        return
    if world > 1 and torch.cuda.device_count() < world:
        sys.exit(f"[bench] WORLD_SIZE={world} but only {torch.cuda.device_count()} HIP device(s) are visible")
    if world > 1 or args.force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    if args.force_dp:
        os.environ["RECHUB_FORCE_DP"] = "1"
    device = torch.device(f"cuda:{local}")
    torch.cuda.set_device(device)

    parallel = world > 1 or args.force_dp
    use_graph = args.graph in ("1", "auto")  # N > 1: the RCCL collectives are captured with the rest of the step
    if args.pmc_inner:
        pmc_inner(args, device)
        return
    if args.trace_inner:
        trace_inner(args, device, rank)
        return
    wl = Workload(args, device, rank)
    B = args.batch

    if not parallel:
        modes = [None]  # one GPU, one copy
    elif args.tables in ("auto", "both"):
        modes = ["replicate", "shard"]
    else:
        modes = [args.tables]
    results = {}
    for i, placement in enumerate(modes):
        ok = torch.ones(1, device=device)
        r = None
        try:
            r = run_mode(args, wl, placement, use_graph, world, rank, device, profile=(rank == 0))
        except Exception as e:  # noqa: BLE001
            if len(modes) == 1:
                raise
            ok.zero_()
            print(f"[bench] rank {rank}: tables={placement} failed ({type(e).__name__}: {e})", file=sys.stderr)
            torch.cuda.synchronize()
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() == 1 and r is not None:
            results[placement] = r
        torch.cuda.empty_cache()  # this mode's model is gone (run_mode returned): room for the next one
    if not results:
        raise RuntimeError("no table placement completed")
    best = max(results, key=lambda k: results[k]["value"])
    head = results[best]

    kernels, gsweep = {}, head.get("gsweep")
    if rank == 0:
        ms = head.get("kernel_ms", {})
        total_elems, sweep_bytes = head["total_elems"], head["sweep_bytes"]
        alg = {"rh_adam_dense": ADAM_BYTES_PER_ELEM * total_elems, "rh_adam_lazy_sweep": sweep_bytes,
               "rh_adam_lazy_step": head.get("step_bytes", sweep_bytes)}
        if wl.name in ("deepfm", "dcnv2") and best != "shard":
            alg.update({"rh_embed_fwd": FWD_BYTES_PER_SAMPLE * B, "rh_embed_bwd": BWD_BYTES_PER_SAMPLE * B,
                        "rh_batch_gather": GATHER_BYTES_PER_SAMPLE * B})
        if head.get("deferred_sweep_ms"):
            ms = dict(ms)
            ms["rh_adam_lazy_sweep"] = head["deferred_sweep_ms"]  # graph regime (replaces the eager-pass figure)
            alg["rh_adam_lazy_sweep"] = head.get("deferred_sweep_bytes", sweep_bytes)
            head.setdefault("kernel_calls_per_step", {})["rh_adam_lazy_sweep"] = 1.0
        for n, t_ms in ms.items():
            if t_ms is None:
                continue
            calls = head.get("kernel_calls_per_step", {}).get(n)
            if n not in alg:
                kernels[n] = {"avg_ms": round(t_ms, 5), "calls_per_step": calls}
                continue
            gbs = alg[n] / (t_ms * 1e-3) / 1e9
            kernels[n] = {"avg_ms": round(t_ms, 5), "calls_per_step": calls, "algorithmic_bytes": alg[n],
                          "achieved_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
    if world > 1:
        dist.barrier()

    if rank == 0:
        timed = {n: k for n, k in kernels.items() if "achieved_GBps" in k}
        dominant = max(timed, key=lambda n: timed[n]["avg_ms"] * (timed[n]["calls_per_step"] or 1)) if timed else None
        roofline = None
        if dominant:
            k = kernels[dominant]
            roofline = {"bound": "hbm", "kernel": dominant, "achieved": k["achieved_GBps"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": k["frac_of_hbm_peak"], "traffic": None,
                        "avg_launch_ms": k["avg_ms"], "algorithmic_bytes_per_launch": k["algorithmic_bytes"],
                        "regime": f"steady state, >= {head['warmup_effective']} steps since the last flush; compare with "
                                  "the kernel's average in profiles/r06_bench_kernel_stats.txt"}
            if dominant == "rh_adam_lazy_sweep" and head.get("deferred_sweep_ms"):
                roofline["regime"] = ("hipGraph-replayed steady-state steps: the deferred window sweep is launched on its "
                                      "side stream after every replay and timed there with HIP events (30 launches) WHILE the "
                                      "captured chain of the step runs beside it, i.e. under contention -- the duration "
                                      "rocprofv3 reports for adam_lazy_sweep_wide_kernel<2, 2> in profiles/r06_bench_kernel_stats.txt (steady-state launches)")
                roofline["hidden_under_the_step"] = True
                if args.vocab_scale == 1.0 and best is None and not args.brief and not args.no_pmc and not args.acct_only:
                    # re-collected now: two nested rocprofv3 --pmc passes over `bench.py --trace-inner` (pmc_traffic)
                    try:
                        pm = pmc_traffic(args)
                    except Exception as e:  # noqa: BLE001 -- the counters must never take the headline down
                        pm = {"error": f"{type(e).__name__}: {e}"}
                    print(f"[bench] pmc_traffic: {pm}", file=sys.stderr)
                    if pm.get("bytes"):
                        roofline["traffic"] = pm["bytes"]
                        roofline["traffic_source"] = ("re-collected by this run: nested rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                      "passes (kernel trace only) over `bench.py --pmc-inner` = the same sweep "
                                                      "launches in their steady state without the step beside them (under "
                                                      "counter collection dispatches are serialised; the replayed step's gate "
                                                      "would time out)")
                    roofline["traffic_pmc"] = pm
                if roofline["traffic"] is None and args.lazy_k in DEFERRED_SWEEP_PMC_TRAFFIC and args.vocab_scale == 1.0 and \
                        best is None:
                    roofline["traffic"], src = DEFERRED_SWEEP_PMC_TRAFFIC[args.lazy_k]
                    roofline["traffic_source"] = (f"profiles/{src}_{{FETCH,WRITE}}_SIZE.txt (rocprofv3 --pmc, separate "
                                                  "passes; a stored figure: the live passes were skipped or failed)")
            stored_traffic = {64: 213.1e6, 32: 421.4e6}.get(args.lazy_k)
            if dominant == "rh_adam_lazy_sweep" and stored_traffic and args.vocab_scale == 1.0 and best is None and \
                    not head.get("deferred_sweep_ms"):
                # measured with separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel in this
                # configuration (2*FETCH + WRITE, gfx950 correction, calibrated on rh_adam_dense): profiles/r01_pmc_traffic.md
                roofline["traffic"] = stored_traffic
                roofline["traffic_source"] = "profiles/r01_pmc_traffic.md (rocprofv3 --pmc, separate passes; not re-collected by bench.py)"
            if dominant == "rh_adam_lazy_step" and args.lazy_k == 64 and args.vocab_scale == 1.0 and best is None and B == 4096:
                # the merged launch under rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes over tools/pmc_probe.py,
                # 2 * FETCH + WRITE with the gfx950 correction calibrated on rh_adam_dense): 2 * 83 842 + 119 775 KiB
                roofline["traffic"] = 294.4e6
                roofline["traffic_source"] = ("profiles/r02_pmc_4096_{FETCH,WRITE}_SIZE.txt (rocprofv3 --pmc, separate passes; "
                                              "not re-collected by bench.py)")
            if dominant in ("rh_adam_lazy_sweep", "rh_adam_lazy_step"):
                # The kernel's own bound is the replay arithmetic.  In the steady state every table element advances
                # lazy_k steps per lazy_k launches, so one launch replays AT MOST total_elems element-steps (the rows the
                # batch touched are replayed by rh_adam_lazy_touched instead): an upper bound on the work, hence on frac.
                # One replay iteration = 16 packed f32 ops (4 cycles / wavefront) + 4 sqrt + 4 rcp (8 cycles each,
                # measured) + 2 VALU ops = 136 cycles per 256 element-steps on each of 1024 SIMDs at 2.4 GHz.
                # A ceiling of THIS kernel's instruction mix (the arithmetic torch.optim.Adam prescribes per element and
                # step, as the ISA issues it) -- not of the chip.  (Round 6 measured a 14-packed-op form -- the step scale
                # inside the denominator's fma -- at the same step time: the replay is bound by its dependent chain
                # sqrt -> fma -> rcp -> fma at two wavefronts per SIMD, not by the packed-op count; reverted.)
                valu_peak = 1024 * 2.4e9 / 136 * 256
                es = total_elems / (k["avg_ms"] * 1e-3)
                frac = es / valu_peak
                # PRIMARY figure (round-3 verdict): the launch is VALU-bound by design -- the HBM view moves to `hbm`
                roofline["hbm"] = {"achieved": roofline["achieved"], "peak": roofline["peak"], "unit": "GB/s",
                                   "frac": roofline["frac"],
                                   "algorithmic_bytes_per_launch": roofline["algorithmic_bytes_per_launch"],
                                   "note": "1 / lazy_k of the dense pass's bytes, by construction"}
                roofline.update({"bound": "valu", "achieved": round(es / 1e9, 1), "peak": round(valu_peak / 1e9, 1),
                                 "unit": "G element-steps/s", "frac": round(frac, 4),
                                 "element_steps_per_launch": total_elems,
                                 "peak_model": "ceiling of the kernel's OWN instruction mix, not of the chip: one replay "
                                               "iteration = 16 v_pk_*_f32 (4 cycles) + 4 v_sqrt_f32 + 4 v_rcp_f32 "
                                               "(8 cycles each, measured) + 2 VALU ops = 136 cycles per 256 element-steps, "
                                               "x 1024 SIMDs x 2.4 GHz (DESIGN 3.3); element-steps per launch = every table "
                                               "element once (an upper bound: the batch's rows are replayed by the touched "
                                               "passes)"})
                if dominant == "rh_adam_lazy_step":
                    roofline["merged"] = ("rh_adam_lazy_touched (the batch's rows, with their gradient) + rh_adam_lazy_sweep "
                                          "(window) as one launch; algorithmic bytes = window rows x (p, m, v both ways + "
                                          "gradient read + last-step word) + touched rows x (p, m, v, g both ways + index)")
                roofline["note"] = ("blocked-lazy exact Adam: this launch moves 1/K of the dense pass's bytes and replays "
                                    "the rest in registers (VALU-bound); the HBM fraction is low BY CONSTRUCTION; the "
                                    "dense pass it replaces (rh_adam_dense) runs at 64-75 % of HBM peak, see profiles/")
        north = None
        if "rh_embed_fwd" in timed and "rh_embed_bwd" in timed:
            f_, b_ = timed["rh_embed_fwd"], timed["rh_embed_bwd"]
            both_ms = f_["avg_ms"] + b_["avg_ms"]
            both_gbs = (f_["algorithmic_bytes"] + b_["algorithmic_bytes"]) / (both_ms * 1e-3) / 1e9
            north = {"kernels": ["rh_embed_fwd", "rh_embed_bwd"], "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "in_step": {"batch": B, "fwd_ms": f_["avg_ms"], "bwd_ms": b_["avg_ms"],
                                 "fwd_frac": f_["frac_of_hbm_peak"], "bwd_frac": b_["frac_of_hbm_peak"],
                                 "achieved": round(both_gbs, 1), "frac": round(both_gbs / HBM_PEAK_GBS, 4),
                                 "batch_assembly_ms": (timed.get("rh_batch_gather") or {}).get("avg_ms")},
                     "algorithmic_bytes_per_sample": {"fwd": FWD_BYTES_PER_SAMPLE, "bwd": BWD_BYTES_PER_SAMPLE}}
            if gsweep:
                big = gsweep[max(gsweep, key=int)]
                north["at_batch_%s" % max(gsweep, key=int)] = {"fwd_frac": big["rh_embed_fwd"]["frac"],
                                                              "bwd_frac": big["rh_embed_bwd"]["frac"]}
        extras = {}
        if world == 1 and not args.force_dp and not args.brief and wl.name == "deepfm" and args.batch == 4096 and \
                args.dist == "uniform" and args.vocab_scale == 1.0:
            # ---- whole-step batch sweep, Zipf(1.05) line, the other configs, whole-step accounting ----
            import copy

            def guarded(what, fn):
                try:
                    t0 = time.perf_counter()
                    out = fn()
                    print(f"[bench] {what}: {time.perf_counter() - t0:.1f} s", file=sys.stderr)
                    return out
                except Exception as e:  # noqa: BLE001 -- a secondary line must never take the headline down
                    torch.cuda.synchronize()
                    torch.cuda.empty_cache()
                    return {"error": f"{type(e).__name__}: {e}"}

            keep_args = wl.args
            sweep = {}
            for b in (() if args.acct_only else (2048, 8192, 32768, 65536)):
                sweep[str(b)] = guarded(f"batch_sweep B={b}", lambda b=b: _short_run(args, device, rank, use_graph, batch=b,
                                                                                    wl=wl)[0])
            wl.args = keep_args
            sweep["4096"] = {"batch": 4096, "ms_per_step": round(head["ms_per_step"], 4), "value": round(head["value"], 1),
                             "unit": "samples/s", "steps": args.steps, "note": "the headline run"}
            if not args.acct_only:
                extras["batch_sweep"] = {"what": "whole hipGraph-replayed training step (batch assembly, forward, backward, "
                                                 "dense-exact Adam) at other per-GPU batch sizes, same tables and dataset; 30 "
                                                 "timed steps each after the warm-up into the steady state", "runs": sweep}
                extras["zipf"] = guarded("zipf", lambda: dict(_short_run(args, device, rank, use_graph, dist_kind="zipf",
                                                                         rows=8_000_000, steps=50)[0],
                                                              index_dist="Zipf(1.05) per field (bounded power law)",
                                                              rows_per_gpu=8_000_000))
                wl2 = copy.copy(wl)
                wl2.name = "dcnv2"
                sec = {"dcnv2": guarded("dcnv2", lambda: dict(_short_run(args, device, rank, use_graph, model="dcnv2", steps=30,
                                                                         wl=wl2)[0],
                                                              workload="BASELINE.json configs[2] shape on one GPU: DCN-v2 "
                                                                       "(CrossNetMix 3 layers, rank 32, 4 experts; parallel DNN)"))}
                wl.args = keep_args
                for name in ("din", "dssm"):
                    def one(name=name):
                        r, w = _short_run(args, device, rank, use_graph, model=name, steps=20, rows=0)
                        r["workload"] = w.desc
                        del w
                        return r
                    sec[name] = guarded(name, one)
                    torch.cuda.empty_cache()
                extras["secondary_configs"] = sec
            if not args.acct_only:
                def dp_leg():
                    r = dp_one_rank(args)
                    for v in r.values():
                        if isinstance(v, dict) and v.get("ms_per_step"):
                            v["ratio_to_single"] = round(v["ms_per_step"] / head["ms_per_step"], 3)
                    r["what"] = ("`bench.py --force-dp --tables <placement>`: the full data-parallel step on a one-rank RCCL group "
                                 "(collectives launched, not skipped), 100 replayed steps; per-rank machinery cost, not a scaling "
                                 "measurement")
                    return r
                extras["dp_one_rank"] = guarded("dp_one_rank", dp_leg)
            if not args.no_step_accounting:
                extras["step_accounting"] = guarded("step_accounting", lambda: step_accounting(args))
        acct = extras.get("step_accounting") or {}
        if north and acct.get("kernels"):
            # the north-star kernels as they run INSIDE the replayed step (nested rocprofv3 trace), not the eager launches
            def in_graph(key):
                hit = [k for k in acct["kernels"] if key in k["kernel"] and "side stream" not in k["kernel"]]
                return round(sum(k["us_per_step"] for k in hit) * 1e-3, 5) if hit else None
            f_ms, b_ms, a_ms = in_graph("embed_fwd"), in_graph("embed_bwd"), in_graph("batch_gather")
            if f_ms and b_ms:
                fb = FWD_BYTES_PER_SAMPLE * B / (f_ms * 1e-3) / 1e9
                bb = BWD_BYTES_PER_SAMPLE * B / (b_ms * 1e-3) / 1e9
                both = (FWD_BYTES_PER_SAMPLE + BWD_BYTES_PER_SAMPLE) * B / ((f_ms + b_ms) * 1e-3) / 1e9
                north["in_step_eager_launches"] = north["in_step"]
                north["in_step"] = {"batch": B, "fwd_ms": f_ms, "bwd_ms": b_ms, "fwd_frac": round(fb / HBM_PEAK_GBS, 4),
                                    "bwd_frac": round(bb / HBM_PEAK_GBS, 4), "achieved": round(both, 1),
                                    "frac": round(both / HBM_PEAK_GBS, 4), "batch_assembly_ms": a_ms,
                                    "source": "in-graph kernel durations of step_accounting (nested rocprofv3 trace of the "
                                              "replayed step, the deferred sweep running beside it)"}
        cpu = None
        if world == 1 and not args.no_cpu_baseline and wl.name == "deepfm":
            from oracle.cpu_port import time_cpu_legs
            cpu = {"unit": "samples/s", "kind": "port", "cores": os.cpu_count(), "cpu_model": cpu_model_string()}
            try:
                # default run: ONE leg of SURVEY 8(d)'s 3 + 10 protocol -- the end-to-end CTRTrainer loop, which is the baseline
                # the metric names (~45 s on the pool's hosts); `--cpu-protocol full` adds the model-step-only leg (~+45 s)
                r = time_cpu_legs(wl.vocabs, N_DENSE, B, budget_s=args.cpu_budget,
                                  full={"auto": "auto", "full": True, "bounded": False}[args.cpu_protocol],
                                  model_step_leg=args.cpu_protocol == "full")
                e, m = r["end_to_end"], r["model_step"]
                cpu["cores"] = r["cores"]
                cpu["value"] = round(e["samples_per_s"], 1)
                cpu["end_to_end"] = {"value": round(e["samples_per_s"], 1), "median_ms_per_step": round(e["median_ms_per_step"], 1),
                                     "loader_median_ms_per_step": round(e["loader_median_ms_per_step"], 1),
                                     "warmup_steps": e["warmup_steps"], "timed_steps": e["timed_steps"],
                                     "step_ms": e["step_ms"], "rows": r["rows"]}
                if r.get("end_to_end_dataframe"):
                    d_ = r["end_to_end_dataframe"]
                    cpu["end_to_end_dataframe"] = {
                        "value": round(d_["samples_per_s"], 1), "median_ms_per_step": round(d_["median_ms_per_step"], 1),
                        "loader_median_ms_per_step": round(d_["loader_median_ms_per_step"], 1), "loader_ms": d_["loader_ms"],
                        "sample": "x as a pandas DataFrame (tutorial 00): TorchDataset.__getitem__ indexes 39 Series per sample "
                                  "(utils/data.py:21-22); " + d_["note"]}
                if m is not None:
                    cpu["model_step"] = {"value": round(m["samples_per_s"], 1), "median_ms_per_step": round(m["median_ms_per_step"], 1),
                                         "warmup_steps": m["warmup_steps"], "timed_steps": m["timed_steps"], "step_ms": m["step_ms"],
                                         "sample": "model-step-only (fwd+bwd+dense Adam), pre-collated batches, no DataLoader; the "
                                                   "model and Adam state are warm from the end-to-end leg"}
                cpu["sample"] = (f"{e['timed_steps']} timed steps (median) after {e['warmup_steps']} warm-up of the reference's "
                                 f"CTRTrainer.train_one_epoch loop (trainers/ctr_trainer.py:77-108) restated on eager torch "
                                 f"CPU (oracle/cpu_port.py): DataGenerator-style loader (TorchDataset + random_split "
                                 f"0.7/0.1/0.2 + DataLoader(shuffle=True, num_workers=0), x = dict of numpy, {r['rows']} "
                                 f"rows) -> DeepFM op chain -> BCELoss -> dense Adam over all 33.76M rows, B={B}")
                cpu["protocol"] = ("SURVEY 8(d): >= 3 warm-up + >= 10 timed steps, median" if r["full_protocol"] else
                                   f"TRUNCATED to about --cpu-budget = {args.cpu_budget:g} s of CPU steps (a step is ~3 s at "
                                   "the full vocabulary): 1 warm-up step, then every step timed on its own and the median "
                                   "taken (step_ms lists them); `--cpu-protocol full` runs SURVEY 8(d)'s 3 + 10 per leg (~90 s)")
            except (MemoryError, RuntimeError) as e:
                cpu["value"] = None
                cpu["error"] = f"{type(e).__name__}: {e}"
        opt_desc = "dense pass per step"
        if args.table_adam == "lazy":
            opt_desc = (f"blocked-lazy exact replay, K={head.get('lazy_k', args.lazy_k)}, window sweep "
                        + ("of step t on a side stream under step t+1's forward/backward (joined before step t+2 refreshes "
                           "its rows; residency-capped, chosen by the trainer's self-tuning)" if head["overlap_sweep"]
                           else "in line")
                        + f"; timed region = steady state ({head['warmup_effective']} untimed steps since the last flush: "
                        "the rows' lag distribution is the same at its start and end, no deferred work crosses it); the "
                        "end-of-epoch flush is timed separately (flush_ms) and verified (rows_behind_after_flush = 0)")
        steps_per_epoch = EPOCH_ROWS // (B * world)
        line = {
            "metric": "CTR train samples/sec, DeepFM Criteo-shape synthetic" if wl.name == "deepfm" else
                      f"CTR train samples/sec, {wl.name} (secondary config)",
            "value": round(head["value"], 1),
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(head["ms_per_step"], 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": wl.desc,
                "workload_id": wl.name, "rows_per_gpu": wl.rows, "batch_per_gpu": B, "global_batch": B * world,
                "index_dist": args.dist, "warmup_effective": head["warmup_effective"],
                "optimizer": "Adam lr=1e-3 weight_decay=1e-5, dense-exact semantics (every table row moves every step, as "
                             "torch.optim.Adam); execution: " + opt_desc,
                "parallelism": f"dp{world}" if parallel else "single", "hipgraph": head["hipgraph"],
                "tables": {"shard": f"row-sharded over {world} ranks (row g on rank g % {world})",
                           "replicate": "one replica per rank, gradient rows exchanged",
                           None: "single copy"}[best],
                "vocab_scale": args.vocab_scale,
                "step_form": head.get("step_form"),
            },
            "ms_per_step_steady": (head.get("steady") or {}).get("ms_per_step"),
            "steady": head.get("steady"),
            "flush_ms": head["flush_ms"],
            "rows_behind_after_flush": head.get("rows_behind_after_flush"),
            "dense_twin_check": head.get("dense_twin_check"),
            "host_enqueue_ms_per_step": round(head.get("host_enqueue_ms_per_step", 0.0), 4),
            "epoch_amortised": {"steps_per_epoch": steps_per_epoch,
                                "value": round(world * B * steps_per_epoch /
                                               (steps_per_epoch * head["ms_per_step"] * 1e-3 + head["flush_ms"] * 1e-3), 1),
                                "note": "one 45M-row epoch = steps_per_epoch steady-state steps + ONE flush"},
            "roofline": roofline,
            "roofline_north_star": north,
            "kernels": kernels,
            "gather_kernel_sweep": gsweep,
            "cpu_baseline": cpu,
        }
        line.update(extras)
        # Record hygiene: the driver's parsed record keeps the VALUES of `config` and `roofline` but only
        # the names of the other keys -- so the figures a reader needs next to `value` ride there too.
        steady_ms = (head.get("steady") or {}).get("ms_per_step")
        line["config"]["ms_per_step_steady"] = steady_ms
        line["config"]["ms_per_step_steady_note"] = ("300 replayed steps: the driver's 20-step figure also holds the last "
                                                     "step's deferred sweep behind the closing fence, once")
        dpo = extras.get("dp_one_rank") if isinstance(extras.get("dp_one_rank"), dict) else None
        if dpo:
            line["config"]["dp_one_rank_ms_per_step"] = {k: (v or {}).get("ms_per_step") for k, v in dpo.items()
                                                         if isinstance(v, dict) and "ms_per_step" in v}
            if steady_ms:
                line["config"]["dp_one_rank_ratio_to_steady_single"] = {
                    k: round(v / steady_ms, 3) for k, v in line["config"]["dp_one_rank_ms_per_step"].items() if v}
            line["config"]["dp_one_rank_note"] = ("the data-parallel step (collectives launched) on a ONE-rank RCCL group: "
                                                  "per-rank machinery cost, not a scaling measurement")
        if isinstance(line.get("roofline"), dict) and north:
            line["roofline"]["north_star"] = {
                "what": "rh_embed_fwd + rh_embed_bwd (fused gather + FM + LR and its backward) against the 8 TB/s HBM peak, "
                        "algorithmic bytes of SURVEY 8(d)",
                "in_step_frac": (north.get("in_step") or {}).get("frac"),
                "in_step_fwd_frac": (north.get("in_step") or {}).get("fwd_frac"),
                "in_step_bwd_frac": (north.get("in_step") or {}).get("bwd_frac"),
                "at_batch_65536": north.get("at_batch_65536"),
                "backward_ceiling": "profiles/r06_bwd_ceiling.txt: the cheapest write-once scatter (timing only) reaches 0.473 "
                                    "of the peak at 65536 samples; the product 0.40-0.43"}
        if kernels:
            line["kernels_timing_note"] = ("`kernels` / roofline.avg_launch_ms: HIP events around EAGER launches of the same "
                                           "step in the same regime (launch-to-launch on the stream); the in-graph durations "
                                           "are in step_accounting (nested rocprofv3) and profiles/")
        if parallel:
            line["scaling_modes"] = {
                k: {"value": round(v["value"], 1), "ms_per_step": round(v["ms_per_step"], 4), "hipgraph": v["hipgraph"],
                    "flush_ms": v["flush_ms"], "comm_us_per_step": v.get("comm_us_per_step")}
                for k, v in results.items()}
            line["scaling_modes"]["headline"] = best
            line["rccl_ranks"] = dist.get_world_size()
            line["comm_us_per_step"] = head.get("comm_us_per_step")
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

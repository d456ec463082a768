#!/usr/bin/env python
"""bench.py — CTR train samples/sec, DeepFM on Criteo-shape synthetic data, N x MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: DeepFM (tutorials/00 wiring: MLP on 13 dense + 26x16 sparse, FM + LR on the
sparse block), 26 sparse fields with the Criteo cardinalities (33 762 577 rows, D=16, fp32, 2.01 GiB of tables),
synthetic rows resident in HBM, B=4096 per GPU, the reference trainer's defaults (Adam lr 1e-3, coupled
weight_decay 1e-5, dropout 0.2) — i.e. a DENSE-exact optimizer step over every table row, as torch.optim.Adam does in
the reference.  A step = batch assembly + forward + BCE + backward + optimizer step.  Weak scaling:
per-GPU batch fixed, one process per GPU, RCCL all-reduce (dense grads) + all-gather (embedding gradient rows).

One JSON line on rank 0; `roofline` is measured live with HIP events around the kernels of the step (same stream),
`cpu_baseline` is oracle/cpu_port.py (the reference's op chain on eager torch CPU) timed on the host cores.
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
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable by a float4 copy

# algorithmic bytes: F=26, D=16, fp32, int64 indices
FWD_BYTES_PER_SAMPLE = 26 * 8 + 26 * 16 * 4 + 26 * 16 * 4 + 8  # 3544
BWD_BYTES_PER_SAMPLE = 26 * 8 + 26 * 16 * 4 + 26 * 16 * 4 + 4 + 26 * 16 * 4  # 5204
ADAM_BYTES_PER_ELEM = 28  # read p,g,m,v + write p,m,v
GATHER_BYTES_PER_SAMPLE = 2 * (26 * 8 + N_DENSE * 4 + 4)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch size")
    ap.add_argument("--rows", type=int, default=45_000_000, help="synthetic dataset rows resident per GPU")
    ap.add_argument("--graph", default="auto", choices=["auto", "0", "1"], help="replay the step from a hipGraph")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "zipf"], help="index distribution")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU baseline steps")
    ap.add_argument("--vocab-scale", type=float, default=1.0, help="debug: shrink every table")
    ap.add_argument("--table-adam", default="lazy", choices=["lazy", "dense"],
                    help="how the dense-exact Adam over the tables is executed (results are bit-identical)")
    ap.add_argument("--lazy-k", type=int, default=64)
    ap.add_argument("--tables", default="auto", choices=["auto", "replicate", "shard"],
                    help="N > 1 placement of the embedding tables; both compute the reference's global-batch update. "
                         "replicate: one replica per rank, gradient rows all-gathered (nn.DataParallel's layout); "
                         "shard: one row-shard per rank, indices all-gathered and rows reduce-scattered (table memory, "
                         "optimizer state and sweep traffic / N); auto = shard when N > 1")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the data-parallel machinery (RCCL collectives, split graphs) even on one GPU")
    return ap.parse_args()


def build_dataset(rows, vocabs, device, seed, dist_kind):
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
        from torch_rechub_amd import ops, optim
        from torch_rechub_amd.utils import data
        return [ops, optim, data]

    def remove(self):
        from torch_rechub_amd import _lib
        _lib.call = self._orig
        for mod in self._modules():
            mod._lib.call = self._orig

    def mean_ms(self):
        torch.cuda.synchronize()
        return {n: (sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else None) for n, ev in self.events.items()}


def gather_sweep(model, sparse_feas, dense_feas, vocabs, device, batches=(4096, 16384, 65536), iters=50):
    """rh_embed_fwd (gather + FM + LR + dense concat) alone at several batch sizes: avg launch time and achieved GB/s."""
    from torch_rechub_amd import ops
    out = {}
    g = torch.Generator(device=device).manual_seed(7)
    w, b = model.linear.fc.weight, model.linear.fc.bias
    with torch.no_grad():
        for B in batches:
            # the batch layout the device loader hands over: one packed (B, F) index matrix and one (B, n_dense) matrix
            idx = torch.stack([torch.randint(0, v, (B,), device=device, generator=g) for v in vocabs], 1).contiguous()
            den = torch.rand(B, len(dense_feas), device=device, generator=g)
            x = {f.name: idx[:, j] for j, f in enumerate(sparse_feas)}
            x.update({f.name: den[:, j] for j, f in enumerate(dense_feas)})
            call = model.embedding.make_call(x, sparse_feas, dense_feas, want_fm=True, want_lr=True)
            for _ in range(5):
                ops.fused_embedding(call, w, b)
            timer = KernelTimer(["rh_embed_fwd"])
            timer.install()
            for _ in range(iters):
                ops.fused_embedding(call, w, b)
            ms = timer.mean_ms()["rh_embed_fwd"]
            timer.remove()
            gbs = FWD_BYTES_PER_SAMPLE * B / (ms * 1e-3) / 1e9
            out[str(B)] = {"avg_ms": round(ms, 5), "achieved_GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
    return out


def main():
    args = parse()
    # stdout carries exactly ONE line, the result JSON: RCCL / the runtime print banners on fd 1 (seen: "RCCL version :
    # ..." after the collectives), so everything else is sent to stderr for the lifetime of the process.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    if args.force_dp:
        os.environ["RECHUB_FORCE_DP"] = "1"
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    device = torch.device(f"cuda:{local}")
    torch.cuda.set_device(device)

    from torch_rechub_amd.basic.features import DenseFeature, SparseFeature
    from torch_rechub_amd.models.ranking import DeepFM
    from torch_rechub_amd.trainers import CTRTrainer
    from torch_rechub_amd.utils.data import DeviceDataLoader

    vocabs = [max(3, int(v * args.vocab_scale)) for v in CRITEO_VOCABS]
    dense_feas = [DenseFeature(f"I{i + 1}") for i in range(N_DENSE)]
    sparse_feas = [SparseFeature(f"C{i + 1}", vocab_size=v, embed_dim=EMBED_DIM) for i, v in enumerate(vocabs)]
    parallel = world > 1 or args.force_dp
    tables = args.tables
    test_fallback = os.environ.get("RECHUB_BENCH_TEST_FALLBACK") == "1"  # exercise the fallback below on one GPU
    if tables == "auto":
        tables = "shard" if (world > 1 or (test_fallback and args.force_dp)) else "replicate"
    if not parallel:
        tables = None  # one GPU, one copy
    use_graph = args.graph in ("1", "auto")  # N > 1: the RCCL collectives are captured with the rest of the step
    sparse, dense, label = build_dataset(args.rows, vocabs, device, seed=2022 + rank, dist_kind=args.dist)

    def build(placement):
        for f in sparse_feas:  # Feature objects cache their nn.Embedding (Q2): a rebuild must start from fresh tables
            if hasattr(f, "embed"):
                del f.embed
        torch.manual_seed(2022)  # identical initial replica on every rank (and broadcast from rank 0 anyway)
        with torch.device(device):  # tables are created directly in HBM (2 GiB; never staged through the host)
            m = DeepFM(dense_feas + sparse_feas, sparse_feas, {"dims": [256, 128], "dropout": 0.2, "activation": "relu"})
        t = CTRTrainer(m, device=str(device), show_progress=False, use_graph=use_graph, table_update=args.table_adam,
                       lazy_k=args.lazy_k, tables=placement)
        ld = DeviceDataLoader(sparse, [f.name for f in sparse_feas], dense, [f.name for f in dense_feas], label,
                              args.batch, shuffle=True)
        ld.reshuffle()
        m.train()
        t.optimizer.sync_hyper()
        return m, t, ld

    model, trainer, loader = build(tables)
    if tables == "shard" and args.tables == "auto":
        # the row-sharded exchange has been validated with two ranks on one GPU and on a one-rank RCCL group, never on
        # this node's N GPUs: one eager step decides, on every rank alike, whether this job keeps it
        ok = torch.ones(1, device=device)
        try:
            trainer.train_step(*loader.load_next())
            torch.cuda.synchronize()
            if test_fallback:
                raise RuntimeError("RECHUB_BENCH_TEST_FALLBACK")
        except Exception as e:  # noqa: BLE001
            ok.zero_()
            print(f"[bench] rank {rank}: row-sharded step failed ({type(e).__name__}: {e})", file=sys.stderr)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() == 0:
            if rank == 0:
                print("[bench] falling back to replicated tables", file=sys.stderr)
            trainer.dp.close()
            del model, trainer, loader
            torch.cuda.empty_cache()
            tables = "replicate"
            model, trainer, loader = build(tables)
    B = args.batch

    graph_ok = False

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
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    trainer.flush()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    trainer.flush()  # lazy mode: every table row is brought to the last step INSIDE the timed region (weights final)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    from torch_rechub_amd import ops
    ops.check_errors(device)

    # ---- per-kernel HIP-event timing of the same step (eager launches, same stream, after the headline loop) ----
    kernels = {}
    gsweep = None
    if rank == 0:
        names = ["rh_embed_fwd", "rh_embed_bwd", "rh_adam_dense", "rh_adam_lazy_touched", "rh_adam_lazy_sweep",
                 "rh_batch_gather", "rh_embed_scatter_rows", "rh_shard_localize"]
        timer = KernelTimer(names)
        timer.install()
        n_prof = max(5, min(args.steps, 30))
        overlap = getattr(trainer.optimizer, "overlap_sweep", None)
        if overlap is not None:
            trainer.optimizer.overlap_sweep = False  # time the sweep alone, not under the forward / backward it hides behind
        for _ in range(n_prof):
            eager_step()
        ms = timer.mean_ms()
        timer.remove()
        if overlap is not None:
            trainer.optimizer.flush()
            trainer.optimizer.overlap_sweep = overlap
        total_elems = sum(p.numel() for p in trainer.optimizer._tables)
        # lazy sweep: bytes one launch must move = its 1/K window of every table (read + write p, m, v; 4 B/row of
        # `last` both ways) + the K=1 (small) tables in full incl. their gradient.  The other 1 - 1/K of the dense
        # pass's traffic is replaced by replay arithmetic, which is what bounds this kernel (VALU).
        opt = trainer.optimizer
        sweep_bytes = 0
        for p_ in opt._tables:
            rows, d_ = p_.shape
            k_ = 1 if (opt.lazy_k <= 1 or rows <= opt.lazy_small_rows) else opt.lazy_k
            win = -(-rows // k_)
            sweep_bytes += win * (d_ * 4 * (7 if k_ == 1 else 6) + 8)
        alg = {"rh_embed_fwd": FWD_BYTES_PER_SAMPLE * B, "rh_embed_bwd": BWD_BYTES_PER_SAMPLE * B,
               "rh_adam_dense": ADAM_BYTES_PER_ELEM * total_elems, "rh_adam_lazy_sweep": sweep_bytes,
               "rh_batch_gather": GATHER_BYTES_PER_SAMPLE * B}
        for n, t_ms in ms.items():
            if t_ms is None:
                continue
            if n not in alg:
                kernels[n] = {"avg_ms": round(t_ms, 5)}
                continue
            gbs = alg[n] / (t_ms * 1e-3) / 1e9
            kernels[n] = {"avg_ms": round(t_ms, 5), "algorithmic_bytes": alg[n], "achieved_GBps": round(gbs, 1),
                          "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
        # the north-star kernel over batch sizes (same tables, same stream, HIP events): its bandwidth regime starts
        # where the launch is no longer three dependent memory round trips long
        if tables != "shard":  # (a shard holds 1/N of the rows under local ids: the study belongs to the full tables)
            gsweep = gather_sweep(model, sparse_feas, dense_feas, vocabs, device)
    elif world > 1:
        for _ in range(max(5, min(args.steps, 30))):  # keep the collectives of the profiling pass matched
            eager_step()
    if world > 1:
        dist.barrier()

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = world * B * args.steps / dt
        timed = {n: k for n, k in kernels.items() if "achieved_GBps" in k}
        dominant = max(timed, key=lambda n: timed[n]["avg_ms"]) if timed else None
        roofline = None
        if dominant:
            k = kernels[dominant]
            roofline = {"bound": "hbm", "kernel": dominant, "achieved": k["achieved_GBps"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": k["frac_of_hbm_peak"], "traffic": None,
                        "avg_launch_ms": k["avg_ms"], "algorithmic_bytes_per_launch": k["algorithmic_bytes"]}
            pmc_traffic = {64: 213.1e6, 32: 421.4e6}.get(args.lazy_k)
            if dominant == "rh_adam_lazy_sweep" and pmc_traffic and args.vocab_scale == 1.0:
                # measured with separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel in this
                # configuration (2*FETCH + WRITE, gfx950 correction, calibrated on rh_adam_dense): profiles/r01_pmc_traffic.md
                roofline["traffic"] = pmc_traffic
                roofline["traffic_source"] = "profiles/r01_pmc_traffic.md (rocprofv3 --pmc, separate passes; not re-collected by bench.py)"
            if dominant == "rh_adam_lazy_sweep":
                # the kernel's own bound: one replay iteration = 16 packed f32 ops (4 cycles / wavefront) + 4 sqrt + 4 rcp
                # (8 cycles each, measured) + 2 scalar-ish VALU ops = 136 cycles per 256 element-steps on each of the
                # 1024 SIMDs at 2.4 GHz (csrc/optim.hip)
                valu_peak = 1024 * 2.4e9 / 136 * 256
                es = total_elems / (k["avg_ms"] * 1e-3)
                roofline["valu"] = {"achieved": round(es / 1e9, 1), "peak": round(valu_peak / 1e9, 1),
                                    "unit": "G element-steps/s", "frac": round(es / valu_peak, 4)}
                roofline["note"] = ("blocked-lazy exact Adam: this launch moves 1/K of the dense pass's bytes and replays "
                                    "the rest in registers (VALU-bound: %.0f M element-steps per launch, %.1f G "
                                    "element-steps/s); the dense pass it replaces is rh_adam_dense at 64-75 %% of HBM "
                                    "peak, see profiles/" % (total_elems / 1e6, total_elems / (k["avg_ms"] * 1e-3) / 1e9))
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle.cpu_port import time_cpu_baseline
            try:
                r = time_cpu_baseline(vocabs, N_DENSE, B, budget_s=args.cpu_budget)
                cpu = {"value": round(r["samples_per_s"], 1), "unit": "samples/s", "cores": r["cores"], "kind": "port",
                       "sample": f"{r['steps']} model-step-only train steps (fwd+bwd+dense Adam) of the reference op chain "
                                 f"on eager torch CPU (oracle/cpu_port.py), same DeepFM shape and vocab, B={B}, "
                                 f"{r['ms_per_step']:.0f} ms/step, pre-collated batches (no DataLoader)"}
            except (MemoryError, RuntimeError) as e:
                cpu = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port",
                       "sample": f"failed: {type(e).__name__}: {e}"}
        line = {
            "metric": "CTR train samples/sec, DeepFM Criteo-shape synthetic",
            "value": round(value, 1),
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[1]: DeepFM Criteo-shape synthetic, 26 sparse fields (33.76M rows total, "
                            "D=16) + 13 dense, MLP 429-256-128-1, fp32, dataset resident in HBM",
                "rows_per_gpu": args.rows, "batch_per_gpu": B, "global_batch": B * world, "index_dist": args.dist,
                "optimizer": "Adam lr=1e-3 weight_decay=1e-5, dense-exact semantics (every table row moves every step, as "
                             "torch.optim.Adam); execution: " + (f"blocked-lazy exact replay, K={args.lazy_k}, window sweep "
                             + ("of step t on a side stream under step t+1's forward/backward, " if getattr(
                                 trainer.optimizer, "overlap_sweep", False) else "in line, ") + "flushed "
                             "inside the timed region" if args.table_adam == "lazy" else "dense pass per step"),
                "parallelism": f"dp{world}" if (world > 1 or args.force_dp) else "single", "hipgraph": graph_ok,
                "tables": {"shard": f"row-sharded over {world} ranks (row g on rank g % {world})",
                           "replicate": "one replica per rank, gradient rows exchanged",
                           None: "single copy"}[tables],
                "vocab_scale": args.vocab_scale,
            },
            "roofline": roofline,
            "kernels": kernels,
            "gather_kernel_sweep": gsweep,
            "cpu_baseline": cpu,
        }
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Which HIP-vs-HIP comparisons of the suite are bit-exact once no fp32 sum depends on an order?

    python tools/bitwise_probe.py > gpurun_out/bitwise_probe.txt

Pairs of trainings over data in which every looked-up table row receives identical addends (collision-free batches, or
every sample twice: tests/test_gpu_models.py::_duplicate_samples), so float atomics cannot make two runs differ.  For
each pair the script prints the state_dict keys that are NOT bit-equal with their largest difference: a pair that prints
"bitwise" can be pinned with torch.equal in the suite, the others run different kernels (different summation orders) and
keep a tolerance.  Pairs: the data-parallel machinery on a one-rank RCCL group (replicated / sharded tables, eager /
single graph / split graph) against plain training; DIN / DIEN / MMOE from the HBM-resident loader under hipGraph against
eager host batches."""
import os
import socket
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def report(name, sa, sb, extra=()):
    bad = []
    for k in sa:
        a, b = sa[k], sb[k]
        if not torch.equal(a, b):
            bad.append((k, float((a.double() - b.double()).abs().max())))
    for k, a, b in extra:
        if not torch.equal(a, b):
            bad.append((k, float((a.double() - b.double()).abs().max())))
    print(f"{name}: " + ("bitwise" if not bad else f"{len(bad)} of {len(sa)} tensors differ"))
    for k, d in bad[:40]:
        print(f"    {k}: max |d| = {d:.3e}")
    sys.stdout.flush()


def dp_pairs(cases=None, layouts=("collision_free", "duplicated_samples")):
    import torch.distributed as dist
    import test_gpu_models as T
    from torch_rechub_amd import sharding
    from torch_rechub_amd.trainers import CTRTrainer
    from torch_rechub_amd.utils.data import DeviceDataLoader
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    dev = torch.device("cuda:0")
    nb, B = 12, 64
    for layout in layouts:
        vocabs, sparse, dense, label = T._loader_twin_data(layout, nb, B, seed=51)
        params = {"lr": 1e-2, "weight_decay": 1e-4, "lazy_small_rows": 64}

        def mk():
            m, dfe, sfe = T._deepfm(vocabs, 3)
            return m, [f.name for f in sfe], [f.name for f in dfe]

        ma, names, dnames = mk()
        sd0 = {k: v.clone() for k, v in ma.state_dict().items()}
        loader = lambda: DeviceDataLoader(sparse.to(dev), names, dense.to(dev), dnames, label.to(dev), B, shuffle=False)
        os.environ["RECHUB_FORCE_DP"] = "0"
        ta = CTRTrainer(ma, optimizer_params=dict(params), device="cuda:0", show_progress=False, lazy_k=4)
        la = ta.train_one_epoch(loader())
        sa = {k: v.detach().clone() for k, v in ma.state_dict().items()}
        for use_graph, tables in (cases or ((False, "replicate"), ("single", "replicate"), ("split", "replicate"), (False, "shard"),
                                            ("single", "shard"))):
            mb, _, _ = mk()
            mb.load_state_dict(sd0)
            os.environ["RECHUB_FORCE_DP"] = "1"
            os.environ["RECHUB_DP_GRAPH"] = use_graph or "single"
            tb = CTRTrainer(mb, optimizer_params=dict(params), device="cuda:0", show_progress=False, lazy_k=4,
                            use_graph=bool(use_graph), tables=tables)
            try:
                lb = tb.train_one_epoch(loader())
                sb = sharding.full_state_dict(mb) if tables == "shard" else mb.state_dict()
                sb = {k: v.detach().clone() for k, v in sb.items()}
            finally:
                tb.dp.close()
            skew = os.environ.get("PROBE_SKEW")  # "j:n": draw n extra pool streams after the j-th data-parallel trainer
            if skew:
                j, n = (int(v) for v in skew.split(":"))
                dp_pairs.count = getattr(dp_pairs, "count", 0) + 1
                if dp_pairs.count == j:
                    dp_pairs.keep = [torch.cuda.Stream() for _ in range(n)]
            print(f"[dp {layout} graph={use_graph} tables={tables}] loss plain {la!r} dp {lb!r}")
            report(f"dp one rank, {layout}, graph={use_graph}, tables={tables}", sa, sb)
            if os.environ.get("PROBE_GC") == "1":  # destroy this trainer's graphs NOW, with an idle device
                import gc
                torch.cuda.synchronize()
                del tb, mb
                gc.collect()
                torch.cuda.synchronize()
    os.environ["RECHUB_FORCE_DP"] = "0"
    dist.destroy_process_group()


def seq_pairs():
    import test_gpu_models as T
    from conftest import build_amd_model, build_mtl_model, features_from_spec, load_golden
    from torch_rechub_amd.trainers import CTRTrainer, MTLTrainer
    from torch_rechub_amd.utils.data import DeviceDataLoader
    import json
    dev = torch.device("cuda:0")
    nb, B, L = 8, 16, 7
    for cfg in ("din", "dien", "mmoe"):
        gold = load_golden(f"model_{cfg}.npz")
        models, trainers = [], []
        for graph in (False, True):
            groups = features_from_spec(gold["spec"])
            seen = {}
            for feas in groups.values():
                for f in feas:
                    if hasattr(f, "vocab_size") and id(f) not in seen:
                        f.vocab_size = 1 + 2 * B * (1 + 2 * L) + 100  # room for collision-free batches over shared tables
                        seen[id(f)] = f
            torch.manual_seed(17)
            if cfg == "mmoe":
                types = json.loads(str(gold["task_types"]))
                m = build_mtl_model("mmoe", groups, types).to(dev)
                t = MTLTrainer(m, task_types=types, optimizer_params={"lr": 1e-2, "weight_decay": 1e-4, "lazy_small_rows": 8},
                               n_epoch=1, device="cuda:0", show_progress=False, use_graph=graph, lazy_k=4)
            else:
                m = build_amd_model(cfg, groups).to(dev)
                t = CTRTrainer(m, optimizer_params={"lr": 1e-2, "weight_decay": 1e-4, "lazy_small_rows": 8}, device="cuda:0",
                               show_progress=False, use_graph=graph, lazy_k=4, loss_mode=cfg != "dien")
            models.append(m)
            trainers.append(t)
        models[1].load_state_dict(models[0].state_dict())
        # columns: one block of columns per TABLE (owner + every feature sharing it), collision-free inside a batch
        owners, order = {}, []
        dense_names = []
        for feas in groups.values():
            for f in feas:
                kind = type(f).__name__
                if kind == "DenseFeature":
                    if f.name not in dense_names:
                        dense_names.append(f.name)
                    continue
                if any(f.name == n for n, _ in order):
                    continue
                w = 1 if kind == "SparseFeature" else L
                order.append((f.name, w))
                owners.setdefault(getattr(f, "shared_with", None) or f.name, []).append((f.name, w, f))
        h = B // 2
        g = torch.Generator().manual_seed(5)
        cols = {}
        for owner, members in owners.items():
            width = sum(w for _, w, _ in members)
            v = members[0][2].vocab_size
            block = T._collision_free_columns([v], [width], nb, h, seed=hash(owner) % 1000)
            block = T._duplicate_samples(block, B)
            at = 0
            for name, w, f in members:
                c = block[:, at:at + w].clone()
                if w > 1:  # post-padded histories of 1 .. L positions (same lengths for a sample and its duplicate)
                    lens = T._duplicate_samples(torch.randint(1, L + 1, (nb * h,), generator=g), B)
                    c[torch.arange(L)[None, :] >= lens[:, None]] = 0
                cols[name] = c
                at += w
        sparse = torch.cat([cols[n] for n, _ in order], 1).contiguous()
        names = [n if w == 1 else (n, w) for n, w in order]
        dense = T._duplicate_samples(torch.rand(nb * h, len(dense_names), generator=g), B) if dense_names else None
        if cfg == "mmoe":
            label = T._duplicate_samples((torch.rand(nb * h, 2, generator=g) < 0.3).float(), B)
        else:
            label = T._duplicate_samples((torch.rand(nb * h, generator=g) < 0.3).float(), B)
        dl = DeviceDataLoader(sparse.to(dev), names, None if dense is None else dense.to(dev), dense_names, label.to(dev), B,
                              shuffle=False)
        la = trainers[0].train_one_epoch(T._host_batches(sparse, names, dense, dense_names, label, B))
        lb = trainers[1].train_one_epoch(dl)
        print(f"[{cfg}] loss host {la!r} graph {lb!r} graph captured {trainers[1]._graph is not None}")
        oa, ob = trainers[0].optimizer, trainers[1].optimizer
        extra = [(f"exp_avg_sq[{i}]", oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"])
                 for i, (pa, pb) in enumerate(zip(oa._tables, ob._tables))]
        report(f"{cfg}: host eager vs device loader + hipGraph, duplicated samples over collision-free rows",
               models[0].state_dict(), models[1].state_dict(), extra)


if __name__ == "__main__":
    import faulthandler
    faulthandler.enable()
    which = sys.argv[1:] or ["dp", "seq"]
    for w in which:
        try:
            {"dp": dp_pairs, "seq": seq_pairs,
             # the captured row-sharded step twice in one process (round 5: the second capture of the full probe segfaulted)
             "shard2": lambda: dp_pairs(cases=(("single", "shard"), ("single", "shard")), layouts=("duplicated_samples",)),
             "shard2x": lambda: dp_pairs(cases=(("single", "shard"),)),
             "seqA": lambda: dp_pairs(cases=((False, "shard"), ("single", "shard")) * 2, layouts=("duplicated_samples",)),
             "seqB": lambda: dp_pairs(cases=(("split", "replicate"), (False, "shard"), ("single", "shard")) * 2,
                                      layouts=("duplicated_samples",)),
             "seqC": lambda: dp_pairs(cases=((False, "replicate"), ("single", "replicate"), (False, "shard"), ("single", "shard")) * 2,
                                      layouts=("duplicated_samples",)),
             "seqD": lambda: dp_pairs(cases=((False, "replicate"), ("single", "replicate"), ("split", "replicate"), (False, "shard"),
                                             ("single", "shard")) * 2, layouts=("duplicated_samples",)),
             "cfcf": lambda: dp_pairs(layouts=("collision_free", "collision_free")),
             "dupdup": lambda: dp_pairs(layouts=("duplicated_samples", "duplicated_samples")),
             "dupcf": lambda: dp_pairs(layouts=("duplicated_samples", "collision_free")),
             "single8": lambda: dp_pairs(cases=(("single", "replicate"),) * 8, layouts=("duplicated_samples",)),
             "mix8": lambda: dp_pairs(cases=(("single", "replicate"), ("split", "replicate"), ("single", "shard")) * 3,
                                      layouts=("duplicated_samples",))}[w]()
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            print(f"{w}: FAILED {type(e).__name__}: {e}")

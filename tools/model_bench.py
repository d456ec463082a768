#!/usr/bin/env python
"""Train-step throughput of the secondary BASELINE.json configs on ONE GPU (eager + hipGraph):
  dcn / dcnv2 : configs[2] shape (Criteo tables, 13 dense + 26 sparse, 3 cross layers)          [C3 runs it data-parallel]
  din         : configs[3] shape (2 history fields x L=100 + 2 targets + user_id, D=16, attention MLP [256,128] Dice)
  dien / bst  : the same tables and histories through DIEN (GRU + AUGRU recurrences, auxiliary loss) / BST (1 encoder layer)
  dssm        : configs[4] shape (100 M-item + 10 M-user tables, history L=50 mean-pooled, towers [256,128,64] prelu,
                in-batch negatives)
    python tools/model_bench.py --models dcn,dcnv2,din,dssm --steps 30
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import CRITEO_VOCABS  # noqa: E402


def build(name, dev, B, scale):
    from torch_rechub_amd.basic.features import DenseFeature, SequenceFeature, SparseFeature
    from torch_rechub_amd.models.matching import DSSM
    from torch_rechub_amd.models.ranking import DCN, DIN, DCNv2
    from torch_rechub_amd.trainers import CTRTrainer, MatchTrainer
    g = torch.Generator(device=dev).manual_seed(0)
    mlp = {"dims": [256, 128], "dropout": 0.2, "activation": "relu"}
    if name in ("dcn", "dcnv2"):
        vocabs = [max(3, int(v * scale)) for v in CRITEO_VOCABS]
        dense = [DenseFeature(f"I{i}") for i in range(13)]
        sparse = [SparseFeature(f"C{i}", v, 16) for i, v in enumerate(vocabs)]
        with torch.device(dev):
            model = DCN(dense + sparse, 3, {"dims": [256, 128]}) if name == "dcn" else DCNv2(dense + sparse, 3, mlp)
        x = {f.name: torch.randint(0, v, (B,), device=dev, generator=g) for f, v in zip(sparse, vocabs)}
        x.update({f.name: torch.rand(B, device=dev, generator=g) for f in dense})
        trainer = CTRTrainer(model, device=str(dev), show_progress=False)
    elif name in ("din", "dien", "bst"):
        nu, ni, nc, L = int(200000 * scale) + 10, int(63001 * scale) + 10, 801, 100
        feats = [SparseFeature("user_id", nu, 16)]
        hist = [SequenceFeature("hist_item", ni, 16, pooling="concat", shared_with="target_item", padding_idx=0),
                SequenceFeature("hist_cate", nc, 16, pooling="concat", shared_with="target_cate", padding_idx=0)]
        tgt = [SparseFeature("target_item", ni, 16, padding_idx=0), SparseFeature("target_cate", nc, 16, padding_idx=0)]
        with torch.device(dev):
            if name == "din":
                model = DIN(feats, hist, tgt, mlp_params={"dims": [256, 128], "dropout": 0.2},
                            attention_mlp_params={"dims": [256, 128]})
            elif name == "dien":  # same tables and history shape; negative histories for the auxiliary loss
                from torch_rechub_amd.models.ranking import DIEN
                neg = [SequenceFeature("neg_hist_item", ni, 16, pooling="concat", shared_with="target_item", padding_idx=0),
                       SequenceFeature("neg_hist_cate", nc, 16, pooling="concat", shared_with="target_cate", padding_idx=0)]
                model = DIEN(feats, hist, neg, tgt, mlp_params={"dims": [256, 128], "dropout": 0.2})
            else:
                from torch_rechub_amd.models.ranking import BST
                model = BST(feats, hist, tgt, mlp_params=mlp, nhead=4, dropout=0.2, num_layers=1, max_seq_len=L + 1)
        lens = torch.randint(1, L + 1, (B,), device=dev, generator=g)
        pad = torch.arange(L, device=dev)[None, :] >= lens[:, None]
        x = {"user_id": torch.randint(0, nu, (B,), device=dev, generator=g),
             "target_item": torch.randint(1, ni, (B,), device=dev, generator=g),
             "target_cate": torch.randint(1, nc, (B,), device=dev, generator=g),
             "hist_item": torch.randint(1, ni, (B, L), device=dev, generator=g).masked_fill(pad, 0),
             "hist_cate": torch.randint(1, nc, (B, L), device=dev, generator=g).masked_fill(pad, 0)}
        if name == "dien":
            x["neg_hist_item"] = torch.randint(1, ni, (B, L), device=dev, generator=g).masked_fill(pad, 0)
            x["neg_hist_cate"] = torch.randint(1, nc, (B, L), device=dev, generator=g).masked_fill(pad, 0)
        trainer = CTRTrainer(model, device=str(dev), show_progress=False, loss_mode=name != "dien")
    elif name == "dssm":
        nu, ni, L = int(10_000_000 * scale) + 10, int(100_000_000 * scale) + 10, 50
        user = [SparseFeature("user_id", nu, 16),
                SequenceFeature("hist_item", ni, 16, pooling="mean", shared_with="item_id", padding_idx=0)]
        item = [SparseFeature("item_id", ni, 16, padding_idx=0), SparseFeature("cate_id", 1000, 16)]
        tower = {"dims": [256, 128, 64], "activation": "prelu"}
        with torch.device(dev):
            model = DSSM(user, item, user_params=dict(tower), item_params=dict(tower), temperature=0.02)
        lens = torch.randint(1, L + 1, (B,), device=dev, generator=g)
        pad = torch.arange(L, device=dev)[None, :] >= lens[:, None]
        x = {"user_id": torch.randint(0, nu, (B,), device=dev, generator=g),
             "item_id": torch.randint(1, ni, (B,), device=dev, generator=g),
             "cate_id": torch.randint(0, 1000, (B,), device=dev, generator=g),
             "hist_item": torch.randint(1, ni, (B, L), device=dev, generator=g).masked_fill(pad, 0)}
        trainer = MatchTrainer(model, mode=0, in_batch_neg=True, in_batch_neg_ratio=20, device=str(dev),
                               show_progress=False)
    else:
        raise ValueError(name)
    y = (torch.rand(B, device=dev, generator=g) < 0.25).float()
    return trainer, x, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="dcn,dcnv2,din,dssm")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--scale", type=float, default=1.0, help="vocabulary scale")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for name in a.models.split(","):
        torch.cuda.empty_cache()
        trainer, x, y = build(name, dev, a.batch, a.scale)
        trainer.model.train()
        trainer.optimizer.sync_hyper()
        print(f"  [{name}] built", flush=True)
        for i in range(5):
            trainer.train_step(x, y)
            if os.environ.get("MB_SYNC"):
                torch.cuda.synchronize()
                print(f"  [{name}] warm step {i} ok", flush=True)
        torch.cuda.synchronize()
        print(f"  [{name}] warm-up done", flush=True)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            trainer.train_step(x, y)
        trainer.flush()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / a.steps
        print(f"  [{name}] eager done {eager * 1e3:.3f} ms/step", flush=True)
        # hipGraph replay of the same step on the same static batch
        graph_ms = float("nan")
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    trainer.train_step(x, y)
            torch.cuda.current_stream().wait_stream(s)
            from torch_rechub_amd.graphs import SegmentedGraph
            gph = SegmentedGraph()  # lets the optimizer cut the step where it launches its side-stream sweep
            gph.capture(lambda: trainer.train_step(x, y))
            print(f"  [{name}] captured ({len(gph.segments)} segments)", flush=True)
            for _ in range(3):
                gph.replay()
            torch.cuda.synchronize()
            print(f"  [{name}] replays ok", flush=True)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                gph.replay()
            trainer.flush()
            torch.cuda.synchronize()
            graph_ms = (time.perf_counter() - t0) / a.steps * 1e3
        except Exception as e:  # noqa: BLE001
            print(f"  [{name}] graph capture failed: {type(e).__name__}: {e}")
            torch.cuda.synchronize()
        mem = torch.cuda.max_memory_allocated() / 2**30
        print(f"{name:6s} B={a.batch} eager {eager * 1e3:8.3f} ms/step ({a.batch / eager / 1e3:9.1f} k samples/s)   "
              f"hipGraph {graph_ms:8.3f} ms/step ({a.batch / graph_ms:9.1f} k samples/s)   peak mem {mem:.1f} GiB",
              flush=True)
        from torch_rechub_amd import ops
        ops.check_errors()
        del trainer, x, y
        torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main()

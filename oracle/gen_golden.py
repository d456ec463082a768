"""Generate tests/golden/*.npz from the UNMODIFIED reference (datawhalechina/torch-rechub v0.8.0, CPU).

Run in the build container only (needs /root/reference):  ``python oracle/gen_golden.py``
The fixtures pin (a) the numpy oracle and (b) the HIP path on the GPU box, where the reference is absent.

Files
  layers.npz          one group of arrays per hot-path layer: inputs, parameters, outputs, input/param gradients
  model_<cfg>.npz     per model config: feature spec (json), batch, initial state_dict, train/eval predictions,
                      BCE loss, all parameter gradients, and the state_dict + mean loss after the reference
                      ``CTRTrainer.train_one_epoch`` ran over three fixed batches (Adam, coupled weight decay)
Everything is seeded with 2022 (the reference's conventional seed, examples/ranking/run_criteo.py:99).
Dropout is 0 in the fixtures (masks are RNG-dependent); BatchNorm runs in train mode for the gradient cases.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_import import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SEED = 2022


def npy(t):
    return t.detach().cpu().numpy().copy()


def spec_of(fea):
    kind = type(fea).__name__
    d = {"kind": kind, "name": fea.name, "embed_dim": fea.embed_dim}
    if kind != "DenseFeature":
        d.update(vocab_size=fea.vocab_size, shared_with=fea.shared_with, padding_idx=fea.padding_idx)
    if kind == "SequenceFeature":
        d["pooling"] = fea.pooling
    return d


def gen_layers(rh):
    from torch_rechub.basic.activation import Dice
    from torch_rechub.basic.features import DenseFeature, SequenceFeature, SparseFeature
    from torch_rechub.basic.layers import (FM, LR, MLP, CrossNetMix, CrossNetV2, CrossNetwork, EmbeddingLayer)
    from torch_rechub.models.ranking.din import ActivationUnit
    torch.manual_seed(SEED)
    g = torch.Generator().manual_seed(SEED)
    out = {}
    B, F, D, ND = 37, 6, 16, 3

    # --- EmbeddingLayer: sparse (+shared table, +padding_idx) + dense, both layouts -------------------------
    vocabs = [3, 11, 50, 7, 200, 11]
    feas = [SparseFeature(f"s{i}", vocab_size=v, embed_dim=D) for i, v in enumerate(vocabs[:5])]
    feas[3] = SparseFeature("s3", vocab_size=7, embed_dim=D, padding_idx=0)
    feas.append(SparseFeature("s5", vocab_size=11, embed_dim=D, shared_with="s1"))
    dense = [DenseFeature(f"d{i}") for i in range(ND)]
    layer = EmbeddingLayer(dense + feas)
    for k, m in layer.embed_dict.items():
        torch.nn.init.normal_(m.weight, 0, 0.5, generator=g)
        if m.padding_idx is not None:
            with torch.no_grad():
                m.weight[m.padding_idx].zero_()
    x = {f.name: torch.randint(0, v, (B,), generator=g) for f, v in zip(feas, vocabs)}
    x.update({f.name: torch.rand(B, generator=g) for f in dense})
    sq = layer(x, dense + feas, squeeze_dim=True)
    ns = layer(x, feas, squeeze_dim=False)
    gsq = torch.randn(sq.shape, generator=g)
    sq.backward(gsq)
    out["emb.idx"] = np.stack([npy(x[f.name]) for f in feas], 1)
    out["emb.dense"] = np.stack([npy(x[f.name]) for f in dense], 1)
    out["emb.vocabs"] = np.array(vocabs)
    for k, m in layer.embed_dict.items():
        out[f"emb.table.{k}"] = npy(m.weight)
        out[f"emb.grad.{k}"] = npy(m.weight.grad)
    out["emb.out_squeeze"] = npy(sq)
    out["emb.out_3d"] = npy(ns)
    out["emb.g_squeeze"] = npy(gsq)

    # --- sequence features: sum / mean / concat, padding_idx set and unset ------------------------------------
    L = 9
    for tag, pad in (("pad0", 0), ("nopad", None)):
        for pooling in ("sum", "mean", "concat"):
            fea = SequenceFeature(f"h_{pooling}_{tag}", vocab_size=40, embed_dim=D, pooling=pooling, padding_idx=pad)
            lay = EmbeddingLayer([fea])
            torch.nn.init.normal_(lay.embed_dict[fea.name].weight, 0, 0.5, generator=g)
            if pad is not None:
                with torch.no_grad():
                    lay.embed_dict[fea.name].weight[pad].zero_()
            idx = torch.randint(1, 40, (B, L), generator=g)
            lens = torch.randint(1, L + 1, (B,), generator=g)
            idx[torch.arange(L)[None, :] >= lens[:, None]] = 0  # post-padding with 0 (utils/data.py:176)
            y = lay({fea.name: idx}, [fea], squeeze_dim=False)
            gy = torch.randn(y.shape, generator=g)
            y.backward(gy)
            key = f"seq.{pooling}.{tag}"
            out[key + ".idx"] = npy(idx)
            out[key + ".table"] = npy(lay.embed_dict[fea.name].weight)
            out[key + ".out"] = npy(y)
            out[key + ".g"] = npy(gy)
            out[key + ".grad"] = npy(lay.embed_dict[fea.name].weight.grad)

    # --- FM / LR ---------------------------------------------------------------------------------------------
    xf = torch.randn(B, F, D, generator=g, requires_grad=True)
    for rs in (True, False):
        y = FM(reduce_sum=rs)(xf)
        gy = torch.randn(y.shape, generator=g)
        (gx,) = torch.autograd.grad(y, xf, gy)
        out[f"fm.{int(rs)}.out"], out[f"fm.{int(rs)}.g"], out[f"fm.{int(rs)}.gx"] = npy(y), npy(gy), npy(gx)
    out["fm.x"] = npy(xf)
    lr = LR(F * D)
    y = lr(xf.flatten(1))
    out["lr.w"], out["lr.b"], out["lr.out"] = npy(lr.fc.weight), npy(lr.fc.bias), npy(y)

    # --- cross networks ---------------------------------------------------------------------------------------
    d = 45
    xc = torch.randn(B, d, generator=g, requires_grad=True)
    for nl in (1, 3, 6):
        cn = CrossNetwork(d, nl)
        for w in cn.w:
            torch.nn.init.normal_(w.weight, 0, 0.3, generator=g)
        for b in cn.b:
            torch.nn.init.normal_(b, 0, 0.3, generator=g)
        y = cn(xc)
        gy = torch.randn(y.shape, generator=g)
        grads = torch.autograd.grad(y, [xc] + [w.weight for w in cn.w] + list(cn.b), gy)
        k = f"cross.{nl}"
        out[k + ".W"] = np.concatenate([npy(w.weight) for w in cn.w], 0)
        out[k + ".B"] = np.stack([npy(b) for b in cn.b], 0)
        out[k + ".out"], out[k + ".g"], out[k + ".gx"] = npy(y), npy(gy), npy(grads[0])
        out[k + ".gW"] = np.concatenate([npy(t) for t in grads[1:1 + nl]], 0)
        out[k + ".gB"] = np.stack([npy(t) for t in grads[1 + nl:]], 0)
    out["cross.x"] = npy(xc)
    v2 = CrossNetV2(d, 2)
    for b in v2.b:
        torch.nn.init.normal_(b, 0, 0.3, generator=g)
    out["crossv2.W"] = np.stack([npy(w.weight) for w in v2.w], 0)
    out["crossv2.B"] = np.stack([npy(b) for b in v2.b], 0)
    out["crossv2.out"] = npy(v2(xc))
    mix = CrossNetMix(d, num_layers=2, low_rank=8, num_experts=3)
    for b in mix.bias:
        torch.nn.init.normal_(b, 0, 0.3, generator=g)
    out["crossmix.U"] = np.stack([npy(t) for t in mix.u_list], 0)
    out["crossmix.V"] = np.stack([npy(t) for t in mix.v_list], 0)
    out["crossmix.C"] = np.stack([npy(t) for t in mix.c_list], 0)
    out["crossmix.Wg"] = np.concatenate([npy(t.weight) for t in mix.gating], 0)
    out["crossmix.bias"] = np.stack([npy(t).reshape(-1) for t in mix.bias], 0)
    out["crossmix.out"] = npy(mix(xc))

    # --- Dice / ActivationUnit -----------------------------------------------------------------------------------
    xd = torch.randn(B, 20, generator=g)
    dice = Dice()
    out["dice.x"], out["dice.alpha"], out["dice.out"] = npy(xd), npy(dice.alpha), npy(dice(xd))
    for sm in (False, True):
        au = ActivationUnit(D, dims=[12, 6], activation="dice", use_softmax=sm)
        au.eval()  # BatchNorm with running stats (0/1): isolates the attention arithmetic
        hist = torch.randn(B, L, D, generator=g)
        tgt = torch.randn(B, D, generator=g)
        k = f"au.{int(sm)}"
        out[k + ".hist"], out[k + ".tgt"], out[k + ".out"] = npy(hist), npy(tgt), npy(au(hist, tgt))
        for n, t in au.state_dict().items():
            out[k + ".sd." + n] = npy(t)

    # --- torch.optim.Adam trajectory on a small tensor (wd coupled) ------------------------------------------------
    p = torch.nn.Parameter(torch.randn(5, 8, generator=g))
    out["adam.p0"] = npy(p)
    opt = torch.optim.Adam([p], lr=1e-2, weight_decay=1e-3)
    gs = []
    for t in range(4):
        gr = torch.randn(5, 8, generator=g)
        if t == 2:
            gr.zero_()  # a step with no data gradient still moves p (Q9)
        gs.append(npy(gr))
        p.grad = gr
        opt.step()
        out[f"adam.p{t + 1}"] = npy(p)
    out["adam.g"] = np.stack(gs, 0)
    out["adam.m"] = npy(opt.state[p]["exp_avg"])
    out["adam.v"] = npy(opt.state[p]["exp_avg_sq"])
    np.savez_compressed(os.path.join(OUT, "layers.npz"), **out)
    print("layers.npz", len(out), "arrays")


# attention MLP widths per DIN fixture.  "din_wide" is the reference's own configuration
# (examples/ranking/run_amazon_electronics.py:57: attention_mlp_params={"dims": [256, 128]}), the shape whose first layer
# runs on csrc/dinmlp.hip (register-built operand, BatchNorm statistics in the epilogue, tile-GEMM input gradient);
# "din_wide64" is the narrowest width that path accepts, with a single hidden layer.
DIN_ATTENTION_DIMS = {"din_wide": [256, 128], "din_wide64": [64], "din_wide_softmax": [128, 64]}


def build_model(rh, cfg):
    from torch_rechub.basic.features import DenseFeature, SequenceFeature, SparseFeature
    from torch_rechub.models.ranking import DCN, DIN, DCNv2, DeepFM, WideDeep
    D = 16
    if cfg == "dssm":  # config 5: two-tower matching, mean-pooled history shares the item table
        from torch_rechub.models.matching import DSSM
        user = [SparseFeature("user_id", vocab_size=40, embed_dim=D),
                SequenceFeature("hist_item", vocab_size=60, embed_dim=D, pooling="mean", shared_with="item_id",
                                padding_idx=0)]
        item = [SparseFeature("item_id", vocab_size=60, embed_dim=D, padding_idx=0),
                SparseFeature("cate_id", vocab_size=12, embed_dim=D)]
        tower = {"dims": [32, 16], "activation": "prelu"}
        return DSSM(user, item, user_params=dict(tower), item_params=dict(tower), temperature=0.02), \
            {"user_features": user, "item_features": item}
    mlp = {"dims": [32, 16], "dropout": 0.0, "activation": "relu"}
    if cfg.startswith("din"):
        feats = [SparseFeature("user_id", vocab_size=30, embed_dim=D)]
        hist = [SequenceFeature("hist_item", vocab_size=50, embed_dim=D, pooling="concat", shared_with="target_item",
                                padding_idx=0),
                SequenceFeature("hist_cate", vocab_size=12, embed_dim=D, pooling="concat", shared_with="target_cate",
                                padding_idx=0)]
        tgt = [SparseFeature("target_item", vocab_size=50, embed_dim=D, padding_idx=0),
               SparseFeature("target_cate", vocab_size=12, embed_dim=D, padding_idx=0)]
        model = DIN(feats, hist, tgt, mlp_params={"dims": [32, 16], "dropout": 0.0},
                    attention_mlp_params={"dims": DIN_ATTENTION_DIMS.get(cfg, [16, 8]),
                                          "use_softmax": cfg.endswith("softmax")})
        return model, {"features": feats, "history_features": hist, "target_features": tgt}
    if cfg in ("bst", "dien"):  # sequence models over the same (history, target) feature pairs as DIN
        feats = [SparseFeature("user_id", vocab_size=30, embed_dim=D)]
        tgt = [SparseFeature("target_item", vocab_size=50, embed_dim=D, padding_idx=0),
               SparseFeature("target_cate", vocab_size=12, embed_dim=D, padding_idx=0)]

        def sequences(prefix):
            return [SequenceFeature(prefix + "item", vocab_size=50, embed_dim=D, pooling="concat",
                                    shared_with="target_item", padding_idx=0),
                    SequenceFeature(prefix + "cate", vocab_size=12, embed_dim=D, pooling="concat",
                                    shared_with="target_cate", padding_idx=0)]

        hist = sequences("hist_")
        if cfg == "bst":
            from torch_rechub.models.ranking import BST
            model = BST(feats, hist, tgt, mlp_params=mlp, nhead=2, dropout=0.0, num_layers=1, max_seq_len=8)
            return model, {"features": feats, "history_features": hist, "target_features": tgt}
        from torch_rechub.models.ranking import DIEN
        neg = sequences("neg_hist_")
        model = DIEN(feats, hist, neg, tgt, mlp_params={"dims": [32, 16], "dropout": 0.0}, alpha=0.2)
        return model, {"features": feats, "history_features": hist, "neg_history_features": neg, "target_features": tgt}
    dense = [DenseFeature(f"I{i + 1}") for i in range(13)]
    vocabs = [3, 4, 10, 27, 105, 305, 583, 40, 1460, 24, 18, 15, 633] * 2
    if cfg in ("dcnv2_full_stacked", "fibinet", "fibinet_each"):
        vocabs = vocabs[:6]  # keeps the (d, d) cross weights / the F (F - 1) D wide MLP input of the fixture small
    sparse = [SparseFeature(f"C{i + 1}", vocab_size=v, embed_dim=D) for i, v in enumerate(vocabs)]
    if cfg == "deepfm_tutorial":  # tutorials/00: deep = dense + sparse, fm = sparse
        return DeepFM(dense + sparse, sparse, mlp), {"deep_features": dense + sparse, "fm_features": sparse}
    if cfg == "deepfm_criteo":  # examples/ranking/run_criteo.py:66: deep = dense only
        return DeepFM(dense, sparse, mlp), {"deep_features": dense, "fm_features": sparse}
    if cfg == "widedeep":
        return WideDeep(dense, sparse, mlp), {"wide_features": dense, "deep_features": sparse}
    if cfg == "dcn":
        return DCN(dense + sparse, 3, {"dims": [32, 16]}), {"features": dense + sparse}
    if cfg == "dcnv2_mix":
        return DCNv2(dense + sparse, 3, mlp, low_rank=8, num_experts=3), {"features": dense + sparse}
    if cfg == "dcnv2_full_stacked":
        return DCNv2(dense + sparse, 2, mlp, model_structure="stacked", use_low_rank_mixture=False), \
            {"features": dense + sparse}
    if cfg == "afm":
        from torch_rechub.models.ranking import AFM
        return AFM(sparse, D, t=8), {"fm_features": sparse}
    if cfg in ("edcn", "edcn_attention"):
        from torch_rechub.models.ranking import EDCN
        kind = "hadamard_product" if cfg == "edcn" else "attention_pooling"
        return EDCN(sparse[:6], 2, {"dropout": 0.0, "activation": "relu"}, bridge_type=kind), {"features": sparse[:6]}
    if cfg == "autoint":
        from torch_rechub.models.ranking import AutoInt
        return AutoInt(sparse[:9], dense[:4], num_layers=2, num_heads=2, dropout=0.0, mlp_params=mlp), \
            {"sparse_features": sparse[:9], "dense_features": dense[:4]}
    if cfg in ("fibinet", "fibinet_each"):
        from torch_rechub.models.ranking import FiBiNet
        kind = "field_interaction" if cfg == "fibinet" else "field_each"
        return FiBiNet(sparse, mlp, reduction_ratio=3, bilinear_type=kind), {"features": sparse}
    raise ValueError(cfg)


def make_batch(groups, B, g, L=7):
    x = {}
    seen = set()
    for feas in groups.values():
        for f in feas:
            if f.name in seen:
                continue
            seen.add(f.name)
            kind = type(f).__name__
            if kind == "DenseFeature":
                x[f.name] = torch.rand(B, generator=g)
            elif kind == "SparseFeature":
                lo = 1 if f.padding_idx == 0 else 0
                x[f.name] = torch.randint(lo, f.vocab_size, (B,), generator=g)
            else:
                idx = torch.randint(1, f.vocab_size, (B, L), generator=g)
                lens = torch.randint(1, L + 1, (B,), generator=g)
                idx[torch.arange(L)[None, :] >= lens[:, None]] = 0
                x[f.name] = idx
    y = (torch.rand(B, generator=g) < 0.25).long()
    return x, y


def gen_model(rh, cfg):
    from torch_rechub.trainers import CTRTrainer
    torch.manual_seed(SEED)
    g = torch.Generator().manual_seed(SEED + 1)
    model, groups = build_model(rh, cfg)
    # larger-than-default table init so that FM / attention terms are numerically visible
    for m in model.modules():
        if isinstance(m, torch.nn.Embedding):
            torch.nn.init.normal_(m.weight, 0, 0.1, generator=g)
            if m.padding_idx is not None:
                with torch.no_grad():
                    m.weight[m.padding_idx].zero_()
    B = 48
    batches = [make_batch(groups, B, g) for _ in range(3)]
    out = {"spec": np.array(json.dumps({k: [spec_of(f) for f in v] for k, v in groups.items()})),
           "cfg": np.array(cfg)}
    for n, t in model.state_dict().items():
        out["sd0." + n] = npy(t)
    x, y = batches[0]
    for bi, (bx, by) in enumerate(batches):
        for k, v in bx.items():
            out[f"x{bi}.{k}"] = npy(v)
        out[f"y{bi}"] = npy(by)
    with_aux = cfg == "dien"  # forward returns (prediction, weighted auxiliary loss): CTRTrainer(loss_mode=False)
    model.eval()
    with torch.no_grad():
        out["pred_eval"] = npy(model(x)[0] if with_aux else model(x))
    model.train()
    sd_backup = {k: v.clone() for k, v in model.state_dict().items()}
    pred = model(x)
    if with_aux:
        pred, aux = pred
        out["aux_train"] = np.array(aux.item())
        loss = torch.nn.BCELoss()(pred, y.float()) + aux
    else:
        loss = torch.nn.BCELoss()(pred, y.float())
    model.zero_grad()
    loss.backward()
    out["pred_train"] = npy(pred)
    out["loss"] = np.array(loss.item())
    for n, p in model.named_parameters():
        out["grad." + n] = npy(p.grad) if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
    model.load_state_dict(sd_backup)  # undo the BatchNorm running-stat update of the probe forward
    model.zero_grad()
    # the reference training loop itself: trainers/ctr_trainer.py:77-108 over three fixed batches
    wd = 1e-3
    if cfg == "dssm":  # trainers/match_trainer.py:105-176, in-batch HARD negatives (deterministic: top-k, no RNG)
        from torch_rechub.trainers import MatchTrainer
        trainer = MatchTrainer(model, mode=0, in_batch_neg=True, in_batch_neg_ratio=3, hard_negative=True,
                               optimizer_params={"lr": 1e-2, "weight_decay": wd}, n_epoch=1, device="cpu")
        with torch.no_grad():
            model.eval()
            out["user_emb"], out["item_emb"] = npy(model.user_tower(x)), npy(model.item_tower(x))
            model.train()
    else:
        trainer = CTRTrainer(model, optimizer_params={"lr": 1e-2, "weight_decay": wd}, n_epoch=1, device="cpu",
                             loss_mode=not with_aux)
    mean_loss = trainer.train_one_epoch(batches)
    out["train.lr"], out["train.wd"], out["train.mean_loss"] = np.array(1e-2), np.array(wd), np.array(mean_loss)
    for n, t in model.state_dict().items():
        out["sd3." + n] = npy(t)
    np.savez_compressed(os.path.join(OUT, f"model_{cfg}.npz"), **out)
    print(f"model_{cfg}.npz", len(out), "arrays, loss", loss.item(), "mean train loss", mean_loss)


def gen_augru(rh):
    """AUGRU of the UNMODIFIED reference (models/ranking/dien.py:38-66) on a ragged batch: states of every step, the
    final state, and the gradients of a random linear functional of them with respect to inputs and parameters."""
    from torch_rechub.models.ranking.dien import AUGRU
    torch.manual_seed(SEED + 7)
    g = torch.Generator().manual_seed(SEED + 8)
    out = {}
    for D, B, T in ((16, 37, 9), (8, 5, 4)):
        net = AUGRU(D)
        for p in net.parameters():
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=g) * 0.4)
        x = (torch.randn(B, T, D, generator=g) * 0.8).requires_grad_(True)
        item = torch.randn(B, D, generator=g).requires_grad_(True)
        lens = torch.randint(1, T + 1, (B,), generator=g)
        mask = torch.arange(T)[None, :] < lens[:, None]
        outs, h = net(x, item, mask)
        G, Gh = torch.randn(B, T, D, generator=g), torch.randn(B, D, generator=g)
        ((outs * G).sum() + (h * Gh).sum()).backward()
        tag = f"d{D}."
        out[tag + "x"], out[tag + "item"], out[tag + "mask"] = npy(x), npy(item), npy(mask)
        out[tag + "outs"], out[tag + "h"], out[tag + "G"], out[tag + "Gh"] = npy(outs), npy(h), npy(G), npy(Gh)
        out[tag + "gx"], out[tag + "gitem"] = npy(x.grad), npy(item.grad)
        for n, p in net.named_parameters():
            out[tag + "p." + n], out[tag + "g." + n] = npy(p), npy(p.grad)
    np.savez_compressed(os.path.join(OUT, "augru.npz"), **out)
    print("augru.npz", len(out), "arrays")


def gen_au_wide(rh):
    """ActivationUnit of the UNMODIFIED reference (models/ranking/din.py:58-93) at the widths its own example uses
    (dims [256, 128], examples/ranking/run_amazon_electronics.py:57) and at [64], in TRAIN mode (BatchNorm on the batch
    statistics of the B*L rows, pads included: SURVEY Q6): output, and the gradients of a random linear functional with
    respect to history, target and every parameter.  Kept in its own file so layers.npz stays byte-identical."""
    from torch_rechub.models.ranking.din import ActivationUnit
    torch.manual_seed(SEED + 11)
    g = torch.Generator().manual_seed(SEED + 12)
    out = {}
    for tag, D, B, L, dims, sm in (("w256", 16, 37, 9, [256, 128], False), ("w64", 16, 21, 5, [64], False),
                                   ("w128sm", 8, 19, 6, [128, 64], True), ("w192d4", 4, 33, 3, [192, 64], False)):
        au = ActivationUnit(D, dims=dims, activation="dice", use_softmax=sm)
        au.train()
        for n, p in au.named_parameters():
            if n.endswith("alpha"):
                with torch.no_grad():
                    p.copy_(torch.randn(p.shape, generator=g) * 0.5)
        hist = (torch.randn(B, L, D, generator=g) * 0.7).requires_grad_(True)
        tgt = (torch.randn(B, D, generator=g) * 0.7).requires_grad_(True)
        sd0 = {n: t.clone() for n, t in au.state_dict().items()}
        y = au(hist, tgt)
        G = torch.randn(y.shape, generator=g)
        (y * G).sum().backward()
        k = f"{tag}."
        out[k + "dims"], out[k + "softmax"] = np.array(dims), np.array(int(sm))
        out[k + "hist"], out[k + "tgt"], out[k + "out"], out[k + "G"] = npy(hist), npy(tgt), npy(y), npy(G)
        out[k + "g_hist"], out[k + "g_tgt"] = npy(hist.grad), npy(tgt.grad)
        for n, t in sd0.items():
            out[k + "sd0." + n] = npy(t)
        for n, t in au.state_dict().items():  # running statistics after the one training forward
            out[k + "sd1." + n] = npy(t)
        for n, p in au.named_parameters():
            out[k + "grad." + n] = npy(p.grad)
    np.savez_compressed(os.path.join(OUT, "au_wide.npz"), **out)
    print("au_wide.npz", len(out), "arrays")


MTL_CONFIGS = ["shared_bottom", "esmm", "mmoe", "mmoe_uwl", "ple", "aitm"]


def build_mtl(rh, cfg):
    """Multi-task models at fixture size; returns (model, feature groups, task types)."""
    from torch_rechub.basic.features import DenseFeature, SparseFeature
    from torch_rechub.models.multi_task import AITM, ESMM, MMOE, PLE, SharedBottom
    D = 16
    dense = [DenseFeature(f"I{i + 1}") for i in range(4)]
    sparse = [SparseFeature(f"C{i + 1}", vocab_size=v, embed_dim=D) for i, v in enumerate([3, 10, 27, 105, 305, 40, 24, 18])]
    feats = dense + sparse
    tower = {"dims": [8], "dropout": 0.0, "activation": "relu"}
    body = {"dims": [32, 16], "dropout": 0.0, "activation": "relu"}
    if cfg == "shared_bottom":
        types = ["classification", "regression"]
        return SharedBottom(feats, types, body, [dict(tower), dict(tower)]), {"features": feats}, types
    if cfg == "esmm":
        types = ["classification"] * 3  # columns: cvr, ctr, ctcvr (the trainer skips the first loss)
        return ESMM(sparse[:3], sparse[3:], dict(body), dict(body)), {"user_features": sparse[:3],
                                                                       "item_features": sparse[3:]}, types
    types = ["classification", "classification"]
    if cfg in ("mmoe", "mmoe_uwl"):
        return MMOE(feats, types, 3, body, [dict(tower), dict(tower)]), {"features": feats}, types
    if cfg == "ple":
        return PLE(feats, types, 2, 2, 1, body, [dict(tower), dict(tower)]), {"features": feats}, types
    if cfg == "aitm":
        return AITM(feats, 2, body, [dict(tower), dict(tower)]), {"features": feats}, types
    raise ValueError(cfg)


def gen_mtl(rh, cfg):
    """Predictions, per-task losses, every gradient and the 3-step trajectory of the reference MTLTrainer
    (trainers/mtl_trainer.py:112-165: mean of the task losses; ESMM: ctr + ctcvr; "uwl": learned loss weights)."""
    from torch_rechub.trainers import MTLTrainer
    torch.manual_seed(SEED)
    g = torch.Generator().manual_seed(SEED + 2)
    model, groups, types = build_mtl(rh, cfg)
    for m in model.modules():
        if isinstance(m, torch.nn.Embedding):
            torch.nn.init.normal_(m.weight, 0, 0.1, generator=g)
    B = 48
    batches = []
    for _ in range(3):
        x, _ = make_batch(groups, B, g)
        ys = torch.stack([(torch.rand(B, generator=g) < 0.3).float() if t == "classification" else
                          torch.randn(B, generator=g) for t in types], dim=1)
        batches.append((x, ys))
    out = {"spec": np.array(json.dumps({k: [spec_of(f) for f in v] for k, v in groups.items()})), "cfg": np.array(cfg),
           "task_types": np.array(json.dumps(types))}
    adaptive = {"method": "uwl"} if cfg.endswith("_uwl") else None
    trainer = MTLTrainer(model, task_types=types, optimizer_params={"lr": 1e-2, "weight_decay": 1e-3},
                         adaptive_params=adaptive, n_epoch=1, device="cpu")
    for n, t in model.state_dict().items():  # after the trainer: "uwl" registers its weights on the model
        out["sd0." + n] = npy(t)
    for bi, (bx, by) in enumerate(batches):
        for k, v in bx.items():
            out[f"x{bi}.{k}"] = npy(v)
        out[f"y{bi}"] = npy(by)
    x, ys = batches[0]
    model.eval()
    with torch.no_grad():
        out["pred_eval"] = npy(model(x))
    model.train()
    backup = {k: v.clone() for k, v in model.state_dict().items()}
    pred = model(x)
    losses = [trainer.loss_fns[i](pred[:, i], ys[:, i].float()) for i in range(len(types))]
    if cfg == "esmm":
        loss = sum(losses[1:])
    elif adaptive:
        loss = 0
        for li, wi in zip(losses, trainer.loss_weight):
            wi = torch.clamp(wi, min=0)
            loss = loss + 2 * li * torch.exp(-wi) + wi
    else:
        loss = sum(losses) / len(types)
    model.zero_grad()
    loss.backward()
    out["pred_train"], out["loss"] = npy(pred), np.array(loss.item())
    out["task_losses"] = np.array([l.item() for l in losses])
    for n, p in model.named_parameters():
        out["grad." + n] = npy(p.grad) if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
    model.load_state_dict(backup)
    model.zero_grad()
    per_task = trainer.train_one_epoch(batches)
    out["train.lr"], out["train.wd"] = np.array(1e-2), np.array(1e-3)
    out["train.task_losses"] = np.array(per_task)
    for n, t in model.state_dict().items():
        out["sd3." + n] = npy(t)
    np.savez_compressed(os.path.join(OUT, f"model_{cfg}.npz"), **out)
    print(f"model_{cfg}.npz", len(out), "arrays, loss", loss.item(), "train task losses", per_task)


def gen_inbatch(rh):
    """Random in-batch negatives of the UNMODIFIED reference (utils/match.py:104-145) on CPU for fixed generator seeds:
    the index stream torch_rechub_amd's ``stream="reference"`` mode must reproduce bit for bit."""
    from torch_rechub.utils.match import inbatch_negative_sampling
    out = {}
    for B, k, seed in [(4, 2, 0), (4, 2, 1), (6, 3, 2022), (9, None, 7), (33, 5, 3)]:
        g = torch.Generator().manual_seed(seed)
        first = inbatch_negative_sampling(torch.zeros((B, B)), neg_ratio=k, generator=g)
        second = inbatch_negative_sampling(torch.zeros((B, B)), neg_ratio=k, generator=g)  # generator state carries on
        out[f"B{B}_k{k}_seed{seed}.0"], out[f"B{B}_k{k}_seed{seed}.1"] = npy(first), npy(second)
    np.savez_compressed(os.path.join(OUT, "inbatch_random.npz"), **out)
    print("inbatch_random.npz", len(out), "arrays")


CONFIGS = ["deepfm_tutorial", "deepfm_criteo", "widedeep", "dcn", "dcnv2_mix", "dcnv2_full_stacked", "din",
           "din_softmax", "din_wide", "din_wide64", "din_wide_softmax", "dssm", "afm", "fibinet", "fibinet_each", "autoint", "edcn", "edcn_attention", "bst", "dien"]

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    rh = import_reference()
    only = sys.argv[1:]  # optional: regenerate just the named fixtures ("layers" or model configs)
    if not only or "layers" in only:
        gen_layers(rh)
    if not only or "augru" in only:
        gen_augru(rh)
    if not only or "inbatch" in only:
        gen_inbatch(rh)
    if not only or "au_wide" in only:
        gen_au_wide(rh)
    for cfg in CONFIGS:
        if not only or cfg in only:
            gen_model(rh, cfg)
    for cfg in MTL_CONFIGS:
        if not only or cfg in only:
            gen_mtl(rh, cfg)

"""INTEGRATION.md section B executed for real: ``torch_rechub_amd.integration.enable()`` against the imported,
UNMODIFIED reference package (only present in the build container; skipped on the GPU box)."""
import pytest
import torch

from oracle.ref_import import available, import_reference

pytestmark = pytest.mark.skipif(not available(), reason="reference checkout not present (GPU box)")


@pytest.fixture()
def patched():
    import_reference()
    import torch_rechub.models.ranking  # noqa: F401  (imported BEFORE enable(): the models already hold the layer classes)
    import torch_rechub.trainers  # noqa: F401
    from torch_rechub_amd import integration
    yield integration
    integration.disable()


def _features(mod):
    dense = [mod.DenseFeature(f"I{i}") for i in range(3)]
    sparse = [mod.SparseFeature(f"C{i}", vocab_size=v, embed_dim=16) for i, v in enumerate([7, 50, 300])]
    return dense, sparse


def test_layer_level_patch_runs_the_unmodified_reference_deepfm_on_the_hip_layers(patched):
    import torch_rechub.basic.features as RF
    import torch_rechub.models.ranking.deepfm as ref_deepfm
    from torch_rechub_amd.basic import layers as H
    mlp = {"dims": [32, 16], "dropout": 0.2, "activation": "relu"}
    torch.manual_seed(0)
    dense, sparse = _features(RF)
    before = ref_deepfm.DeepFM(dense + sparse, sparse, dict(mlp))  # the reference as it is
    ref_cls, ref_src = ref_deepfm.DeepFM, ref_deepfm.DeepFM.forward.__code__

    names = patched.enable(models=False, trainers=False)
    assert "torch_rechub.basic.layers.EmbeddingLayer" in names and "torch_rechub.basic.features.SparseFeature" in names
    assert ref_deepfm.DeepFM is ref_cls and ref_deepfm.DeepFM.forward.__code__ is ref_src  # model source untouched
    assert ref_deepfm.EmbeddingLayer is H.EmbeddingLayer and ref_deepfm.MLP is H.MLP  # ... but bound to the HIP layers
    dense, sparse = _features(RF)  # RF.SparseFeature is now the mirror class (attribute-compatible)
    after = ref_deepfm.DeepFM(dense + sparse, sparse, dict(mlp))
    assert type(after.embedding) is H.EmbeddingLayer and type(after.mlp) is H.MLP and type(after.fm) is H.FM
    sa, sb = before.state_dict(), after.state_dict()
    assert list(sa.keys()) == list(sb.keys())
    assert all(sa[k].shape == sb[k].shape and sa[k].dtype == sb[k].dtype for k in sa)
    after.load_state_dict(sa)  # a reference checkpoint loads into the patched model
    # and it IS the HIP path: CPU tensors are refused, nothing falls back to eager ATen
    x = {f.name: torch.zeros(4, dtype=torch.long) for f in sparse}
    x.update({f.name: torch.zeros(4) for f in dense})
    with pytest.raises(RuntimeError, match="HIP device"):
        after(x)

    patched.disable()
    assert ref_deepfm.EmbeddingLayer is not H.EmbeddingLayer
    import torch_rechub.basic.layers as RL
    assert ref_deepfm.EmbeddingLayer is RL.EmbeddingLayer


def test_model_and_trainer_level_patch(patched):
    import torch_rechub.basic.features as RF
    import torch_rechub.models.ranking as RR
    import torch_rechub.trainers as RT
    from torch_rechub_amd.models.ranking import DCNv2, DeepFM
    from torch_rechub_amd.trainers import CTRTrainer
    dense, sparse = _features(RF)
    ref_model = RR.DeepFM(dense + sparse, sparse, {"dims": [32, 16]})
    keys = list(ref_model.state_dict().keys())
    names = patched.enable()
    assert RR.DeepFM is DeepFM and RR.DCNv2 is DCNv2 and RT.CTRTrainer is CTRTrainer
    assert {"torch_rechub.models.ranking.DeepFM", "torch_rechub.trainers.CTRTrainer"} <= set(names)
    dense, sparse = _features(RF)
    m = RR.DeepFM(dense + sparse, sparse, {"dims": [32, 16]})
    assert list(m.state_dict().keys()) == keys
    with pytest.raises(RuntimeError, match="HIP"):  # the patched trainer drives the HIP path only
        RT.CTRTrainer(m, device="cpu")
    assert patched.enable() == []  # idempotent


@pytest.mark.gpu
def test_unmodified_reference_deepfm_runs_numerically_on_the_hip_layers(patched):
    """INTEGRATION.md section B on a device: fires wherever a reference checkout is mounted next to a GPU
    (RECHUB_REFERENCE=/path/to/torch-rechub; on the driver's GPU box there is none and the module-level skip applies).
    The UNMODIFIED ``torch_rechub.models.ranking.DeepFM`` source, bound to the HIP layers by ``enable(models=False,
    trainers=False)``, loaded with the reference fixture's weights: predictions, loss and every gradient against the vectors
    the same class produced on the reference's own CPU layers (tests/golden/model_deepfm_tutorial.npz), then three steps of
    the reference's OWN CTRTrainer (torch.optim.Adam reading the published ``weight.grad`` buffers) against its trajectory."""
    import numpy as np
    import torch_rechub.basic.features as RF
    import torch_rechub.models.ranking.deepfm as ref_deepfm
    import torch_rechub.trainers as RT
    from conftest import golden_batch, golden_state, load_golden
    import json
    gold = load_golden("model_deepfm_tutorial.npz")
    patched.enable(models=False, trainers=False)
    spec = json.loads(str(gold["spec"]))
    made = {}

    def fea(d):
        if d["name"] not in made:
            made[d["name"]] = (RF.DenseFeature(d["name"]) if d["kind"] == "DenseFeature" else
                               RF.SparseFeature(d["name"], vocab_size=d["vocab_size"], embed_dim=d["embed_dim"]))
        return made[d["name"]]

    deep, fm = [fea(d) for d in spec["deep_features"]], [fea(d) for d in spec["fm_features"]]
    model = ref_deepfm.DeepFM(deep, fm, {"dims": [32, 16], "dropout": 0.0, "activation": "relu"})
    model.load_state_dict(golden_state(gold, "sd0."))
    model = model.to("cuda:0")
    x, y = golden_batch(gold, 0)
    xd = {k: v.to("cuda:0") for k, v in x.items()}
    model.train()
    pred = model(xd)
    np.testing.assert_allclose(pred.detach().cpu().numpy(), gold["pred_train"], rtol=1e-5, atol=2e-6)
    loss = torch.nn.BCELoss()(pred, y.to("cuda:0").float())
    assert abs(loss.item() - float(gold["loss"])) < 2e-6
    loss.backward()
    gmax = max(float(np.abs(gold["grad." + n]).max()) for n, _ in model.named_parameters())
    for n, p_ in model.named_parameters():
        if n.endswith("mlp.0.bias") or n.endswith("mlp.4.bias"):  # in front of BatchNorm: rounding noise on both sides
            continue
        np.testing.assert_allclose(p_.grad.detach().cpu().numpy(), gold["grad." + n], rtol=1e-4, atol=2e-6 * gmax, err_msg=n)
    # the reference's own trainer loop over the three fixture batches
    model.load_state_dict(golden_state(gold, "sd0."))
    model.zero_grad()
    trainer = RT.CTRTrainer(model, optimizer_params={"lr": float(gold["train.lr"]), "weight_decay": float(gold["train.wd"])},
                            n_epoch=1, device="cuda:0")
    mean_loss = trainer.train_one_epoch([golden_batch(gold, i) for i in range(3)])
    assert abs(mean_loss - float(gold["train.mean_loss"])) < 5e-5
    ref = golden_state(gold, "sd3.")
    name = next(k for k in ref if "embed_dict" in k)
    got = model.state_dict()[name].cpu().numpy()
    assert np.abs(got - ref[name].numpy()).max() < 3e-4 + 1e-3 * np.abs(ref[name].numpy()).max()


def _reference_features(RF, spec_json):
    """Feature objects of the (patched) reference module from a fixture's json spec, shared by name (Q2)."""
    import json
    spec, made, groups = json.loads(str(spec_json)), {}, {}
    for gname, feas in spec.items():
        lst = []
        for d in feas:
            key = (d["kind"], d["name"])
            if key not in made:
                if d["kind"] == "DenseFeature":
                    made[key] = RF.DenseFeature(d["name"])
                elif d["kind"] == "SparseFeature":
                    made[key] = RF.SparseFeature(d["name"], vocab_size=d["vocab_size"], embed_dim=d["embed_dim"],
                                                 shared_with=d["shared_with"], padding_idx=d["padding_idx"])
                else:
                    made[key] = RF.SequenceFeature(d["name"], vocab_size=d["vocab_size"], embed_dim=d["embed_dim"],
                                                   pooling=d["pooling"], shared_with=d["shared_with"],
                                                   padding_idx=d["padding_idx"])
            lst.append(made[key])
        groups[gname] = lst
    return groups


def _unmodified_reference_model(cfg, groups):
    """The reference's OWN model classes (their source untouched; enable(models=False) only rebinds the layer names they
    import), built with the constructor calls of oracle/gen_golden.py::build_model."""
    import torch_rechub.models.ranking.dcn as ref_dcn
    import torch_rechub.models.ranking.dcn_v2 as ref_dcn_v2
    import torch_rechub.models.ranking.din as ref_din
    from torch_rechub_amd.models.ranking import DCN, DIN, DCNv2
    mlp = {"dims": [32, 16], "dropout": 0.0, "activation": "relu"}
    if cfg.startswith("din"):
        from conftest import DIN_ATTENTION_DIMS
        assert ref_din.DIN is not DIN
        return ref_din.DIN(groups["features"], groups["history_features"], groups["target_features"],
                           mlp_params={"dims": [32, 16], "dropout": 0.0},
                           attention_mlp_params={"dims": DIN_ATTENTION_DIMS.get(cfg, [16, 8]),
                                                 "use_softmax": cfg.endswith("softmax")})
    if cfg == "dcn":
        assert ref_dcn.DCN is not DCN
        return ref_dcn.DCN(groups["features"], 3, {"dims": [32, 16]})
    if cfg == "dcnv2_mix":
        assert ref_dcn_v2.DCNv2 is not DCNv2
        return ref_dcn_v2.DCNv2(groups["features"], 3, mlp, low_rank=8, num_experts=3)
    if cfg == "dcnv2_full_stacked":
        return ref_dcn_v2.DCNv2(groups["features"], 2, mlp, model_structure="stacked", use_low_rank_mixture=False)
    raise ValueError(cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["din", "din_wide", "din_softmax", "dcn", "dcnv2_mix", "dcnv2_full_stacked"])
def test_unmodified_reference_din_dcn_dcnv2_run_numerically_on_the_hip_layers(patched, cfg):
    """Same as the DeepFM run above for the UNMODIFIED reference DIN (its own ``ActivationUnit`` composed of the patched
    ``MLP`` / ``Dice``, models/ranking/din.py:38-92), DCN (models/ranking/dcn.py:32-38) and DCNv2
    (models/ranking/dcn_v2.py:47-59, CrossNetMix / CrossNetV2 from the patched layers): predictions (eval + train), loss,
    every gradient against the reference's CPU vectors, then three steps of the reference's own CTRTrainer."""
    import numpy as np
    import torch_rechub.basic.features as RF
    import torch_rechub.trainers as RT
    from conftest import assert_state_follows_reference_trajectory, golden_batch, golden_state, load_golden
    from torch_rechub_amd import ops
    from torch_rechub_amd.basic import layers as H
    gold = load_golden(f"model_{cfg}.npz")
    patched.enable(models=False, trainers=False)
    model = _unmodified_reference_model(cfg, _reference_features(RF, gold["spec"]))
    assert type(model.embedding) is H.EmbeddingLayer
    model.load_state_dict(golden_state(gold, "sd0."))
    model = model.to("cuda:0")
    x, y = golden_batch(gold, 0)
    xd = {k: v.to("cuda:0") for k, v in x.items()}
    model.eval()
    with torch.no_grad():
        pe = model(xd)
    np.testing.assert_allclose(pe.cpu().numpy(), gold["pred_eval"], rtol=1e-5, atol=2e-6)
    model.train()
    pred = model(xd)
    np.testing.assert_allclose(pred.detach().cpu().numpy(), gold["pred_train"], rtol=1e-5, atol=2e-6)
    loss = torch.nn.BCELoss()(pred, y.to("cuda:0").float())
    assert abs(loss.item() - float(gold["loss"])) < 2e-6
    loss.backward()
    ops.check_errors()
    gmax = max(float(np.abs(gold["grad." + n]).max()) for n, _ in model.named_parameters())
    noise = set()
    for mn, m in model.named_modules():
        if isinstance(m, torch.nn.Sequential):
            mods = list(m)
            noise |= {f"{mn}.{i}.bias" for i in range(len(mods) - 1)
                      if isinstance(mods[i], torch.nn.Linear) and isinstance(mods[i + 1], torch.nn.BatchNorm1d)}
    for n, p_ in model.named_parameters():
        ref = gold["grad." + n]
        got = p_.grad.detach().cpu().numpy() if p_.grad is not None else np.zeros_like(ref)
        if n in noise:
            assert np.abs(got).max() <= 1e-5 * max(gmax, 1e-3) + 1e-6, n
            continue
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-6 * gmax, err_msg=f"{cfg}: grad of {n}")
    model.load_state_dict(golden_state(gold, "sd0."))
    model.zero_grad()
    trainer = RT.CTRTrainer(model, optimizer_params={"lr": float(gold["train.lr"]), "weight_decay": float(gold["train.wd"])},
                            n_epoch=1, device="cuda:0")
    nb = sum(1 for k in gold.files if k.startswith("y") and k[1:].isdigit())
    mean_loss = trainer.train_one_epoch([golden_batch(gold, i) for i in range(nb)])
    assert abs(mean_loss - float(gold["train.mean_loss"])) < 5e-5
    assert_state_follows_reference_trajectory(gold, model.state_dict(), cfg, steps=nb)

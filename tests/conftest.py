import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "noise_tolerant: a HIP-vs-HIP comparison of two runs whose fp32 atomic order differs "
                                       "(tolerance with an outlier budget); ordered behind every oracle / bit-exact test")


def pytest_collection_modifyitems(config, items):
    """Order of the GPU suite under ``-x``: (0) everything that compares with the oracle, the reference's golden vectors or
    demands bit-equality -- kernels, full-size properties, the bit-equality tests of the captured step forms bench.py times --
    then (1) the tolerance comparisons with the reference trainer's trajectory, and LAST (2) the tests marked
    ``noise_tolerant``: two HIP runs compared under nondeterministic float atomics, where Adam turns summation-order noise
    of near-cancelling gradient elements into lr-sized steps.  Round 4's driver run stopped at such a test (#318 of 384) and
    never executed the 66 tests behind it, among them every bit-equality test of the timed step form.  The sort is stable:
    inside a class the file order is kept."""
    def rank(item):
        if item.get_closest_marker("gpu") is None:
            return 0
        if item.get_closest_marker("noise_tolerant") is not None:
            return 3
        name = item.nodeid
        if "test_gpu_kernels.py" in name or "test_gpu_properties.py" in name:
            return 0
        if "bitwise" in name or "bit_identical" in name or "leaves_no_row_behind" in name or "abandoned_step" in name:
            return 1
        return 2
    items.sort(key=rank)


@pytest.fixture(autouse=True)
def poison_free_device_memory(request):
    """Before every GPU test: fill what the caching allocator will hand out next with NaN bit patterns, so that a kernel
    that leaves part of a ``torch.empty`` partial / workspace buffer unwritten produces NaN instead of whatever the
    previous test left there (found that way: a launch with fewer workgroups than the caller's partial rows)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    if torch.cuda.is_available():
        junk = [torch.full((64 * 1024,), float("nan"), device="cuda") for _ in range(48)]  # small-block pool (256 KB each)
        junk.append(torch.full((16 * 1024 * 1024,), float("nan"), device="cuda"))         # large-block pool (64 MB)
        del junk
    yield


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def layers_golden():
    return load_golden("layers.npz")


MODEL_CONFIGS = ["deepfm_tutorial", "deepfm_criteo", "widedeep", "dcn", "dcnv2_mix", "dcnv2_full_stacked", "din",
                 "din_softmax", "din_wide", "din_wide64", "din_wide_softmax", "dssm", "afm", "fibinet", "fibinet_each", "autoint", "edcn", "edcn_attention", "bst", "dien"]
# attention MLP widths per DIN fixture (oracle/gen_golden.py::DIN_ATTENTION_DIMS).  The "wide" ones are the shapes whose
# first attention layer runs on csrc/dinmlp.hip -- "din_wide" = the reference's own [256, 128]
# (examples/ranking/run_amazon_electronics.py:57).
DIN_ATTENTION_DIMS = {"din_wide": [256, 128], "din_wide64": [64], "din_wide_softmax": [128, 64]}
AUX_LOSS_CONFIGS = {"dien"}  # forward returns (prediction, weighted auxiliary loss): CTRTrainer(loss_mode=False)


def features_from_spec(spec_json):
    """Rebuild feature objects (torch_rechub_amd classes) from the json spec stored in a model fixture.

    Feature objects are shared between groups by name, as in the generator (Q2 semantics)."""
    from torch_rechub_amd.basic.features import DenseFeature, SequenceFeature, SparseFeature
    spec = json.loads(str(spec_json))
    made = {}
    groups = {}
    for gname, feas in spec.items():
        lst = []
        for d in feas:
            key = (d["kind"], d["name"])
            if key not in made:
                if d["kind"] == "DenseFeature":
                    made[key] = DenseFeature(d["name"], d["embed_dim"])
                elif d["kind"] == "SparseFeature":
                    made[key] = SparseFeature(d["name"], d["vocab_size"], d["embed_dim"], shared_with=d["shared_with"],
                                              padding_idx=d["padding_idx"])
                else:
                    made[key] = SequenceFeature(d["name"], d["vocab_size"], d["embed_dim"], pooling=d["pooling"],
                                                shared_with=d["shared_with"], padding_idx=d["padding_idx"])
            lst.append(made[key])
        groups[gname] = lst
    return groups


def build_amd_model(cfg, groups):
    """Same constructor calls as oracle/gen_golden.py::build_model, on the torch_rechub_amd classes."""
    from torch_rechub_amd.models.ranking import DCN, DIN, DCNv2, DeepFM, WideDeep
    mlp = {"dims": [32, 16], "dropout": 0.0, "activation": "relu"}
    if cfg == "dssm":
        from torch_rechub_amd.models.matching import DSSM
        tower = {"dims": [32, 16], "activation": "prelu"}
        return DSSM(groups["user_features"], groups["item_features"], user_params=dict(tower), item_params=dict(tower),
                    temperature=0.02)
    if cfg == "bst":
        from torch_rechub_amd.models.ranking import BST
        return BST(groups["features"], groups["history_features"], groups["target_features"], mlp_params=mlp, nhead=2,
                   dropout=0.0, num_layers=1, max_seq_len=8)
    if cfg == "dien":
        from torch_rechub_amd.models.ranking import DIEN
        return DIEN(groups["features"], groups["history_features"], groups["neg_history_features"],
                    groups["target_features"], mlp_params={"dims": [32, 16], "dropout": 0.0}, alpha=0.2)
    if cfg.startswith("din"):
        return DIN(groups["features"], groups["history_features"], groups["target_features"],
                   mlp_params={"dims": [32, 16], "dropout": 0.0},
                   attention_mlp_params={"dims": DIN_ATTENTION_DIMS.get(cfg, [16, 8]),
                                         "use_softmax": cfg.endswith("softmax")})
    if cfg.startswith("deepfm"):
        return DeepFM(groups["deep_features"], groups["fm_features"], mlp)
    if cfg == "widedeep":
        return WideDeep(groups["wide_features"], groups["deep_features"], mlp)
    if cfg == "dcn":
        return DCN(groups["features"], 3, {"dims": [32, 16]})
    if cfg == "dcnv2_mix":
        return DCNv2(groups["features"], 3, mlp, low_rank=8, num_experts=3)
    if cfg == "dcnv2_full_stacked":
        return DCNv2(groups["features"], 2, mlp, model_structure="stacked", use_low_rank_mixture=False)
    if cfg == "afm":
        from torch_rechub_amd.models.ranking import AFM
        return AFM(groups["fm_features"], 16, t=8)
    if cfg in ("edcn", "edcn_attention"):
        from torch_rechub_amd.models.ranking import EDCN
        return EDCN(groups["features"], 2, {"dropout": 0.0, "activation": "relu"},
                    bridge_type="hadamard_product" if cfg == "edcn" else "attention_pooling")
    if cfg == "autoint":
        from torch_rechub_amd.models.ranking import AutoInt
        return AutoInt(groups["sparse_features"], groups["dense_features"], num_layers=2, num_heads=2, dropout=0.0,
                       mlp_params=mlp)
    if cfg in ("fibinet", "fibinet_each"):
        from torch_rechub_amd.models.ranking import FiBiNet
        return FiBiNet(groups["features"], mlp, reduction_ratio=3,
                       bilinear_type="field_interaction" if cfg == "fibinet" else "field_each")
    raise ValueError(cfg)


MTL_CONFIGS = ["shared_bottom", "esmm", "mmoe", "mmoe_uwl", "ple", "aitm"]


def build_mtl_model(cfg, groups, task_types):
    """Same constructor calls as oracle/gen_golden.py::build_mtl, on the torch_rechub_amd classes."""
    from torch_rechub_amd.models.multi_task import AITM, ESMM, MMOE, PLE, SharedBottom
    tower = {"dims": [8], "dropout": 0.0, "activation": "relu"}
    body = {"dims": [32, 16], "dropout": 0.0, "activation": "relu"}
    towers = [dict(tower), dict(tower)]
    if cfg == "shared_bottom":
        return SharedBottom(groups["features"], task_types, body, towers)
    if cfg == "esmm":
        return ESMM(groups["user_features"], groups["item_features"], dict(body), dict(body))
    if cfg in ("mmoe", "mmoe_uwl"):
        return MMOE(groups["features"], task_types, 3, body, towers)
    if cfg == "ple":
        return PLE(groups["features"], task_types, 2, 2, 1, body, towers)
    if cfg == "aitm":
        return AITM(groups["features"], 2, body, towers)
    raise ValueError(cfg)


def golden_batch(gold, bi):
    import torch
    x = {k[len(f"x{bi}."):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(f"x{bi}.")}
    y = torch.from_numpy(gold[f"y{bi}"])
    return x, y


def golden_state(gold, prefix):
    import torch
    return {k[len(prefix):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(prefix)}


def assert_trajectory_close(got, want, travel, what, atol=3e-4, rtol=1e-3, outlier_frac=5e-3):
    """Adam divides by sqrt(v): an element whose gradient nearly cancels amplifies fp32 summation-order noise, so a
    handful of elements may drift by a fraction of lr per step.  >= 99.5 % of the elements must agree to atol/rtol
    and every element to within a quarter of the distance Adam can travel in these steps."""
    diff = np.abs(got - want)
    bad = diff > atol + rtol * np.abs(want)
    # (at least two elements: in a 3 x 16 table 0.5 % is less than one, and single near-cancelling elements do jump)
    assert bad.sum() <= max(2, outlier_frac * bad.size), f"{what}: {bad.sum()} / {bad.size} elements off (max {diff.max():.3e})"
    assert diff.max() <= 0.25 * travel + atol, f"{what}: max diff {diff.max():.3e}"


def assert_state_follows_reference_trajectory(gold, mine, cfg, steps=3):
    """``mine`` (a state_dict after ``steps`` trainer steps from the fixture's ``sd0.``) against the state the reference's
    own trainer reached (``sd3.``), tensor by tensor.  Tensors whose gradient is rounding noise on both sides (a bias in
    front of BatchNorm, a cross-layer bias feeding Linear -> BatchNorm, the KEY bias of an attention softmax) have no
    defined trajectory under Adam -- the SIGN of the noise becomes +-lr steps -- and are only bounded."""
    ref = golden_state(gold, "sd3.")
    gmax = max(float(np.abs(gold[k]).max()) for k in gold.files if k.startswith("grad."))
    lr = float(gold["train.lr"])
    for k, v in ref.items():
        got = mine[k].detach().cpu().numpy()
        if k.endswith("num_batches_tracked"):
            assert int(got) == int(v)
            continue
        if "grad." + k in gold.files and float(np.abs(gold["grad." + k]).max()) < 1e-5 * gmax:
            assert np.abs(got - v.numpy()).max() <= 2.1 * lr * steps, k
            continue
        if k.endswith("running_mean"):
            # the batch mean of (W x + b) carries the noise-driven drift of the bias b above one-for-one
            assert np.abs(got - v.numpy()).max() <= 0.5 * lr * steps, k
            continue
        want = v.numpy()
        if k.endswith("self_attn.in_proj_bias"):
            d = got.shape[0] // 3
            assert np.abs(got[d:2 * d] - want[d:2 * d]).max() <= 2.1 * lr * steps, k
            got, want = np.delete(got, np.s_[d:2 * d]), np.delete(want, np.s_[d:2 * d])
        assert_trajectory_close(got, want, lr * steps, f"{cfg}: {k} after {steps} steps")
    return ref

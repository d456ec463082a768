"""Host-side logic that needs no GPU: API surface, checkpoint keys, feature sharing, loud failure on CPU
tensors, loaders, regularisation arithmetic (values from the reference's tests/test_regularization.py)."""
import numpy as np
import pytest
import torch
from torch import nn

from conftest import MODEL_CONFIGS, MTL_CONFIGS, build_amd_model, build_mtl_model, features_from_spec, golden_batch, golden_state, load_golden


@pytest.mark.parametrize("cfg", MODEL_CONFIGS)
def test_state_dict_keys_and_shapes_match_reference(cfg):
    gold = load_golden(f"model_{cfg}.npz")
    model = build_amd_model(cfg, features_from_spec(gold["spec"]))
    ref = golden_state(gold, "sd0.")
    mine = model.state_dict()
    assert list(mine.keys()) == list(ref.keys())  # same names, same order: model.pth round-trips
    for k in ref:
        assert tuple(mine[k].shape) == tuple(ref[k].shape), k
    model.load_state_dict(ref)  # strict


def test_tables_stay_nn_embedding_and_are_shared_across_models():
    from torch_rechub_amd.basic.features import DenseFeature, SparseFeature
    from torch_rechub_amd.models.ranking import DCN, DeepFM
    dense = [DenseFeature("d0")]
    sparse = [SparseFeature("a", 10, 16), SparseFeature("b", 7, 16, shared_with="a")]
    m1 = DeepFM(dense + sparse, sparse, {"dims": [8]})
    m2 = DCN(dense + sparse, 2, {"dims": [8]})
    assert isinstance(m1.embedding.embed_dict["a"], nn.Embedding)
    assert "b" not in m1.embedding.embed_dict  # shared_with features own no table (layers.py:69-72)
    assert m1.embedding.embed_dict["a"] is m2.embedding.embed_dict["a"]  # cached on the Feature object (Q2)
    assert m1.embedding.n_dense == 1


def test_initializers_zero_padding_row_and_default_std():
    from torch_rechub_amd.basic.features import SparseFeature
    from torch_rechub_amd.basic.initializers import RandomUniform, XavierNormal
    torch.manual_seed(0)
    t = SparseFeature("x", 5000, 16, padding_idx=3).get_embedding_layer()
    assert torch.all(t.weight[3] == 0) and t.padding_idx == 3
    assert 0.5e-4 < t.weight.std().item() < 2e-4  # RandomNormal(0, 1e-4) default (features.py:54)
    u = RandomUniform(-1, 1)(50, 8, padding_idx=0)
    assert torch.all(u.weight[0] == 0) and u.weight.abs().max() <= 1
    assert XavierNormal()(50, 8).weight.shape == (50, 8)


def test_auto_embedding_dim_rule():
    from torch_rechub_amd.basic.features import SparseFeature, get_auto_embedding_dim
    assert get_auto_embedding_dim(10000) == 60
    assert SparseFeature("x", 16).embed_dim == 12


def test_ops_refuse_cpu_tensors_loudly():
    gold = load_golden("model_deepfm_tutorial.npz")
    model = build_amd_model("deepfm_tutorial", features_from_spec(gold["spec"]))
    x, _ = golden_batch(gold, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(x)
    from torch_rechub_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.fm(torch.zeros(2, 3, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.cross_network(torch.zeros(2, 8), torch.zeros(1, 8), torch.zeros(1, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.augru(torch.zeros(2, 3, 12), torch.ones(2, 3), torch.zeros(4, 12))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.fused_rows(torch.zeros(2, 8), 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.shard_localize(torch.zeros(2, 1, dtype=torch.int64), torch.tensor([4, -1, 2]), 2, 0)
    for cfg in ("dien", "bst", "din"):  # sequence models: nothing of them runs on CPU tensors either
        g = load_golden(f"model_{cfg}.npz")
        m = build_amd_model(cfg, features_from_spec(g["spec"]))
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m(golden_batch(g, 0)[0])


def test_trainer_refuses_cpu_device():
    from torch_rechub_amd.trainers import CTRTrainer
    gold = load_golden("model_dcn.npz")
    model = build_amd_model("dcn", features_from_spec(gold["spec"]))
    with pytest.raises(RuntimeError, match="HIP device"):
        CTRTrainer(model, device="cpu")


def test_embedding_layer_error_messages_match_reference():
    from torch_rechub_amd.basic.features import DenseFeature, SequenceFeature
    from torch_rechub_amd.basic.layers import EmbeddingLayer, InputMask
    lay = EmbeddingLayer([DenseFeature("d")])
    with pytest.raises(ValueError, match="expected SparseFeatures"):
        lay({"d": torch.zeros(3)}, [DenseFeature("d")], squeeze_dim=False)
    assert lay({"d": torch.arange(3)}, [DenseFeature("d")], squeeze_dim=True).shape == (3, 1)  # dense-only path
    bad = SequenceFeature("h", 10, 16, pooling="max")
    lay2 = EmbeddingLayer([bad])
    with pytest.raises(ValueError, match="Sequence pooling method supports only"):
        lay2({"h": torch.zeros(2, 3, dtype=torch.long)}, [bad])
    with pytest.raises(ValueError, match="Only SparseFeature or SequenceFeature"):
        InputMask()({"d": torch.zeros(3)}, DenseFeature("d"))


def test_input_mask_and_pooling_shims():
    from torch_rechub_amd.basic.features import SequenceFeature
    from torch_rechub_amd.basic.layers import AveragePooling, InputMask, SumPooling
    idx = torch.tensor([[3, 4, 0, 0], [1, 0, 0, 0]])
    x = {"h": idx}
    m0 = InputMask()(x, SequenceFeature("h", 10, 4, padding_idx=0))
    assert m0.shape == (2, 1, 4) and m0.sum().item() == 3
    m1 = InputMask()(x, SequenceFeature("h", 10, 4))  # sentinel -1: nothing masked
    assert m1.sum().item() == 8
    e = torch.arange(2 * 4 * 3, dtype=torch.float32).view(2, 4, 3)
    np.testing.assert_allclose(SumPooling()(e, m0).numpy(), torch.bmm(m0, e).squeeze(1).numpy())
    np.testing.assert_allclose(AveragePooling()(e, m0).numpy(),
                               (torch.bmm(m0, e).squeeze(1) / (m0.sum(-1) + 1e-16)).numpy())


def test_regularization_loss_values():
    """Known answers of the reference's tests/test_regularization.py style: exact sums by hand."""
    from torch_rechub_amd.basic.loss_func import RegularizationLoss

    class M(nn.Module):

        def __init__(self):
            super().__init__()
            self.emb = nn.Embedding(3, 2)
            self.fc = nn.Linear(2, 1)
            self.bn = nn.BatchNorm1d(1)
            with torch.no_grad():
                self.emb.weight.copy_(torch.tensor([[1., -2.], [3., 0.], [0.5, 0.5]]))
                self.fc.weight.copy_(torch.tensor([[2., -1.]]))
                self.fc.bias.fill_(0.5)

    m = M()
    assert RegularizationLoss()(m) == 0.0  # python float when disabled (ctr_trainer.py adds it to the loss)
    l = RegularizationLoss(embedding_l1=0.1, embedding_l2=0.01, dense_l1=0.2, dense_l2=0.02)(m)
    emb_l1, emb_l2 = 1 + 2 + 3 + 0.5 + 0.5, 1 + 4 + 9 + 0.25 + 0.25
    den_l1, den_l2 = 2 + 1 + 0.5, 4 + 1 + 0.25  # BatchNorm params skipped
    assert abs(l.item() - (0.1 * emb_l1 + 0.01 * emb_l2 + 0.2 * den_l1 + 0.02 * den_l2)) < 1e-6
    l.backward()
    assert m.emb.weight.grad is not None and m.bn.weight.grad is None


def test_data_generator_split_and_batches():
    from torch_rechub_amd.utils.data import DataGenerator
    n = 100
    x = {"a": np.arange(n), "b": np.random.rand(n).astype(np.float32)}
    y = (np.arange(n) % 2)
    tr, va, te = DataGenerator(x, y).generate_dataloader(split_ratio=[0.7, 0.1], batch_size=16)
    assert len(tr.dataset) == 70 and len(va.dataset) == 10 and len(te.dataset) == 20
    xb, yb = next(iter(tr))
    assert set(xb) == {"a", "b"} and xb["a"].shape == (16,) and yb.dtype == torch.int64
    seen = sorted(int(v) for loader in (tr, va, te) for xb, _ in loader for v in xb["a"])
    assert seen == list(range(n))  # every row exactly once across the three splits


def test_pack_indices_zero_copy_and_fallback():
    from torch_rechub_amd.distributed import pack_indices
    base = torch.arange(24).view(6, 4)
    cols = [base[:, j] for j in range(4)]
    packed = pack_indices(cols)
    assert packed.data_ptr() == base.data_ptr() and torch.equal(packed, base)
    loose = [torch.arange(6), torch.arange(6) + 10]
    assert torch.equal(pack_indices(loose), torch.stack(loose, 1))


def test_dense_grad_bucket_packs_fresh_gradients_cpu():
    from torch_rechub_amd.distributed import DenseGradBucket
    lin = nn.Sequential(nn.Linear(3, 2), nn.Linear(2, 1))
    bucket = DenseGradBucket(list(lin.parameters()))
    assert bucket.flat.numel() == 6 + 2 + 2 + 1 and bucket.world == 1
    bucket.zero()
    assert all(p.grad is None for p in lin.parameters())  # autograd will hand over fresh tensors (no accumulate kernels)
    lin(torch.ones(4, 3)).sum().backward()
    assert bucket.all_present()
    grads = [p.grad.clone() for p in lin.parameters()]
    lin[0].bias.grad = None  # a parameter without gradient contributes zeros
    bucket.finish(assign_views=True)
    assert all(bucket.packed)
    for i, (p, g) in enumerate(zip(lin.parameters(), grads)):
        want = torch.zeros_like(g) if i == 1 else g
        assert torch.equal(bucket.view(i), want)
        assert p.grad.data_ptr() == bucket.view(i).data_ptr()  # stock optimizers read the bucket through p.grad
    bucket.zero()
    assert not any(bucket.packed)


def test_table_parameters_detection():
    from torch_rechub_amd.distributed import table_parameters
    gold = load_golden("model_din.npz")
    model = build_amd_model("din", features_from_spec(gold["spec"]))
    names = {n for n, p in model.named_parameters() if any(p is q for q in table_parameters(model))}
    assert names == {"embedding.embed_dict.user_id.weight", "embedding.embed_dict.target_item.weight",
                     "embedding.embed_dict.target_cate.weight"}


def test_inbatch_sampling_known_answers_cpu():
    """The reference's own unit tests for the in-batch sampler (tests/test_inbatch_sampling.py:12-30)."""
    from torch_rechub_amd.utils.match import gather_inbatch_logits, inbatch_negative_sampling
    scores = torch.tensor([[1., 2, 3], [4, 5, 6], [7, 8, 0]])
    assert inbatch_negative_sampling(scores, neg_ratio=1, hard_negative=True).flatten().tolist() == [2, 2, 1]
    big = torch.randn(4, 4)
    g = torch.Generator().manual_seed(0)
    a = inbatch_negative_sampling(big, neg_ratio=3, generator=g)
    assert a.shape == (4, 3) and not (a == torch.arange(4).unsqueeze(1)).any()
    assert inbatch_negative_sampling(big, neg_ratio=2, generator=torch.Generator().manual_seed(1)).shape == (4, 2)
    logits = gather_inbatch_logits(scores, torch.tensor([[2], [2], [1]]))
    assert logits.tolist() == [[1., 3.], [5., 6.], [0., 8.]]
    with pytest.raises(ValueError):
        inbatch_negative_sampling(torch.zeros(1, 1))
    with pytest.raises(ValueError):
        inbatch_negative_sampling(torch.zeros(3))


def test_inbatch_random_stream_is_the_references_bit_for_bit():
    """stream="reference" (what CPU tensors always use): same indices as the unmodified reference for the same generator
    state, also on the second call (the generator carries on) -- tests/golden/inbatch_random.npz was written by
    oracle/gen_golden.py::gen_inbatch from /root/reference/torch_rechub/utils/match.py:104-145."""
    from torch_rechub_amd.utils.match import inbatch_negative_sampling
    gold = load_golden("inbatch_random.npz")
    seen = 0
    for key in gold.files:
        if not key.endswith(".0"):
            continue
        name = key[:-2]
        B, k, seed = (name.split("_")[0][1:], name.split("_")[1][1:], name.split("_")[2][4:])
        B, k, seed = int(B), (None if k == "None" else int(k)), int(seed)
        g = torch.Generator().manual_seed(seed)
        for call in (0, 1):
            got = inbatch_negative_sampling(torch.zeros(B, B), neg_ratio=k, generator=g)
            assert torch.equal(got, torch.from_numpy(gold[f"{name}.{call}"])), (name, call)
        seen += 1
    assert seen == 5
    # rank slices of a global problem: two "ranks" seeded alike draw what one process draws for the 6 x 6 batch
    whole = inbatch_negative_sampling(torch.zeros(6, 6), neg_ratio=3, generator=torch.Generator().manual_seed(5))
    for r in range(2):
        part = inbatch_negative_sampling(torch.zeros(3, 6), neg_ratio=3, generator=torch.Generator().manual_seed(5),
                                         row_offset=3 * r)
        assert torch.equal(part, whole[3 * r:3 * r + 3])
    # the form without a score matrix (the trainer's default for random negatives: ops.inbatch_logits needs only the
    # indices): same stream, same indices, same argument checks
    from torch_rechub_amd.utils.match import random_inbatch_negatives
    for B, C, row0, k in ((7, 7, 0, 3), (3, 6, 3, 2), (5, 5, 0, None)):
        a = inbatch_negative_sampling(torch.zeros(B, C), neg_ratio=k, generator=torch.Generator().manual_seed(9),
                                      row_offset=row0)
        b = random_inbatch_negatives(B, C, torch.device("cpu"), neg_ratio=k, generator=torch.Generator().manual_seed(9),
                                     row_offset=row0)
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        random_inbatch_negatives(1, 1, torch.device("cpu"))
    with pytest.raises(ValueError):
        random_inbatch_negatives(3, 4, torch.device("cpu"), row_offset=2)


def test_inbatch_random_stream_against_the_live_reference():
    from oracle.ref_import import available, import_reference
    if not available():
        pytest.skip("reference checkout not present (GPU box)")
    import_reference()
    from torch_rechub.utils.match import inbatch_negative_sampling as ref_fn
    from torch_rechub_amd.utils.match import inbatch_negative_sampling
    for B, k, seed in [(5, 2, 11), (17, 16, 12), (40, 7, 13)]:
        a = ref_fn(torch.zeros(B, B), neg_ratio=k, generator=torch.Generator().manual_seed(seed))
        b = inbatch_negative_sampling(torch.zeros(B, B), neg_ratio=k, generator=torch.Generator().manual_seed(seed))
        assert torch.equal(a, b)


def test_match_trainer_rejects_models_without_towers():
    from torch_rechub_amd.trainers import MatchTrainer
    with pytest.raises(ValueError, match="does not support in-batch negative sampling"):
        MatchTrainer(nn.Linear(2, 1), in_batch_neg=True, device="cuda:0")


@pytest.mark.parametrize("cfg", MTL_CONFIGS)
def test_multi_task_state_dict_keys_and_shapes_match_reference(cfg):
    """Checkpoint ABI of the multi-task mirrors (models/multi_task/*.py): same keys, order and shapes as the reference
    model the fixture was dumped from ("uwl": the trainer registers the loss weights under "loss weight")."""
    import json
    gold = load_golden(f"model_{cfg}.npz")
    types = json.loads(str(gold["task_types"]))
    model = build_mtl_model(cfg, features_from_spec(gold["spec"]), types)
    if cfg.endswith("_uwl"):
        model.add_module("loss weight", torch.nn.ParameterList(torch.nn.Parameter(torch.zeros(1)) for _ in types))
    want = golden_state(gold, "sd0.")
    have = model.state_dict()
    assert list(have.keys()) == list(want.keys())
    for k in want:
        assert tuple(have[k].shape) == tuple(want[k].shape), k


def test_mtl_trainer_argument_checks():
    from torch_rechub_amd.trainers import MTLTrainer
    from torch_rechub_amd.utils.data import get_loss_func, get_metric_func
    assert isinstance(get_loss_func("classification"), torch.nn.BCELoss)
    assert isinstance(get_loss_func("regression"), torch.nn.MSELoss)
    assert get_metric_func("classification").__name__ == "roc_auc_score"
    with pytest.raises(ValueError):
        get_loss_func("ranking")
    with pytest.raises(NotImplementedError):
        MTLTrainer(torch.nn.Linear(2, 2), ["classification"] * 2, adaptive_params={"method": "gradnorm"}, device="cuda:0")
    with pytest.raises(RuntimeError):  # the HIP hot path has no CPU mode
        MTLTrainer(torch.nn.Linear(2, 2), ["classification"] * 2, device="cpu")


def test_early_stopper_patience_rule_and_best_weight_snapshot():
    """Reference rule (basic/callback.py:17-33): a strictly better AUC resets the counter and deep-copies the weights;
    otherwise the counter advances and the call turns True on the patience-th evaluation in a row without improvement."""
    from torch_rechub_amd.basic.callback import EarlyStopper
    es = EarlyStopper(patience=3)
    w = {"a": torch.zeros(2)}
    assert es.stop_training(0.60, w) is False and es.best_auc == 0.60
    w["a"].add_(1)  # the snapshot must not follow later in-place updates
    assert torch.equal(es.best_weights["a"], torch.zeros(2))
    assert [es.stop_training(v, w) for v in (0.60, 0.59)] == [False, False]  # equal is not an improvement
    assert es.trial_counter == 2
    assert es.stop_training(0.61, w) is False and es.trial_counter == 0 and torch.equal(es.best_weights["a"], torch.ones(2))
    assert [es.stop_training(0.5, w) for _ in range(3)] == [False, False, True]
    assert EarlyStopper(patience=1).stop_training(0.0, w) is True  # AUC 0 never beats the initial best of 0


def test_bpr_loss_values():
    """-log(sigmoid(pos - neg)).mean() for 1-D negatives and for (B, K) in-batch negatives (basic/loss_func.py:95-107)."""
    from torch_rechub_amd.basic.loss_func import BPRLoss
    pos = torch.tensor([[1.0], [0.5], [-0.2]])
    neg = torch.tensor([0.3, 0.9, -0.1])
    want = -torch.log(torch.sigmoid(pos.view(-1) - neg)).mean()
    assert torch.allclose(BPRLoss()(pos, neg), want)
    negs = torch.tensor([[0.3, 0.1], [0.9, 0.2], [-0.1, 0.0]])
    want = -torch.log(torch.sigmoid(pos.view(-1, 1) - negs)).mean()
    assert torch.allclose(BPRLoss()(pos, negs, in_batch_neg=True), want)


def test_padded_width_tables_keep_the_reference_checkpoint_layout():
    """embed_dim outside 4, 8, 16, 32, 64, 128 (the reference's default embed_dim=None gives floor(6 V^0.25): 10, 18, 33,
    features.py:54-60): the table is STORED at the next kernel width with zero columns, state_dict / load_state_dict
    keep the reference's (vocab, embed_dim) layout."""
    from torch_rechub_amd.basic.features import SparseFeature
    from torch_rechub_amd.basic.initializers import PaddedEmbedding, XavierNormal
    from torch_rechub_amd.basic.layers import EmbeddingLayer
    from torch_rechub_amd.models.ranking import DeepFM
    f_auto = SparseFeature("a", vocab_size=100)  # 6 * 100^0.25 = 18.97 -> 18
    assert f_auto.embed_dim == 18
    feas = [f_auto, SparseFeature("b", 50, embed_dim=10, padding_idx=0, initializer=XavierNormal()),
            SparseFeature("c", 7, embed_dim=16)]
    layer = EmbeddingLayer(feas)
    ta, tb, tc = (layer.embed_dict[n] for n in "abc")
    assert isinstance(ta, PaddedEmbedding) and ta.weight.shape == (100, 32) and ta.embedding_dim == 18
    assert isinstance(tb, PaddedEmbedding) and tb.weight.shape == (50, 16) and not isinstance(tc, PaddedEmbedding)
    assert not ta.weight[:, 18:].any() and not tb.weight[:, 10:].any() and not tb.weight[0].any()
    assert ta.weight[:, :18].abs().max() > 0 and tb.weight[1:, :10].abs().max() > 0
    sd = layer.state_dict()
    assert sd["embed_dict.a.weight"].shape == (100, 18) and sd["embed_dict.b.weight"].shape == (50, 10)
    assert sd["embed_dict.a.weight"].data_ptr() == ta.weight.data_ptr()  # a view of the parameter, as state_dict() is
    new = {"embed_dict.a.weight": torch.randn(100, 18), "embed_dict.b.weight": torch.randn(50, 10),
           "embed_dict.c.weight": torch.randn(7, 16)}
    layer.load_state_dict(new)
    assert torch.equal(ta.weight[:, :18], new["embed_dict.a.weight"]) and not ta.weight[:, 18:].any()
    assert torch.equal(ta(torch.tensor([3, 4])), new["embed_dict.a.weight"][[3, 4]])  # direct use: logical rows
    # model level: a DeepFM over width-10 features has the reference's parameter shapes
    fm = [SparseFeature(f"s{i}", 20 + i, embed_dim=10) for i in range(3)]
    m = DeepFM(fm, fm, {"dims": [8]})
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert shapes["embedding.embed_dict.s0.weight"] == (20, 10) and shapes["linear.fc.weight"] == (1, 30)
    assert shapes["mlp.mlp.0.weight"] == (8, 30)
    assert layer.compact(torch.arange(32 + 16 + 16 + 2.0).view(1, -1), feas, 2).shape == (1, 18 + 10 + 16 + 2)
    got = layer.compact(torch.arange(66.0).view(1, -1), feas, 2)[0].tolist()
    assert got == list(range(18)) + list(range(32, 42)) + list(range(48, 64)) + [64, 65]
    lw = layer.pad_lr_weight(torch.arange(44.0).view(1, 44), feas)
    assert lw.shape == (1, 64) and lw[0, 18:32].abs().sum() == 0 and lw[0, 32] == 18 and lw[0, 48] == 28


def test_torch_library_ops_trace_under_fake_tensors():
    """the stateless interaction kernels are registered with torch.library (schema + fake impl + autograd):
    FakeTensorMode propagates shapes through them without a device or a kernel, and the real implementations still
    refuse CPU tensors (no fallback)."""
    from torch._subclasses.fake_tensor import FakeTensorMode

    import torch_rechub_amd.library  # noqa: F401  (registers torch.ops.rechub_hip.*)
    with FakeTensorMode():
        x = torch.empty(64, 26, 16)
        assert torch.ops.rechub_hip.fm(x, True).shape == (64, 1)
        assert torch.ops.rechub_hip.fm(x, False).shape == (64, 16)
        z = torch.empty(64, 429, requires_grad=True)
        W, b = torch.empty(3, 429, requires_grad=True), torch.empty(3, 429, requires_grad=True)
        out = torch.ops.rechub_hip.cross_network(z, W, b)
        assert out.shape == (64, 429) and out.requires_grad
        out.sum().backward()  # the autograd formula traces too
        assert z.grad.shape == (64, 429) and W.grad.shape == (3, 429) and b.grad.shape == (3, 429)
        assert torch.ops.rechub_hip.dice(torch.empty(100, 36), torch.empty(1), 1e-9).shape == (100, 36)
        # the functional form of the fused gather + FM + LR (shared table in fields 1 and 3) and its autograd formula
        tabs = [torch.empty(7, 16, requires_grad=True), torch.empty(300, 16, requires_grad=True),
                torch.empty(41, 16, requires_grad=True)]
        tabs.append(tabs[1])
        lw, lb = torch.empty(1, 64, requires_grad=True), torch.empty(1, requires_grad=True)
        o, f, l, S = torch.ops.rechub_hip.embedding_fm_lr(tabs, torch.empty(53, 4, dtype=torch.int64), torch.empty(53, 3), lw, lb)
        assert o.shape == (53, 67) and f.shape == (53, 1) and l.shape == (53, 1) and S.shape == (53, 16)
        (o.sum() + f.sum() + l.sum()).backward()
        assert tabs[1].grad.shape == (300, 16) and lw.grad.shape == (1, 64) and lb.grad.shape == (1,)
        assert torch.ops.rechub_hip.adam_step_(torch.empty(8, 4), torch.empty(8, 4), torch.empty(8, 4), torch.empty(8, 4), 1,
                                               1e-3, 0.9, 0.999, 1e-8, 0.0) is None
        # round 4: the rest of SURVEY 8(b)'s list -- CrossNetV2, CrossNetMix, the DIN attention kernels, the masked
        # embedding bag (sum / mean / concat) and the in-batch sampler -- with their autograd formulas
        x2 = torch.empty(64, 429, requires_grad=True)
        W2, b2 = torch.empty(3, 429, 429, requires_grad=True), torch.empty(3, 429, requires_grad=True)
        o2 = torch.ops.rechub_hip.cross_net_v2(x2, W2, b2)
        o2.sum().backward()
        assert o2.shape == (64, 429) and W2.grad.shape == (3, 429, 429) and b2.grad.shape == (3, 429) and x2.grad.shape == (64, 429)
        U = torch.empty(3, 4, 429, 32, requires_grad=True)
        V = torch.empty(3, 4, 429, 32, requires_grad=True)
        C = torch.empty(3, 4, 32, 32, requires_grad=True)
        bm, gt = torch.empty(3, 429, requires_grad=True), torch.empty(4, 429, requires_grad=True)
        xm = torch.empty(64, 429, requires_grad=True)
        om = torch.ops.rechub_hip.cross_net_mix(xm, U, V, C, bm, gt)
        om.sum().backward()
        assert om.shape == (64, 429) and U.grad.shape == U.shape and C.grad.shape == C.shape and gt.grad.shape == (4, 429)
        hist, tgt = torch.empty(8, 100, 16, requires_grad=True), torch.empty(8, 16, requires_grad=True)
        ai = torch.ops.rechub_hip.din_attention_input(hist, tgt)
        aw = torch.empty(8, 100, requires_grad=True)
        ap = torch.ops.rechub_hip.din_attention_pool(aw, hist)
        assert ai.shape == (800, 64) and ap.shape == (8, 16)
        (ai.sum() + ap.sum()).backward()
        assert hist.grad.shape == (8, 100, 16) and tgt.grad.shape == (8, 16) and aw.grad.shape == (8, 100)
        tab = torch.empty(500, 16, requires_grad=True)
        seq = torch.empty(8, 50, dtype=torch.int64)
        assert torch.ops.rechub_hip.embedding_bag_masked(tab, seq, 0, "mean").shape == (8, 16)
        bag = torch.ops.rechub_hip.embedding_bag_masked(tab, seq, -1, "concat")
        assert bag.shape == (8, 50, 16)
        bag.sum().backward()
        assert tab.grad.shape == (500, 16)
        neg = torch.ops.rechub_hip.inbatch_negative_sample(torch.empty(64, 64), 20, False, 2022, 0)
        assert neg.shape == (64, 20) and neg.dtype == torch.int64
    with pytest.raises(RuntimeError, match="HIP device"):
        torch.ops.rechub_hip.fm(torch.zeros(2, 3, 4), True)
    with pytest.raises(RuntimeError, match="HIP device"):
        torch.ops.rechub_hip.cross_net_v2(torch.zeros(2, 4), torch.zeros(1, 4, 4), torch.zeros(1, 4))
    schema = str(torch.ops.rechub_hip.cross_network.default._schema)
    assert schema.startswith("rechub_hip::cross_network(Tensor x, Tensor W, Tensor b) -> Tensor")


def test_mlp_head_after_finds_only_a_lone_output_layer_behind_inactive_dropouts(monkeypatch):
    """MLP.head_after (the host-side pattern match for BatchNorm1d -> Dice -> Linear(C, 1), ops.bn_dice_head): the output
    Linear(width, 1) is taken only when nothing but inactive Dropouts stand between the Dice and it and it ends the stack."""
    import torch
    from torch_rechub_amd.basic.layers import MLP
    mlp = MLP(64, dims=[32, 16], activation="dice")  # reference default dropout = 0 (layers.py:269)
    mods = list(mlp.mlp)
    assert [type(m).__name__ for m in mods] == ["Linear", "BatchNorm1d", "Dice", "Dropout"] * 2 + ["Linear"]
    assert MLP.head_after(mods, 7, 16, mods[5], mods[6]) is mods[8]   # behind the LAST hidden block
    assert MLP.head_after(mods, 3, 32, mods[1], mods[2]) is None      # another hidden block follows the first
    assert MLP.head_after(mods, 7, 32, mods[5], mods[6]) is None      # width of the producer does not match
    active = list(MLP(64, dims=[16], activation="dice", dropout=0.5).mlp)
    assert MLP.head_after(active, 3, 16, active[1], active[2]) is None   # training-mode dropout with p > 0 stays a kernel
    for m in active:
        m.eval()
    assert MLP.head_after(active, 3, 16, active[1], active[2]) is active[4]
    no_out = list(MLP(64, output_layer=False, dims=[16], activation="dice").mlp)
    assert MLP.head_after(no_out, 3, 16, no_out[1], no_out[2]) is None
    wide = list(MLP(64, dims=[16], activation="dice").mlp)
    wide[-1] = torch.nn.Linear(16, 2)
    assert MLP.head_after(wide, 3, 16, wide[1], wide[2]) is None
    from torch_rechub_amd import ops as _ops
    monkeypatch.setattr(_ops, "FUSE_DICE_HEAD", False)
    assert MLP.head_after(mods, 7, 16, mods[5], mods[6]) is None


def test_ab_switch_parsing(monkeypatch):
    """RECHUB_AB="name=0,other=1": ONE variable for the same-box A/B twins of benchmarks (torch_rechub_amd/_lib.py)."""
    from torch_rechub_amd import _lib
    monkeypatch.delenv("RECHUB_AB", raising=False)
    assert _lib.ab("chain") is True and _lib.ab("chain", default=False) is False
    monkeypatch.setenv("RECHUB_AB", "chain=0, ahead=off,lookahead=1,,junk")
    assert _lib.ab("chain") is False and _lib.ab("ahead") is False and _lib.ab("lookahead") is True
    assert _lib.ab("latepack") is True  # not named: the default
    assert _lib.ab("hain") is True      # exact names only


def test_bench_lazy_k_default_follows_the_trainers_rule(monkeypatch):
    """bench.py --lazy-k: default = what CTRTrainer(lazy_k=None) takes (128 up to 8192 samples per step, 64 beyond); an explicit
    value is kept for every batch size of the sweep."""
    import sys
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.lazy_k == 128 and a.lazy_k_explicit is False
    monkeypatch.setattr(sys, "argv", ["bench.py", "--batch", "32768"])
    a = bench.parse()
    assert a.lazy_k == 64 and a.lazy_k_explicit is False
    monkeypatch.setattr(sys, "argv", ["bench.py", "--batch", "32768", "--lazy-k", "96"])
    a = bench.parse()
    assert a.lazy_k == 96 and a.lazy_k_explicit is True


def test_step_form_candidates_carry_form_grid_and_hold_back():
    """CTRTrainer.TUNE_CANDIDATES: (form, residency cap, hold-back ns) triples; bench.py reports all three."""
    from torch_rechub_amd.trainers import CTRTrainer
    cands = CTRTrainer.TUNE_CANDIDATES
    assert all(len(c) == 3 for c in cands)
    assert {c[0] for c in cands} == {"deferred", "inline"}
    assert [c for c in cands if c[0] == "inline"] == [("inline", 0, 0)]
    assert all(c[1] in (256, 512) and 20000 <= c[2] <= 200000 for c in cands if c[0] == "deferred")
    assert len({c[:2] for c in cands}) < len(cands)  # at least one (form, grid) comes with two hold-backs



def test_environment_switches_stay_few_and_documented():
    """at most twelve RECHUB_* switches in the product, each named in INTEGRATION.md; every A/B twin of
    RECHUB_AB that the code reads is listed there too."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "torch_rechub_amd", "**", "*.py"), recursive=True) + [os.path.join(root, "bench.py")]
    text = "\n".join(open(f).read() for f in files)
    names = set(re.findall(r"RECHUB_[A-Z_]+", text))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    assert len(names) <= 12, sorted(names)
    assert all(n in doc for n in names), sorted(n for n in names if n not in doc)
    twins = set(re.findall(r"_lib\.ab\(\"([a-z]+)\"", text)) | set(re.findall(r"[^_]ab\(\"([a-z]+)\"", text))
    assert twins and all(t + "=0" in doc for t in twins), sorted(t for t in twins if t + "=0" not in doc)


def test_short_sweep_rule_of_the_row_sharded_step():
    """optim.TableAdam.prefer_inline_for_short_sweeps (host logic only; the GPU tests run both forms): a rank whose lazily stepped
    tables hold fewer than SHORT_SWEEP_ELEMENTS elements takes the in-line window sweep and, when lazy_k was not given, lazy_k 64;
    a full-size table keeps the deferred sweep; nothing changes once a step has run or a sweep is pending."""
    import types

    from torch_rechub_amd.optim import TableAdam

    def table(rows, d=16):
        return types.SimpleNamespace(shape=(rows, d), numel=lambda: rows * d)

    def opt(tables, lazy_k=128, **kw):
        o = types.SimpleNamespace(lazy_k=lazy_k, _tables=tables, lazy_small_rows=4096, _dense_by_volume=set(), _host_step=0,
                                  _sweep_pending=False, _sweep_inflight=False, overlap_sweep=True, _lazy_groups="built",
                                  SHORT_SWEEP_ELEMENTS=TableAdam.SHORT_SWEEP_ELEMENTS)
        o.table_k = lambda p: TableAdam.table_k(o, p)
        o.__dict__.update(kw)
        return o

    criteo = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306, 10, 5652,
              2173, 4, 7046547, 18, 15, 286181, 105, 142572]
    for world, inline in ((1, False), (2, False), (4, True), (8, True)):
        o = opt([table(-(-v // world)) for v in criteo])
        assert TableAdam.prefer_inline_for_short_sweeps(o, auto_k=True) is inline, world
        assert o.overlap_sweep is (not inline) and o.lazy_k == (64 if inline else 128)
        assert (o._lazy_groups is None) == inline  # the window layout is rebuilt for the new lazy_k
    o = opt([table(-(-v // 8)) for v in criteo], lazy_k=32)
    assert TableAdam.prefer_inline_for_short_sweeps(o, auto_k=False) and o.lazy_k == 32 and not o.overlap_sweep  # explicit K kept
    o = opt([table(-(-v // 8)) for v in criteo])
    assert TableAdam.prefer_inline_for_short_sweeps(o, auto_k=False) and o.lazy_k == 128
    for late in (dict(_host_step=3), dict(_sweep_pending=True), dict(_sweep_inflight=True), dict(lazy_k=0)):
        o = opt([table(1000000)], **late)
        assert TableAdam.prefer_inline_for_short_sweeps(o, auto_k=True) is False and o.overlap_sweep
    # tables stepped densely (<= lazy_small_rows rows) do not count: their pass is not a window sweep
    o = opt([table(4096, 64)] * 1000 + [table(13_000_000)])
    assert TableAdam.prefer_inline_for_short_sweeps(o, auto_k=True) is False

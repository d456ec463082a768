"""Patch an installed ``torch_rechub`` in place so that its model zoo and trainers run on the MI355X hot path
(INTEGRATION.md section B — the binding a maintainer of the reference would add as ``torch_rechub/_hip.py``).

The reference has no plugin layer: its models bind their layers BY NAME at import time
(``from ...basic.layers import FM, LR, MLP, EmbeddingLayer``, torch_rechub/models/ranking/deepfm.py:10), so swapping
the implementation means rebinding those names in ``torch_rechub.basic.layers`` AND in every already-imported
``torch_rechub.*`` module that holds a reference to the original class.  ``enable()`` does exactly that, at three levels:

  layers    EmbeddingLayer, InputMask, the pooling layers, LR, MLP, FM, CrossNetwork, CrossNetV2, CrossNetMix, Dice and
            the Feature descriptors -> an UNMODIFIED reference model class (e.g. the source of
            torch_rechub/models/ranking/deepfm.py as it stands) then builds and runs on the HIP layers;
  models    DeepFM, WideDeep, DCN, DCNv2, DIN, ... -> the fused forwards (one gather launch emits the MLP input, FM and LR);
  trainers  CTRTrainer / MatchTrainer / MTLTrainer -> TableAdam, device-resident loader, hipGraph step, RCCL data parallel.

Constructor signatures, attribute names and ``state_dict`` keys are the reference's at every level
(tests/test_integration_patch.py).  ``disable()`` restores the original bindings.
"""
import importlib
import sys

_LAYERS = ("EmbeddingLayer", "InputMask", "SumPooling", "AveragePooling", "ConcatPooling", "LR", "MLP", "FM",
           "CrossNetwork", "CrossNetV2", "CrossNetMix", "SENETLayer", "BiLinearInteractionLayer", "InteractingLayer",
           "CrossLayer")
_FEATURES = ("DenseFeature", "SparseFeature", "SequenceFeature")
_ACTIVATIONS = ("Dice", "activation_layer")
_MODELS = {"ranking": ("DeepFM", "WideDeep", "DCN", "DCNv2", "DIN", "DIEN", "BST", "AFM", "AutoInt", "EDCN", "FiBiNet"),
           "matching": ("DSSM",),
           "multi_task": ("SharedBottom", "ESMM", "MMOE", "PLE", "AITM")}
_TRAINERS = ("CTRTrainer", "MatchTrainer", "MTLTrainer")

_undo = []  # (module, attribute, original object)


def _rebind_everywhere(root, original, replacement):
    """Point every attribute of every imported ``<root>.*`` module that IS ``original`` at ``replacement``."""
    prefix = root + "."
    for modname, mod in list(sys.modules.items()):
        if mod is None or not (modname == root or modname.startswith(prefix)):
            continue
        for attr, val in list(vars(mod).items()):
            if val is original:
                setattr(mod, attr, replacement)
                _undo.append((mod, attr, original))


def _swap(root, ref_modname, amd_module, names):
    try:
        ref_mod = importlib.import_module(ref_modname)
    except ImportError:
        return []
    done = []
    for name in names:
        ours = getattr(amd_module, name, None)
        theirs = getattr(ref_mod, name, None)
        if ours is None or theirs is None or ours is theirs:
            continue
        _rebind_everywhere(root, theirs, ours)
        if getattr(ref_mod, name) is not ours:  # not reached by identity (e.g. re-exported under another object)
            setattr(ref_mod, name, ours)
            _undo.append((ref_mod, name, theirs))
        done.append(f"{ref_modname}.{name}")
    return done


def enable(layers=True, models=True, trainers=True, package="torch_rechub"):
    """Rebind the reference package's hot-path classes to the HIP implementations.  Returns the list of patched names.

    Call it after ``import torch_rechub`` (and its ``models`` / ``trainers`` sub-packages, if the level is wanted) and
    before models are built.  Idempotent."""
    from . import basic, trainers as amd_trainers
    from .basic import activation, features, layers as amd_layers
    importlib.import_module(package)
    done = []
    if layers:
        done += _swap(package, f"{package}.basic.features", features, _FEATURES)
        done += _swap(package, f"{package}.basic.activation", activation, _ACTIVATIONS)
        done += _swap(package, f"{package}.basic.layers", amd_layers, _LAYERS)
    if models:
        for sub, names in _MODELS.items():
            amd_sub = importlib.import_module(f"{__package__}.models.{sub}")
            done += _swap(package, f"{package}.models.{sub}", amd_sub, names)
    if trainers:
        done += _swap(package, f"{package}.trainers", amd_trainers, _TRAINERS)
    del basic
    return done


def disable():
    """Undo every rebinding made by ``enable`` (latest first)."""
    while _undo:
        mod, attr, original = _undo.pop()
        setattr(mod, attr, original)

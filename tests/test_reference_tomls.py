"""Every TOML the reference ships (recipes/dns_interspeech_2020/*/*.toml) names a model class and its keyword
arguments; `initialize_module(path, args)` (audio_zen/utils.py:70-105) builds the class from them.  The drop-in
contract: each `[model.args]` table instantiates the corresponding fullsubnet_amd class unchanged, with the
reference's parameter names, shapes and counts.  Needs the reference checkout to read the TOMLs (skipped on the GPU
box, where /root/reference does not exist).  CPU only."""
import glob
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"
RECIPE = os.path.join(REF, "recipes", "dns_interspeech_2020")
TOMLS = sorted(glob.glob(os.path.join(RECIPE, "*", "*.toml")))

# [model].path of the TOML (relative to its recipe directory or absolute) -> the class here
OURS = {"fullsubnet": "fullsubnet_amd.model.Model", "fast_fullsubnet": "fullsubnet_amd.fast_fullsubnet.Model",
        "fullband_baseline": "fullsubnet_amd.fullband_baseline.Model"}
# the reference's own stale file: its [model.args] (n_freqs, use_offline_laplace_norm ...) are not the keywords of
# fullband_baseline/model.py:9-18 - the reference class rejects them too
STALE = {os.path.join("fullband_baseline", "inference.toml")}

pytestmark = pytest.mark.skipif(not os.path.isdir(RECIPE), reason="the reference checkout is not on this box")


def _load(dotted):
    mod, cls = dotted.rsplit(".", 1)
    return getattr(__import__(mod, fromlist=[cls]), cls)


def _reference_class(recipe):
    sys.modules.setdefault("librosa", types.ModuleType("librosa"))  # feature.py:3, used by load_wav only
    if recipe == "fast_fullsubnet":  # torchaudio / torchinfo are not in the image: the stubs of make_golden_family.py
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
        import make_golden_family  # noqa: F401  (installs the stubs and the reference paths)
    for p in (REF, RECIPE):
        if p not in sys.path:
            sys.path.insert(0, p)
    return _load(f"{recipe}.model.Model")


def test_the_reference_ships_the_tomls_this_file_expects():
    assert len(TOMLS) == 9, TOMLS


@pytest.mark.parametrize("path", TOMLS, ids=[os.path.relpath(p, RECIPE) for p in TOMLS])
def test_shipped_toml_instantiates_the_drop_in_class(path):
    import tomli
    rel = os.path.relpath(path, RECIPE)
    recipe = rel.split(os.sep)[0]
    with open(path, "rb") as f:
        cfg = tomli.load(f)
    args = cfg["model"]["args"]
    assert cfg["model"]["path"].endswith("model.Model")
    ours_cls = _load(OURS[recipe])
    ref_cls = _reference_class(recipe)
    if rel in STALE:
        with pytest.raises(TypeError):
            ref_cls(**args)
        with pytest.raises(TypeError):
            ours_cls(**args)
        return
    torch.manual_seed(0)
    ours = ours_cls(**args)
    ref = ref_cls(**args)
    want = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    got = {k: tuple(v.shape) for k, v in ours.state_dict().items()}
    assert got == want  # names and shapes: strict load_state_dict works in both directions
    assert sum(p.numel() for p in ours.parameters()) == sum(p.numel() for p in ref.parameters())
    ours.load_state_dict(ref.state_dict(), strict=True)
    ref.load_state_dict(ours.state_dict(), strict=True)
    # the inferencer / trainer sections name classes as well: their mirrors exist with the same entry points
    if "inferencer" in cfg:  # */inference*.toml:8-13: the inference mode is a method name of the Inferencer
        from fullsubnet_amd.inferencer import Inferencer
        assert callable(getattr(Inferencer, cfg["inferencer"]["type"]))
    if "trainer" in cfg:
        from fullsubnet_amd.trainer import Trainer
        assert hasattr(Trainer, "_train_epoch") and hasattr(Trainer, "_validation_epoch")

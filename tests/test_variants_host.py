"""Host logic of round 5's variant fixtures (no GPU, no reference tree): the table of the other RAD-NeRF configurations the reference ships, the
audio-driven sequence generator, the coin helper of the head-aware parity harness."""
import random

import numpy as np

from geneface_amd import hparams as HP
from geneface_amd import synthetic as S


def test_variant_table_overrides_only_what_the_yaml_files_state():
    base = HP.may_hparams(True)
    assert HP.variant_hparams("default", True) == base
    for name, (over, files) in HP.VARIANTS.items():
        hp = HP.variant_hparams(name, True)
        assert files and all(f.startswith("egs/") and f.endswith(".yaml") for f in files)
        for k, v in over.items():
            assert hp[k] == v
        changed = {k for k in hp if hp[k] != base.get(k)}
        assert changed <= set(over) | {"head_model_dir"}, (name, changed)
    assert HP.variant_hparams("hash")["grid_type"] == "hashgrid" and HP.variant_hparams("hash_smoothstep")["grid_interpolation_type"] == "smoothstep"
    assert HP.variant_hparams("head_aware")["torso_head_aware"] is True
    a = HP.variant_hparams("audio")
    assert (a["cond_type"], a["cond_win_size"], a["smo_win_size"], a["individual_embedding_num"]) == ("esperanto", 16, 8, 10000)
    assert HP.may_hparams(True) == base                                   # the table hands out copies


def test_audio_sequence_has_the_dataset_layout():
    """cond_wins of the audio-driven config: per frame the smo_win = 8 frames around it (get_audio_features att_mode 2: left = i - 4,
    right = i + 4, zero padded at the ends; modules/radnerfs/utils.py:85-101) of [16, 44] feature windows."""
    hp = HP.variant_hparams("audio", True)
    seq = S.make_sequence(12, 32, 32, hp, seed=1000)
    w = seq["cond_wins"]
    assert w.shape == (12, 8, 16, 44) and w.dtype == np.float32
    feats = S.make_audio_features(12, 44, 16, seed=13 + 1000)
    assert feats.shape == (12, 16, 44)
    for i in (0, 3, 5, 11):
        for k in range(8):
            j = i - 4 + k
            want = feats[j] if 0 <= j < 12 else np.zeros((16, 44), np.float32)
            assert np.array_equal(w[i, k], want), (i, k)
    assert np.array_equal(S.make_sequence(12, 32, 32, hp, seed=1000)["cond_wins"], w)            # seeded
    assert not np.array_equal(S.make_sequence(12, 32, 32, hp, seed=0)["cond_wins"], w)
    assert abs(float(feats[2:10].std()) - 1.0) < 0.2                                             # unit-variance AR(1) stream
    # the landmark-driven default is untouched by the audio branch
    d = S.make_sequence(12, 32, 32, HP.may_hparams(True))
    assert d["cond_wins"].shape == (12, 5, 1, 204)
    sd = S.make_state_dict(hp, True, seed=1000)
    assert sd["cond_prenet.encoder_conv.0.weight"].shape == (32, 44, 3) and sd["cond_att_net.attentionNet.0.weight"].shape == (8, 8)
    assert sd["individual_embeddings"].shape == (10000, 4)


def test_coin_helper_previews_the_next_draw():
    import bench
    assert bench.coin({"torso_head_aware": False}, 5) is False
    hp = {"torso_head_aware": True}
    outcomes = set()
    for seed in range(40):
        c = bench.coin(hp, seed)
        assert (random.random() < 0.5) == c          # the very next draw is the one it looked at
        outcomes.add(c)
    assert outcomes == {True, False}


def test_half_on_the_model_or_its_holder_selects_the_f16_tier_and_keeps_fp32_masters():
    """inference/nerfs/radnerf_gui.py:604-605: `nerf_task = nerf_task.half()` when `amp` is set.  The reference then computes in half from half
    parameters; here `.half()` selects `render_precision = "fast"` (the f16 tier) and leaves the fp32 masters alone -- the C ABI takes fp32
    tables and weights.  `.float()` restores the previous tier; conversions without a tier are refused, device moves pass through."""
    import pytest
    import torch
    from geneface_amd.radnerf_torso import RADNeRFTorso
    model = RADNeRFTorso(HP.may_hparams(True))
    before = {k: v.clone() for k, v in model.state_dict().items()}
    assert model.render_precision == "fp32" and model.half() is model
    assert model.render_precision == "fast"
    assert all(p.dtype == torch.float32 for p in model.parameters()) and model.density_grid.dtype == torch.float32
    assert all(torch.equal(v, before[k]) for k, v in model.state_dict().items())
    model.float()
    assert model.render_precision == "fp32"

    class Task(torch.nn.Module):          # what the GUI halves is the task that holds the model
        def __init__(self, m):
            super().__init__()
            self.model, self.criterion = m, torch.nn.Linear(2, 2)

    task = Task(model).half()
    assert model.render_precision == "fast" and task.criterion.weight.dtype == torch.float16
    assert next(model.parameters()).dtype == torch.float32
    task.float()
    assert model.render_precision == "fp32" and task.criterion.weight.dtype == torch.float32
    model.render_precision = "split"
    model.to(torch.float16)
    assert model.render_precision == "fast"
    model.to("cpu")                                                       # a device move says nothing about the tier
    assert model.render_precision == "fast"
    model.float()
    assert model.render_precision == "split"
    for convert in (model.double, model.bfloat16):
        with pytest.raises(NotImplementedError):
            convert()


def test_model_state_for_copies_leaves_the_packed_device_state_behind_and_versions_of_inference_tensors():
    """Host halves of two GPU tests: NeRFRenderer.__getstate__ drops the fused path's per-object caches (so deepcopy / pickle work after a
    render), and fused._ver answers for tensors without a version counter with a number it never gave before (so nothing keyed on it is reused)."""
    import copy
    import pickle
    import torch
    from geneface_amd import fused
    from geneface_amd.radnerf import RADNeRF
    model = RADNeRF(HP.may_hparams(False))
    object.__setattr__(model, "_fused_state", object())          # what fused.get_state hangs on the module (ctypes inside: not picklable)
    object.__setattr__(model, "_torso_occ_any", (None, True, None, None))
    twin = copy.deepcopy(model)
    assert not hasattr(twin, "_fused_state") and not hasattr(twin, "_torso_occ_any") and hasattr(model, "_fused_state")
    again = pickle.loads(pickle.dumps(model))
    assert not hasattr(again, "_fused_state")
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), again.state_dict().values()))
    t = torch.zeros(3)
    assert fused._ver(t) == t._version == 0
    t.add_(1)
    assert fused._ver(t) == 1
    with torch.inference_mode():
        u = torch.zeros(3)
    a, b = fused._ver(u), fused._ver(u)
    assert a < 0 and b < 0 and a != b

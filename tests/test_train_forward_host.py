"""Host logic of the train-mode forward (pantomatrix_amd/training.py) on CPU stand-ins of the kernels: the order in which the
dropout masks are consumed, the BatchNorm bookkeeping, the un-folded WavEncoder — against oracle/emage_train_oracle.py, which
is pinned to the reference's training step (tests/test_train_oracle.py)."""
import pytest
import torch

import common
import fake_ops
import train_common as tc
from oracle import emage_train_oracle as tro
from pantomatrix_amd import training


def test_recorder_is_transparent():
    """Recording the masks does not change the oracle: same outputs as the plain run with the same seed."""
    (audio, spk, motion, mask), out, masks, ns = tc.oracle_forward(seed=3)
    import common as c
    from pantomatrix_amd import synthetic
    from pantomatrix_amd.configuration_emage_audio import EmageAudioConfig
    sd = synthetic.audio_model_state(EmageAudioConfig(**c.cfg_dicts()[0]), 0)
    torch.manual_seed(3)
    with torch.no_grad():
        ref = tro.forward_train(sd, audio, spk, motion, mask)
    assert len(masks) == training.dropout_mask_count()
    for k in ref:
        assert torch.equal(out[k], ref[k]), k


@pytest.mark.parametrize("precision,use_audio", [("fp32", True), ("f16x3", True), ("fp32", False)])
def test_train_forward_host_logic(precision, use_audio):
    (audio, spk, motion, mask), ref, masks, ref_stats = tc.oracle_forward(seed=7, use_audio=use_audio)
    model, _ = common.product_models(precision=precision)
    fwd = training.TrainForward(model)
    with fake_ops.installed(), torch.no_grad():
        out, stats = fwd(audio, spk, motion, mask, masks, use_audio=use_audio)
        assert "bn_stats" in fake_ops.CALLS and "attention_dropout" in fake_ops.CALLS and "attention" not in fake_ops.CALLS
    for k in ref:
        err = float((out[k] - ref[k]).abs().max())
        assert err < 2e-4, (k, err)
    assert set(stats) == set(ref_stats)
    for k, v in ref_stats.items():
        if k.endswith("num_batches_tracked"):
            assert int(stats[k]) == int(v)
        else:
            assert float((stats[k] - v).abs().max()) < 1e-5 * max(1.0, float(v.abs().max())), k


def test_running_stats_chain_and_mask_checks():
    """Two forwards of one step: the second starts from the first one's running buffers (three forwards per step in
    train_emage_audio.py:138-172); wrong mask counts / shapes are refused."""
    (audio, spk, motion, mask), _, masks1, ns = tc.oracle_forward(seed=1)
    _, ref2, masks2, ns = tc.oracle_forward(seed=2, new_stats=ns)
    model, _ = common.product_models(precision="fp32")
    fwd = training.TrainForward(model)
    with fake_ops.installed(), torch.no_grad():
        _, stats = fwd(audio, spk, motion, mask, masks1)
        out2, stats = fwd(audio, spk, motion, mask, masks2, new_stats=stats)
        with pytest.raises(RuntimeError, match="masks given"):
            fwd(audio, spk, motion, mask, masks1[:-1])
        with pytest.raises(RuntimeError, match="the reference draws"):
            fwd(audio, spk, motion, mask, masks1[1:] + masks1[:1])
    for k in ref2:
        assert float((out2[k] - ref2[k]).abs().max()) < 2e-4, k
    key = "audio_encoder_face.feat_extractor.0.bn1"
    assert int(stats[key + ".num_batches_tracked"]) == int(ns[key + ".num_batches_tracked"])
    assert float((stats[key + ".running_var"] - ns[key + ".running_var"]).abs().max()) < 1e-5


def test_step_losses_host_logic(golden_dir):
    """training.step_losses (targets through the product VQ model, three forwards, the two loss kernels) on the CPU stand-ins:
    the oracle's losses for the same draws — which are the REAL reference's (tests/golden/train_step_b2.npz)."""
    import os
    import numpy as np
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    batch, ref, masks, random_mask, ref_stats = tc.oracle_step(int(g["seed"]), int(g["iteration"]))
    model, vq = common.product_models(precision="fp32")
    fwd = training.TrainForward(model)
    with fake_ops.installed(), torch.no_grad():
        got, stats = training.step_losses(fwd, vq, batch, int(g["iteration"]), masks, random_mask)
        assert fake_ops.CALLS.count("mse_loss") == 12 and fake_ops.CALLS.count("nll_loss") == 12
    for k in ("rec_seed", "cls_seed", "rec_audio", "cls_audio", "rec_mask", "cls_mask", "all"):
        assert abs(got[k] - ref[k]) < 2e-4 * max(1.0, abs(ref[k])), (k, got[k], ref[k])
        assert abs(got[k] - float(g["loss_" + k])) < 2e-4 * max(1.0, abs(float(g["loss_" + k]))), k
    key = "audio_encoder_body.feat_extractor.5.downsample.1.running_mean"
    assert float((stats[key] - ref_stats[key]).abs().max()) < 1e-5


def _oracle_grads(seed, use_audio=True):
    """Autograd gradients of rec_loss + cls_loss of ONE train-mode forward of the oracle (same draws as oracle_forward(seed))."""
    from pantomatrix_amd import synthetic
    from pantomatrix_amd.configuration_emage_audio import EmageAudioConfig
    cfg = EmageAudioConfig(**common.cfg_dicts()[0])
    sd = synthetic.audio_model_state(cfg, 0)
    keys = tro.trainable_keys(sd)
    work = dict(sd)
    for k in keys:
        work[k] = sd[k].detach().clone().requires_grad_(True)
    audio, spk, motion, mask = common.window_inputs(2)
    g = torch.Generator().manual_seed(99)
    latent = {q: torch.randn(2, motion.shape[1], 256, generator=g) for q in ("face", "upper", "hands", "lower")}
    index = {q: torch.randint(0, 256, (2, motion.shape[1]), generator=g) for q in ("face", "upper", "hands", "lower")}
    masks = []
    torch.manual_seed(seed)
    with tc.recorded_masks(masks):
        pred = tro.forward_train(work, audio, spk, motion, mask, use_audio=use_audio)
    loss = tro.rec_loss(pred, latent, cfg) + tro.cls_loss(pred, index, cfg)
    loss.backward()
    grads = {k: work[k].grad for k in keys if work[k].grad is not None}
    return (audio, spk, motion, mask), masks, index, latent, grads


FRONT_END = ("audio_encoder_face.", "audio_encoder_body.", "motion_encoder.", "mask_embedding")


@pytest.mark.parametrize("use_audio", [True, False])
def test_backward_host_logic(use_audio):
    """The tape-driven backward (training.TrainForward.backward) on the CPU stand-ins against torch autograd through the
    training oracle: every parameter behind the convolutional front ends — the 16 transformer layers, projections, MLP heads,
    speaker embeddings."""
    (audio, spk, motion, mask), masks, index, latent, ref = _oracle_grads(seed=4, use_audio=use_audio)
    model, _ = common.product_models(precision="fp32")
    fwd = training.TrainForward(model)
    with fake_ops.installed(), torch.no_grad():
        fwd(audio, spk, motion, mask, masks, use_audio=use_audio, tape=True)
        grads = fwd.backward(index, latent)
    covered = [k for k in ref if not k.startswith(FRONT_END)]
    assert len(covered) > 300
    gmax = max(float(ref[k].abs().max()) for k in covered)
    missing = [k for k in covered if k not in grads and float(ref[k].abs().max()) > 1e-7 * gmax]
    assert not missing, missing[:8]
    for k in covered:
        if k not in grads:
            continue
        err = float((grads[k] - ref[k]).abs().max())
        assert err <= 2e-4 * float(ref[k].abs().max()) + 1e-6 * gmax, (k, err, float(ref[k].abs().max()))
    assert not [k for k in grads if k.startswith(FRONT_END)]

"""Host logic of the DisCo / CaMN product classes (weight packing incl. the per-unit gate interleave, column-block
operands, seed-motion length reconciliation, step schedule of the bidirectional LSTM) on CPU, with every C-ABI call
replaced by its torch restatement (tests/fake_ops.py), against the oracle and the REFERENCE's golden outputs.
The kernels themselves are compared on the MI355X in tests/test_lstm_gpu.py."""
import os

import numpy as np
import pytest
import torch

import fake_ops
from oracle import lstm_models_oracle as lo
from test_lstm_models_oracle import CFG, inputs, weights, run_oracle


def product(kind, precision="f16x3", device=None):
    from pantomatrix_amd import modeling_lstm_audio as L
    cls, ccls = (L.DiscoAudioModel, L.DiscoAudioConfig) if kind == "disco" else (L.CamnAudioModel, L.CamnAudioConfig)
    m = cls(ccls(**CFG)).set_precision(precision)
    m.load_state_dict(weights(kind))
    return m.to(device) if device else m


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
@pytest.mark.parametrize("kind", ["disco", "camn"])
def test_forward_matches_golden(golden_dir, kind, precision):
    g = np.load(os.path.join(golden_dir, "lstm_models.npz"))
    model = product(kind, precision)
    for tag, wsm in (("plain", False), ("seeded", True)):
        audio, spk, motion = inputs(with_seed_motion=wsm)
        with fake_ops.installed(), torch.no_grad():
            out = model(audio, spk, seed_frames=CFG["seed_frames"], seed_motion=motion)
        np.testing.assert_allclose(out["motion"].reshape(2, -1, 258).numpy(), g[f"{kind}_{tag}_motion"], atol=5e-5, rtol=0)
        np.testing.assert_allclose(out["motion_axis_angle"].numpy(), g[f"{kind}_{tag}_axis_angle"], atol=1e-3, rtol=0)
        assert out["motion_axis_angle"].shape == (2, out["motion"].shape[1], 165)
        if kind == "disco":
            ref = run_oracle(kind, weights(kind), audio, spk, motion)
            for k in ("audio_fea_c", "audio_fea_r"):
                assert float((out[k] - ref[k]).abs().max()) < 2e-5, k
        else:
            assert out["motion"].shape[2:] == (43, 6)
    t = out["motion"].shape[1]
    n_lstm = (1 if kind == "disco" else 2) * CFG["n_layer"]
    if precision == "f16x3":        # one persistent launch per LSTM layer
        assert fake_ops.CALLS.count("lstm_layer") == n_lstm and "lstm_step_pair" not in fake_ops.CALLS
    else:                           # exact-fp32 mode: one launch per layer and step (both directions)
        assert fake_ops.CALLS.count("lstm_step_pair") == n_lstm * t and "lstm_layer" not in fake_ops.CALLS


def test_per_step_recurrence_switch():
    """`persistent_lstm = False` routes the f16x3 recurrence through one paired launch per step: the same result."""
    audio, spk, motion = inputs(with_seed_motion=True)
    outs = []
    for persistent in (True, False):
        model = product("disco")
        model.persistent_lstm = persistent
        with fake_ops.installed(), torch.no_grad():
            outs.append(model(audio, spk, seed_frames=CFG["seed_frames"], seed_motion=motion)["motion"])
            assert ("lstm_layer" in fake_ops.CALLS) == persistent and ("lstm_step_pair" in fake_ops.CALLS) != persistent
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("kind", ["disco", "camn"])
def test_seed_length_quirks(kind):
    """Seed motion longer than the audio frames is cut; shorter follows the reference's `cat(seed, seed[:, -diff:])`
    (D:238-242) — lengths that line up give the oracle's result, the others raise like the reference's torch.cat."""
    model = product(kind)
    sd = weights(kind)
    audio, spk, _ = inputs(frames=20)
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        t = lo.wav_encoder(sd, "audio_encoder", audio).shape[1]
    for t_m in (t + 5, (t + 2) // 2 + (t % 2), t - 1):
        motion = 0.3 * torch.randn(2, t_m, CFG["pose_dims"], generator=g)
        try:
            ref = run_oracle(kind, sd, audio, spk, motion)
        except RuntimeError:
            ref = None
        with fake_ops.installed(), torch.no_grad():
            if ref is None:
                with pytest.raises(RuntimeError):
                    model(audio, spk, seed_frames=4, seed_motion=motion)
            else:
                out = model(audio, spk, seed_frames=4, seed_motion=motion)
                assert float((out["motion"].reshape(ref["motion"].shape) - ref["motion"]).abs().max()) < 5e-5


def test_checkpoint_keys_and_no_cpu_fallback(tmp_path):
    model = product("camn")
    assert list(model.state_dict()) == list(lo.camn_spec(CFG))
    model.save_pretrained(str(tmp_path / "camn"))
    from pantomatrix_amd import modeling_lstm_audio as L
    again = L.CamnAudioModel.from_pretrained(str(tmp_path / "camn"))
    assert all(torch.equal(v, again.state_dict()[k]) for k, v in model.state_dict().items())
    audio, spk, _ = inputs()
    with pytest.raises(RuntimeError, match="MI355X"):
        model(audio, spk)
    with pytest.raises(ValueError):
        model.set_precision("bf16")

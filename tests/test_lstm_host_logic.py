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


from tools.workloads import lstm_product as product  # noqa: E402  (shared with bench.py)


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


def test_narrow_blocks_run_as_position_pairs():
    """The 32-channel WavEncoder blocks go through the 64-channel kernels on (L/2, 64) pair rows (`_Packed.conv_pairs`):
    the same features as the zero-padded route (`pair_convs = False`) up to fp32 summation order, no padded launches."""
    audio, spk, _ = inputs()
    audio = audio[:, :30000]                       # 6638 / 1104 / 1104 frames after blocks 0-2: even, like the 8.5 s and 28 s clips
    outs, calls = [], []
    for pair in (True, False):
        model = product("camn")
        model.pair_convs = pair
        with fake_ops.installed(), torch.no_grad():
            lens = model._wav_lengths(audio.shape[1])
            assert model._wav_pairs_ok(lens) and lens[:3] == [6638, 1104, 1104]
            outs.append(model(audio, spk, seed_frames=CFG["seed_frames"])["motion"])
            calls.append(list(fake_ops.CALLS))
    assert float((outs[0] - outs[1]).abs().max()) < 2e-5
    # pair route: 2 first-layer launches, conv2 of blocks 0-2 + conv1 of block 2 as slabs (4) + the wide blocks' 4 slabs
    assert calls[0].count("wav_conv_in") == 2 and calls[0].count("conv_slab") == 8
    assert calls[1].count("wav_conv_in") == 1 and calls[1].count("conv_slab") == 4
    odd = product("camn")                          # an odd frame count anywhere in the narrow blocks: the padded route, silently
    assert not odd._wav_pairs_ok(odd._wav_lengths(inputs()[0].shape[1]))


@pytest.mark.parametrize("kind", ["disco", "camn"])
def test_pair_rows_route_matches_reference_golden(golden_dir, kind):
    """The pair-rows route against the REAL reference (tests/golden/lstm_models_even.npz: a clip with even frame counts after
    WavEncoder blocks 0-2, seed motion longer than the audio frames)."""
    g = np.load(os.path.join(golden_dir, "lstm_models_even.npz"))
    audio, spk, motion = inputs(with_seed_motion=True)
    audio = audio[:, :30000]
    model = product(kind)
    with fake_ops.installed(), torch.no_grad():
        assert model._wav_pairs_ok(model._wav_lengths(audio.shape[1]))
        out = model(audio, spk, seed_frames=CFG["seed_frames"], seed_motion=motion)
        assert fake_ops.CALLS.count("wav_conv_in") == 2
    np.testing.assert_allclose(out["motion"].reshape(2, -1, 258).numpy(), g[f"{kind}_motion"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(out["motion_axis_angle"].numpy(), g[f"{kind}_axis_angle"], atol=1e-3, rtol=0)


def test_pair_weights_equal_the_convolution():
    """`conv_pairs` against torch's conv1d on random narrow rows: stride 1 (pad 7) and stride 6 (pad 0)."""
    import torch.nn.functional as F
    from pantomatrix_amd import modeling_emage_audio as M
    from pantomatrix_amd._lib import F32
    g = torch.Generator().manual_seed(4)
    pk = M._Packed({}, torch.device("cpu"), F32)
    x = torch.randn(3, 32, 40, generator=g)                                   # (seq, C, L), L even
    for stride, pad, cout in ((1, 7, 32), (6, 0, 32), (6, 0, 128)):
        w, b = 0.1 * torch.randn(cout, 32, 15, generator=g), torch.randn(cout, generator=g)
        pk.conv_pairs("t", w, b, torch.ones(cout), stride, pad)
        e = pk.w["t"]
        ref = F.conv1d(x, w, b, stride=stride, padding=pad)                  # (seq, cout, Lout)
        rows = x.permute(0, 2, 1).reshape(3, 20, 64)                          # pair rows
        wp = e["w"].view(e["n"], e["taps"], 64)
        lout = ref.shape[2] // (2 if stride == 1 else 1)
        got = torch.zeros(3, lout, e["n"])
        for p in range(lout):
            for j in range(e["taps"]):
                r = e["stride"] * p + j - e["pad"]
                if 0 <= r < 20:
                    got[:, p] += rows[:, r] @ wp[:, j].T
        got = got + e["b"]
        got = got.reshape(3, -1, cout).permute(0, 2, 1) if stride == 1 else got.permute(0, 2, 1)
        assert float((got - ref).abs().max()) < 1e-4


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

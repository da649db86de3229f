"""DisCo / CaMN inference oracle (oracle/lstm_models_oracle.py) against the reference modules run
live in the build container and against the golden fixture generated from them.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import lstm_models_oracle as lo
from oracle import reference_harness as rh
from pantomatrix_amd import synthetic

CFG = dict(lo.DEFAULT_CFG)


def weights(kind, seed=0):
    """The synthetic weights of tools/workloads.py (what bench.py loads); the product's spec must name the oracle's tensors."""
    from tools import workloads
    sd = workloads.lstm_weights(kind, seed)
    spec = lo.disco_spec(CFG) if kind == "disco" else lo.camn_spec(CFG)
    assert list(sd) == list(spec) and all(tuple(sd[k].shape) == tuple(spec[k][0]) for k in sd)
    return sd


def inputs(bs=2, frames=34, seed=3, with_seed_motion=False):
    g = torch.Generator().manual_seed(seed)
    audio = 0.1 * torch.randn(bs, frames * 16000 // 15, generator=g)
    motion = 0.3 * torch.randn(bs, frames, CFG["pose_dims"], generator=g) if with_seed_motion else None
    return audio, torch.zeros(bs, 1, dtype=torch.long), motion


def run_oracle(kind, sd, audio, spk, motion):
    fn = lo.disco_forward if kind == "disco" else lo.camn_forward
    with torch.no_grad():
        return fn(sd, CFG, audio, spk, CFG["seed_frames"], motion)


def test_lstm_direction_matches_torch():
    g = torch.Generator().manual_seed(0)
    lstm = torch.nn.LSTM(11, 7, num_layers=2, batch_first=True, bidirectional=True).eval()
    x = torch.randn(3, 9, 11, generator=g)
    with torch.no_grad():
        want, _ = lstm(x)
        got = lo.lstm_bidirectional({"l." + k: v for k, v in lstm.state_dict().items()}, "l", x, 2)
    assert torch.allclose(got, want, atol=1e-6)


@pytest.mark.skipif(not rh.available(), reason="needs /root/reference (build container only)")
@pytest.mark.parametrize("kind", ["disco", "camn"])
@pytest.mark.parametrize("with_seed_motion", [False, True])
def test_matches_reference_live(kind, with_seed_motion):
    sd = weights(kind)
    model = rh.build_reference_lstm_model(kind, CFG, sd)         # strict load: the spec IS the reference's key / shape set
    audio, spk, motion = inputs(with_seed_motion=with_seed_motion)
    with torch.no_grad():
        ref = model(audio, spk, seed_frames=CFG["seed_frames"], seed_motion=motion)
    got = run_oracle(kind, sd, audio, spk, motion)
    for k, v in ref.items():
        assert got[k].shape == v.shape, (k, got[k].shape, v.shape)
        tol = 1e-3 if k == "motion_axis_angle" else 2e-5      # near angle pi, sqrt(1 + trace) turns 1-ulp differences into ~5e-4
        assert float((got[k] - v).abs().max()) < tol, (kind, k, float((got[k] - v).abs().max()))
    assert got["motion_axis_angle"].shape[-1] == 165


def test_matches_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "lstm_models.npz"))
    for kind in ("disco", "camn"):
        for tag, wsm in (("plain", False), ("seeded", True)):
            audio, spk, motion = inputs(with_seed_motion=wsm)
            got = run_oracle(kind, weights(kind), audio, spk, motion)
            np.testing.assert_allclose(got["motion"].reshape(2, -1, 258).numpy(), g[f"{kind}_{tag}_motion"], atol=2e-5, rtol=0)
            np.testing.assert_allclose(got["motion_axis_angle"].numpy(), g[f"{kind}_{tag}_axis_angle"], atol=1e-3, rtol=0)

"""Live comparison oracle <-> reference (imported read-only).  Only runs where /root/reference
exists (the build container); the committed golden vectors cover the GPU box."""
import pytest
import torch

import common
from oracle import emage_oracle as orc
from oracle import reference_harness as rh
from pantomatrix_amd import synthetic, spec
from pantomatrix_amd.configuration_emage_audio import EmageAudioConfig

pytestmark = pytest.mark.skipif(not rh.available(), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref_models():
    acfg, vqc, gc = common.cfg_dicts()
    return rh.build_reference(acfg, vqc, gc, seed=0)


def test_state_dict_keys_match_reference(ref_models):
    """The spec table (the checkpoint-format contract) has exactly the reference's keys and shapes."""
    model, vq = ref_models
    sd = model.state_dict()
    mine = spec.audio_model_spec(EmageAudioConfig(**spec.EMAGE_AUDIO_DEFAULTS))
    assert list(sd.keys()) == list(mine.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(mine[k][0]), k


def test_inference_matches_reference(ref_models):
    model, vq = ref_models
    om, ovq = common.oracle_models()
    audio = synthetic.synthetic_audio(1, synthetic.samples_for_frames(133), seed=99)
    spk = torch.zeros(1, 1, dtype=torch.long)
    with torch.no_grad():
        r = model.inference(audio, spk, vq)
        o = om.inference(audio, spk, ovq)
    for k in orc.OUT_KEYS:
        assert r[k].shape == o[k].shape == (1, 133, 256)
        assert float((r[k] - o[k]).abs().max()) < 2e-4, k


def test_decode_matches_reference(ref_models):
    _, vq = ref_models
    _, ovq = common.oracle_models()
    g = torch.Generator().manual_seed(21)
    idx = {p: torch.randint(0, 256, (2, 33), generator=g) for p in ("upper", "hands", "lower")}
    lat = torch.randn(2, 33, 256, generator=g)
    with torch.no_grad():
        r = vq.decode(face_latent=lat, upper_index=idx["upper"], hands_index=idx["hands"], lower_index=idx["lower"],
                      get_global_motion=True, ref_trans=torch.zeros(1, 3))
        o = ovq.decode(face_latent=lat, upper_index=idx["upper"], hands_index=idx["hands"], lower_index=idx["lower"],
                       get_global_motion=True, ref_trans=torch.zeros(1, 3))
    for k in ("expression", "all_motion4inference", "motion_axis_angle", "trans"):
        assert float((r[k] - o[k]).abs().max()) < 1e-4, k

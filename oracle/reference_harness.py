"""Run the REAL reference implementation (imported read-only from /root/reference) on the
synthetic weights — TEST INFRASTRUCTURE.  Used to pin `oracle/emage_oracle.py` and to
generate `tests/golden/*.npz`.  /root/reference exists only in the build container; every
caller must check `available()` first (the GPU box has no reference).
"""
from __future__ import annotations

import os
import sys
import warnings

REFERENCE_ROOT = "/root/reference"
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_stubs")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models", "emage_audio"))


def import_reference():
    """Import `models.emage_audio` from the reference.  `omegaconf` is not installed, so a
    stub exposing `OmegaConf.to_container` is put on the path."""
    try:
        import omegaconf  # noqa: F401
    except ImportError:
        if _STUBS not in sys.path:
            sys.path.insert(0, _STUBS)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(1, REFERENCE_ROOT)
    import models.emage_audio as ref  # namespace package: the reference has no models/__init__.py
    return ref


def build_reference(audio_cfg: dict, vq_cfgs: dict, global_cfg: dict, seed: int = 0):
    """Instantiate the reference modules and load the synthetic weights into them with
    `load_state_dict(strict=True)` (from_pretrained is broken under transformers 5.x).
    Returns (EmageAudioModel, EmageVQModel), both in eval mode."""
    from pantomatrix_amd import synthetic
    from pantomatrix_amd.configuration_emage_audio import EmageAudioConfig, EmageVQVAEConvConfig, EmageVAEConvConfig
    ref = import_reference()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = ref.EmageAudioModel(ref.EmageAudioConfig(**audio_cfg))
    model.load_state_dict(synthetic.audio_model_state(EmageAudioConfig(**audio_cfg), seed), strict=True)
    parts = {}
    for part in ("face", "upper", "hands", "lower"):
        m = ref.EmageVQVAEConv(ref.EmageVQVAEConvConfig(**vq_cfgs[part]))
        m.load_state_dict(synthetic.vqvae_state(EmageVQVAEConvConfig(**vq_cfgs[part]), part, seed), strict=True)
        parts[part] = m
    g = ref.EmageVAEConv(ref.EmageVAEConvConfig(**global_cfg))
    g.load_state_dict(synthetic.vae_state(EmageVAEConvConfig(**global_cfg), seed), strict=True)
    vq = ref.EmageVQModel(face_model=parts["face"], upper_model=parts["upper"], hands_model=parts["hands"],
                          lower_model=parts["lower"], global_model=g)
    return model.eval(), vq.eval()

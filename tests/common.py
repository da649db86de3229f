"""Shared builders for the parity tests: synthetic configs, oracle models, product models."""
from __future__ import annotations

import functools

import torch

from pantomatrix_amd import spec, synthetic
from pantomatrix_amd.configuration_emage_audio import EmageAudioConfig, EmageVQVAEConvConfig, EmageVAEConvConfig

PARTS = ("face", "upper", "hands", "lower")


def cfg_dicts(vae_layer=2, global_layer=4, global_length=240):
    return (dict(spec.EMAGE_AUDIO_DEFAULTS), {p: spec.default_vq_cfg_dict(p, vae_layer) for p in PARTS},
            spec.default_global_cfg_dict(global_layer, global_length))


@functools.lru_cache(maxsize=4)
def oracle_models(seed=0, vae_layer=2):
    from oracle import emage_oracle as orc
    acfg, vqc, gc = cfg_dicts(vae_layer)
    cfg = EmageAudioConfig(**acfg)
    model = orc.AudioModel(synthetic.audio_model_state(cfg, seed), cfg)
    parts = [orc.VQVAE(synthetic.vqvae_state(EmageVQVAEConvConfig(**vqc[p]), p, seed), EmageVQVAEConvConfig(**vqc[p])) for p in PARTS]
    vq = orc.VQModel(*parts, orc.VAE(synthetic.vae_state(EmageVAEConvConfig(**gc), seed), EmageVAEConvConfig(**gc)))
    return model, vq


def window_inputs(batch, frames=64, seed=7):
    """A forward() window: audio, speaker ids, a partly-masked motion window."""
    from oracle import emage_oracle as orc
    g = torch.Generator().manual_seed(seed)
    audio = 0.1 * torch.randn(batch, frames * 533, generator=g)
    aa = 0.3 * torch.randn(batch, frames, 55, 3, generator=g)
    motion = torch.cat([orc.axis_angle_to_rotation_6d(aa).reshape(batch, frames, 330),
                        0.1 * torch.randn(batch, frames, 7, generator=g)], dim=-1)
    mask = torch.ones(batch, frames, 337)
    mask[:, :4] = 0
    mask[:, 20:24, :100] = 0
    return audio, torch.zeros(batch, 1, dtype=torch.long), motion, mask


def vq_api_inputs(batch=3, frames=24, seed=41):
    """Seeded inputs of EmageVQModel.map2index / map2latent: (rot6d (B,T,330), expression (B,T,100), contact (B,T,4) in
    {0,1}, trans (B,T,3))."""
    from oracle import emage_oracle as orc
    g = torch.Generator().manual_seed(seed)
    aa = 0.4 * torch.randn(batch, frames, 55, 3, generator=g)
    rot6d = orc.axis_angle_to_rotation_6d(aa).reshape(batch, frames, 330)
    expr = 0.5 * torch.randn(batch, frames, 100, generator=g)
    contact = (torch.rand(batch, frames, 4, generator=g) > 0.5).float()
    trans = 0.1 * torch.randn(batch, frames, 3, generator=g)
    return rot6d, expr, contact, trans


_STATE_CACHE = {}


def _synthetic_state(make, *key):
    """The seeded synthetic weights of a model, generated once per process (`load_state_dict` copies them into the parameters, so the
    cached tensors are never aliased by a model; the suites build ~100 model sets from the same seeds)."""
    if key not in _STATE_CACHE:
        _STATE_CACHE[key] = make()
    return _STATE_CACHE[key]


def product_models(seed=0, vae_layer=2, precision="fp32", device="cpu"):
    """pantomatrix_amd model objects loaded with the same synthetic weights as `oracle_models`."""
    import json
    import pantomatrix_amd as pa
    acfg, vqc, gc = cfg_dicts(vae_layer)
    cfg = pa.EmageAudioConfig(**acfg)
    model = pa.EmageAudioModel(cfg)
    model.load_state_dict(_synthetic_state(lambda: synthetic.audio_model_state(cfg, seed), "audio", json.dumps(acfg, sort_keys=True), seed))
    parts = {}
    for p in PARTS:
        c = pa.EmageVQVAEConvConfig(**vqc[p])
        parts[p] = pa.EmageVQVAEConv(c)
        parts[p].load_state_dict(_synthetic_state(lambda: synthetic.vqvae_state(c, p, seed), "vq", p, json.dumps(vqc[p], sort_keys=True), seed))
    g = pa.EmageVAEConv(pa.EmageVAEConvConfig(**gc))
    g.load_state_dict(_synthetic_state(lambda: synthetic.vae_state(pa.EmageVAEConvConfig(**gc), seed), "global", json.dumps(gc, sort_keys=True), seed))
    vq = pa.EmageVQModel(face_model=parts["face"], upper_model=parts["upper"], hands_model=parts["hands"],
                         lower_model=parts["lower"], global_model=g)
    model.set_precision(precision)
    vq.set_precision(precision)
    if device != "cpu":
        model.to(device)
        vq.to(device)
    return model.eval(), vq.eval()


def product_infer_clip(model, vq, audio, speaker_id=None):
    """test_emage_audio.py:16-53 against the product classes; returns numpy (poses, expressions, trans)."""
    bs = audio.shape[0]
    dev = model.device
    if speaker_id is None:
        speaker_id = torch.zeros(bs, 1, dtype=torch.long, device=dev)
    lat = model.inference(audio.to(dev), speaker_id, vq)
    pred = vq.decode(**model._select_codes(lat), get_global_motion=True, ref_trans=torch.zeros(1, 3, device=dev))
    return (pred["motion_axis_angle"].cpu().numpy(), pred["expression"].cpu().numpy(), pred["trans"].cpu().numpy()), lat

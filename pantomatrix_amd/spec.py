"""Checkpoint layout of the EMAGE hot path: parameter names, shapes and roles.

This table IS the weight-format half of the drop-in boundary: the
state-dict keys a HuggingFace EMAGE checkpoint carries, so that `load_state_dict`
accepts reference checkpoints unchanged.  The key names follow the reference module
tree:

* ``EmageAudioModel``      /root/reference/models/emage_audio/modeling_emage_audio.py:211-263
* ``EmageVQVAEConv``       modeling_emage_audio.py:34-45 (encoder / quantizer / decoder)
* ``EmageVAEConv``         modeling_emage_audio.py:19-25 (encoder / decoder)
* ``VQEncoderV5/V6``       processing_emage_audio.py:189-235 (``main.{3i, 3i+2}``)
* ``VQDecoderV5``          processing_emage_audio.py:237-261
* ``WavEncoder/BasicBlock`` processing_emage_audio.py:263-314
* torch ``nn.TransformerEncoderLayer`` / ``nn.TransformerDecoderLayer`` parameter names.

Each entry maps ``name -> (shape, role)``; *role* tells the synthetic-weight generator
and the weight packer what the tensor is (it never changes the name or shape).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

Spec = "OrderedDict[str, Tuple[Tuple[int, ...], str]]"

# WavEncoder geometry, processing_emage_audio.py:300-307: (cin, cout, stride, pad_conv1, has_downsample)
def wav_encoder_blocks(out_dim: int):
    q, h = out_dim // 4, out_dim // 2
    return [
        (1, q, 5, 1600, True),
        (q, q, 6, 0, True),
        (q, q, 1, 7, False),
        (q, h, 6, 0, True),
        (h, h, 1, 7, False),
        (h, out_dim, 3, 0, True),
    ]

WAV_KERNEL = 15  # processing_emage_audio.py:301-306 (ker_size)


def _conv(spec, name, cout, cin, k):
    spec[name + ".weight"] = ((cout, cin, k), "conv_w")
    spec[name + ".bias"] = ((cout,), "bias")


def _linear(spec, name, cout, cin):
    spec[name + ".weight"] = ((cout, cin), "linear_w")
    spec[name + ".bias"] = ((cout,), "bias")


def _bn(spec, name, c):
    spec[name + ".weight"] = ((c,), "norm_w")
    spec[name + ".bias"] = ((c,), "norm_b")
    spec[name + ".running_mean"] = ((c,), "bn_mean")
    spec[name + ".running_var"] = ((c,), "bn_var")
    spec[name + ".num_batches_tracked"] = ((), "bn_count")


def _ln(spec, name, c):
    spec[name + ".weight"] = ((c,), "norm_w")
    spec[name + ".bias"] = ((c,), "norm_b")


def _mha(spec, name, d):
    spec[name + ".in_proj_weight"] = ((3 * d, d), "linear_w")
    spec[name + ".in_proj_bias"] = ((3 * d,), "bias")
    _linear(spec, name + ".out_proj", d, d)


def _encoder_layer(spec, name, d, ff):
    _mha(spec, name + ".self_attn", d)
    _linear(spec, name + ".linear1", ff, d)
    _linear(spec, name + ".linear2", d, ff)
    _ln(spec, name + ".norm1", d)
    _ln(spec, name + ".norm2", d)


def _decoder_layer(spec, name, d, ff):
    _mha(spec, name + ".self_attn", d)
    _mha(spec, name + ".multihead_attn", d)
    _linear(spec, name + ".linear1", ff, d)
    _linear(spec, name + ".linear2", d, ff)
    _ln(spec, name + ".norm1", d)
    _ln(spec, name + ".norm2", d)
    _ln(spec, name + ".norm3", d)


def _mlp(spec, name, cin, mid, cout):
    _linear(spec, name + ".fc1", mid, cin)
    _linear(spec, name + ".fc2", cout, mid)


def _wav_encoder(spec, name, out_dim):
    for i, (cin, cout, _s, _p, ds) in enumerate(wav_encoder_blocks(out_dim)):
        b = f"{name}.feat_extractor.{i}"
        _conv(spec, b + ".conv1", cout, cin, WAV_KERNEL)
        _bn(spec, b + ".bn1", cout)
        _conv(spec, b + ".conv2", cout, cout, WAV_KERNEL)
        _bn(spec, b + ".bn2", cout)
        if ds:
            _conv(spec, b + ".downsample.0", cout, cin, WAV_KERNEL)
            _bn(spec, b + ".downsample.1", cout)


def conv_encoder_spec(spec, name, in_dim, length, n_layer):
    """VQEncoderV5 / V6: n_layer x [Conv k3, LeakyReLU(0.2), ResBlock]."""
    for i in range(n_layer):
        _conv(spec, f"{name}.main.{3 * i}", length, in_dim if i == 0 else length, 3)
        _conv(spec, f"{name}.main.{3 * i + 2}.model.0", length, length, 3)
        _conv(spec, f"{name}.main.{3 * i + 2}.model.2", length, length, 3)


def conv_decoder_spec(spec, name, out_dim, length, n_layer):
    """VQDecoderV5: 2 ResBlocks, n_layer x [Conv k3, LeakyReLU(0.2)], final Conv k3."""
    for i in range(2):
        _conv(spec, f"{name}.main.{i}.model.0", length, length, 3)
        _conv(spec, f"{name}.main.{i}.model.2", length, length, 3)
    chans = [length] * n_layer + [out_dim]
    for i in range(n_layer):
        _conv(spec, f"{name}.main.{2 + 2 * i}", chans[i + 1], chans[i], 3)
    _conv(spec, f"{name}.main.{2 + 2 * n_layer}", out_dim, out_dim, 3)


def vqvae_spec(cfg) -> Dict:
    """EmageVQVAEConv state dict (modeling_emage_audio.py:37-41)."""
    spec = OrderedDict()
    conv_encoder_spec(spec, "encoder", cfg.vae_test_dim, cfg.vae_length, cfg.vae_layer)
    spec["quantizer.embedding.weight"] = ((cfg.vae_codebook_size, cfg.vae_length), "codebook")
    conv_decoder_spec(spec, "decoder", cfg.vae_test_dim, cfg.vae_length, cfg.vae_layer)
    return spec


def vae_spec(cfg) -> Dict:
    """EmageVAEConv state dict (modeling_emage_audio.py:22-25)."""
    spec = OrderedDict()
    conv_encoder_spec(spec, "encoder", cfg.vae_test_dim, cfg.vae_length, cfg.vae_layer)
    conv_decoder_spec(spec, "decoder", cfg.vae_test_dim, cfg.vae_length, cfg.vae_layer)
    return spec


N_HEAD = 4               # modeling_emage_audio.py:238,241 (nhead=4)
MOTION_ENC_LAYERS = 3    # modeling_emage_audio.py:228 (args_top.vae_layer = 3)
N_CROSS_LAYERS = 8       # modeling_emage_audio.py:242
N_FACE_LAYERS = 4        # modeling_emage_audio.py:261


def audio_model_spec(cfg) -> Dict:
    """EmageAudioModel state dict, in the reference's registration order
    (modeling_emage_audio.py:211-263)."""
    d, ff = cfg.hidden_size, cfg.hidden_size * 2
    mf, af, cb = cfg.motion_f, cfg.audio_f, cfg.vae_codebook_size
    motion_dim = cfg.pose_dims + 3 + 4
    spec = OrderedDict()
    spec["mask_embedding"] = ((1, 1, motion_dim), "mask_emb")
    _wav_encoder(spec, "audio_encoder_face", af)
    _wav_encoder(spec, "audio_encoder_body", af)
    spec["speaker_embedding_body.weight"] = ((cfg.speaker_dims, d), "embedding")
    spec["speaker_embedding_face.weight"] = ((cfg.speaker_dims, d), "embedding")
    conv_encoder_spec(spec, "motion_encoder", motion_dim, mf, MOTION_ENC_LAYERS)
    _mlp(spec, "bodyhints_face", mf, d, mf)
    _mlp(spec, "bodyhints_body", mf, d, mf)
    _linear(spec, "audio_body_motion_proj", d, af)
    _linear(spec, "moton_proj", d, mf)  # (sic) reference spelling
    spec["position_embeddings.pe"] = ((1, 2 * cfg.pose_length, d), "ppe")
    _encoder_layer(spec, "transformer_en_layer", d, ff)          # unused template, kept for key parity
    _encoder_layer(spec, "motion_self_encoder.layers.0", d, ff)
    _decoder_layer(spec, "audio_motion_cross_attn_layer", d, ff)  # unused template
    for i in range(N_CROSS_LAYERS):
        _decoder_layer(spec, f"audio_motion_cross_attn.layers.{i}", d, ff)
    for part in ("upper", "hands", "lower"):
        _mlp(spec, f"motion2latent_{part}", d, d, d)
    for part in ("upper", "hands", "lower"):
        _decoder_layer(spec, f"body_motion_decoder_{part}.layers.0", d, ff)
    for part in ("upper", "hands", "lower"):
        _linear(spec, f"motion_out_proj_{part}", cb, d)
    for part in ("upper", "hands", "lower"):
        _mlp(spec, f"motion_cls_{part}", cb, d, cb)
    _linear(spec, "audio_face_motion_proj", d, af + mf)
    for i in range(N_FACE_LAYERS):
        _decoder_layer(spec, f"face_motion_decoder.layers.{i}", d, ff)
    _linear(spec, "face_out_proj", cb, d)
    _mlp(spec, "face_cls", cb, d, cb)
    return spec


# The `model:` block of /root/reference/configs/emage_audio.yaml:24-52.
EMAGE_AUDIO_DEFAULTS = dict(
    pose_fps=30, motion_f=256, pose_dims=330, pose_rep="smplx", audio_rep="wave16k",
    audio_sr=16000, audio_fps=16000, audio_norm=False, audio_f=256, speaker_f=768,
    speaker_dims=1, hidden_size=768, seed_frames=4, pose_length=64, stride=20,
    test_length=64, joint_mask=None, vae_codebook_size=256,
    ll=3, lf=3, lu=3, lh=3, cl=1, cf=0, cu=1, ch=1,
)

# Part VQ-VAE widths forced by EmageVQModel.decode / spilt_inputs slicing
# (modeling_emage_audio.py:100-107,137,149,158,168).  vae_layer / vae_length come from the
# checkpoint's config.json (SURVEY.md §8a note); these defaults are the synthetic-weight choice.
PART_DIMS = dict(face=106, upper=78, hands=180, lower=61)


def default_vq_cfg_dict(part: str, vae_layer: int = 2):
    return dict(vae_layer=vae_layer, vae_length=256, vae_test_dim=PART_DIMS[part],
                vae_codebook_size=256, vae_quantizer_lambda=1.0)


def default_global_cfg_dict(vae_layer: int = 4, vae_length: int = 240):
    return dict(vae_layer=vae_layer, vae_length=vae_length, vae_test_dim=61)


# ======================================================================================
# DisCo / CaMN: checkpoint layouts of
#   DiscoAudioModel   /root/reference/models/disco_audio/modeling_disco_audio.py:171-197   (D:)
#   CamnAudioModel    /root/reference/models/camn_audio/modeling_camn_audio.py:180-221     (C:)
# ======================================================================================
def lstm_wav_encoder_blocks():
    """WavEncoder of DisCo / CaMN (D:133-142): widths 32,32,32,64,64,128, strides 5,6,1,6,1,6 (15 fps features)."""
    return [(1, 32, 5, 1600, True), (32, 32, 6, 0, True), (32, 32, 1, 7, False), (32, 64, 6, 0, True),
            (64, 64, 1, 7, False), (64, 128, 6, 0, True)]


LSTM_MODEL_DEFAULTS = dict(pose_dims=258, body_dims=78, hands_dims=180, audio_f=128, speaker_f=16, speaker_dims=1,
                           hidden_size=512, n_layer=4, dropout_prob=0.1, seed_frames=4, joint_mask="local_upper",
                           pose_rep="smplx", pose_fps=15, motion_f=256)      # configs/disco_audio.yaml, configs/camn_audio.yaml

# MASK_DICT["local_upper"] (D:19-27): the 13 upper-body joints and the 30 hand joints, in joint order
LOCAL_UPPER_JOINTS = [j for j in range(55) if j in (3, 6, 9, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21) or j >= 25]


def _lstm_wav_encoder(spec, name):
    for i, (cin, cout, _stride, _pad, ds) in enumerate(lstm_wav_encoder_blocks()):
        b = f"{name}.feat_extractor.{i}"
        _conv(spec, b + ".conv1", cout, cin, WAV_KERNEL)
        _bn(spec, b + ".bn1", cout)
        _conv(spec, b + ".conv2", cout, cout, WAV_KERNEL)
        _bn(spec, b + ".bn2", cout)
        if ds:
            _conv(spec, b + ".downsample.0", cout, cin, WAV_KERNEL)
            _bn(spec, b + ".downsample.1", cout)


def _lstm(spec, name, cin, hid, n_layers):
    """torch.nn.LSTM(bidirectional=True) parameter names; gate order in the packed rows: input, forget, cell, output."""
    for k in range(n_layers):
        for suffix in ("", "_reverse"):
            spec[f"{name}.weight_ih_l{k}{suffix}"] = ((4 * hid, cin if k == 0 else 2 * hid), "linear_w")
            spec[f"{name}.weight_hh_l{k}{suffix}"] = ((4 * hid, hid), "linear_w")
            spec[f"{name}.bias_ih_l{k}{suffix}"] = ((4 * hid,), "bias")
            spec[f"{name}.bias_hh_l{k}{suffix}"] = ((4 * hid,), "bias")


def disco_model_spec(cfg) -> Dict:
    spec = OrderedDict()
    af, hid, pd = cfg.audio_f, cfg.hidden_size, cfg.pose_dims
    _lstm_wav_encoder(spec, "audio_encoder")
    if cfg.speaker_f > 0:
        spec["speaker_embedding.weight"] = ((cfg.speaker_dims, cfg.speaker_f), "embedding")
    for nm in ("audio_encoder_c1", "audio_encoder_c2", "audio_encoder_r"):
        _mlp(spec, nm, af, hid, af)
    _mlp(spec, "selector", af, hid, 2)
    _lstm(spec, "body_motion_decoder", pd + 1 + cfg.speaker_f + 2 * af, hid, cfg.n_layer)
    _mlp(spec, "body_out", hid, hid, pd)
    return spec


def camn_model_spec(cfg) -> Dict:
    spec = OrderedDict()
    af, hid, pd = cfg.audio_f, cfg.hidden_size, cfg.pose_dims
    _lstm_wav_encoder(spec, "audio_encoder")
    if cfg.speaker_f > 0:
        spec["speaker_embedding.weight"] = ((cfg.speaker_dims, cfg.speaker_f), "embedding")
    cin = pd + 1 + cfg.speaker_f + af
    _lstm(spec, "body_motion_decoder", cin, hid, cfg.n_layer)
    _mlp(spec, "body_out", hid, hid, cfg.body_dims)
    _lstm(spec, "hands_motion_decoder", cin + cfg.body_dims, hid, cfg.n_layer)
    _mlp(spec, "hands_out", hid, hid, cfg.hands_dims)
    return spec

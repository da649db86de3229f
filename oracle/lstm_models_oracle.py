"""CPU restatement of the two LSTM gesture generators of the reference, DisCo and CaMN (audio variants) — TEST
INFRASTRUCTURE for SURVEY.md §8(f) rows 3 and 4 (no product code imports this file; no HIP path exists for them yet).

  D: /root/reference/models/disco_audio/modeling_disco_audio.py    (DiscoAudioModel.forward D:199-266)
  C: /root/reference/models/camn_audio/modeling_camn_audio.py      (CamnAudioModel.forward  C:223-280)

Both: raw 16 kHz waveform -> WavEncoder (six Conv1d k=15 residual blocks with BatchNorm, 32/32/32/64/64/128 channels,
strides 5,6,1,6,1,6 -> 15 fps features, D:133-149 / C:133-149) -> concatenated with a speaker embedding and the seed
motion -> 4-layer bidirectional LSTM (hidden 512; forward and backward halves ADDED, D:252-253) -> MLP -> rot-6D ->
axis-angle scattered to the 55 SMPL-X joints.  DisCo adds a two-expert content branch gated by a softmax selector and a
"rhythm" branch (D:244-249); CaMN cascades a second LSTM for the hands on [inputs | body output] (C:262-270).

The rotation helpers of D / C are verbatim the ones of the EMAGE path (same formulas), so `emage_oracle`'s are used.
Pinned against the reference modules run live in the build container and through tests/golden/lstm_models.npz
(tests/test_lstm_models_oracle.py, tests/golden/make_golden_lstm.py).
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import emage_oracle as orc

# (cin, cout, stride, pad of conv1 and of the shortcut conv): D:137-142.  A block has the conv+BN shortcut when the stride
# is not 1 or the width changes (D:106-112).
WAV_BLOCKS = [(1, 32, 5, 1600), (32, 32, 6, 0), (32, 32, 1, 7), (32, 64, 6, 0), (64, 64, 1, 7), (64, 128, 6, 0)]

MASK_LOCAL_UPPER = [j in (3, 6, 9, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21) or j >= 25 for j in range(55)]   # D:19-27
DEFAULT_CFG = dict(pose_dims=258, body_dims=78, hands_dims=180, audio_f=128, speaker_f=16, speaker_dims=1,
                   hidden_size=512, n_layer=4, dropout_prob=0.1, seed_frames=4, joint_mask="local_upper",
                   pose_rep="smplx", pose_fps=15, motion_f=256)          # configs/disco_audio.yaml, configs/camn_audio.yaml


def wav_encoder(sd, prefix, wav):
    """WavEncoder.forward (D:144-149), eval-mode BatchNorm: (B, L) -> (B, T', 128)."""
    h = wav.unsqueeze(1)
    for i, (cin, cout, stride, pad) in enumerate(WAV_BLOCKS):
        b = f"{prefix}.feat_extractor.{i}"
        y = F.conv1d(h, sd[b + ".conv1.weight"], sd[b + ".conv1.bias"], stride=stride, padding=pad)
        y = F.leaky_relu(orc._bn_eval(sd, b + ".bn1", y), 0.01)
        y = orc._bn_eval(sd, b + ".bn2", F.conv1d(y, sd[b + ".conv2.weight"], sd[b + ".conv2.bias"], stride=1, padding=7))
        if stride != 1 or cin != cout:
            h = orc._bn_eval(sd, b + ".downsample.1",
                             F.conv1d(h, sd[b + ".downsample.0.weight"], sd[b + ".downsample.0.bias"], stride=stride, padding=pad))
        h = F.leaky_relu(y + h, 0.01)
    return h.transpose(1, 2)


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one nn.LSTM layer, batch-first, zero initial state.  Gate order in the packed weights is
    input, forget, cell, output: c' = sigmoid(f) c + sigmoid(i) tanh(g); h' = sigmoid(o) tanh(c')."""
    bsz, t, _ = x.shape
    hid = w_hh.shape[1]
    h = x.new_zeros(bsz, hid)
    c = x.new_zeros(bsz, hid)
    gx = F.linear(x, w_ih, b_ih)                              # the input projection of all steps at once
    out = [None] * t
    for s in (range(t - 1, -1, -1) if reverse else range(t)):
        i, f, g, o = (gx[:, s] + F.linear(h, w_hh, b_hh)).chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[s] = h
    return torch.stack(out, dim=1)


def lstm_bidirectional(sd, name, x, n_layers):
    """nn.LSTM(bidirectional=True, batch_first=True) in eval mode (no inter-layer dropout): layer k consumes the
    concatenated [forward | backward] outputs of layer k-1."""
    for k in range(n_layers):
        outs = []
        for suffix, rev in (("", False), ("_reverse", True)):
            outs.append(lstm_direction(x, sd[f"{name}.weight_ih_l{k}{suffix}"], sd[f"{name}.weight_hh_l{k}{suffix}"],
                                       sd[f"{name}.bias_ih_l{k}{suffix}"], sd[f"{name}.bias_hh_l{k}{suffix}"], rev))
        x = torch.cat(outs, dim=2)
    return x


def _inputs(sd, cfg, audio, speaker_id, seed_frames, seed_motion):
    """Shared head of both forwards (D:200-243, C:224-259): audio features, speaker features, the seed-motion channel
    block with its `is seed` flag, length reconciliation between motion and audio frames."""
    audio_feat = wav_encoder(sd, "audio_encoder", audio)
    bs, t, _ = audio_feat.shape
    if cfg["speaker_f"] > 0:
        speaker_feat = sd["speaker_embedding.weight"][speaker_id].repeat(1, t, 1)
    else:
        speaker_feat = audio_feat.new_zeros(bs, t, 0)
    pd = cfg["pose_dims"]
    if seed_motion is None:
        seed = audio_feat.new_zeros(bs, t, pd + 1)
        seed[:, :seed_frames, -1] = 1
    else:
        t_m = seed_motion.shape[1]
        seed = audio_feat.new_zeros(bs, t_m, pd + 1)
        seed[:, :seed_frames, :-1] = seed_motion[:, :seed_frames]
        seed[:, :seed_frames, -1] = 1
        if t_m != t:
            diff = t_m - t
            # D:238-242 verbatim: for a SHORTER seed the reference appends `seed[:, -diff:]` with diff < 0, i.e. the frames
            # from index |diff| on (not the last |diff| frames); the lengths then only line up when t_m = (t + |diff|) / 2
            seed = seed[:, :t] if diff > 0 else torch.cat((seed, seed[:, -diff:]), 1)
    return audio_feat, speaker_feat, seed


def _axis_angle(recombine, cfg):
    bs, t = recombine.shape[:2]
    pd = cfg["pose_dims"]
    aa = orc.rotation_6d_to_axis_angle(recombine.reshape(-1, pd // 6, 6)).reshape(bs, t, -1)
    mask = torch.tensor(MASK_LOCAL_UPPER if cfg["joint_mask"] == "local_upper" else [False] + [True] * 54)
    n_sel = int(mask.sum())
    out = aa.new_zeros(bs, t, 55, aa.shape[-1] // n_sel)
    out[:, :, mask] = aa.reshape(bs, t, n_sel, -1)                                       # recover_from_mask_ts, D:81-96
    return out.reshape(bs, t, -1)


def disco_forward(sd, cfg, audio, speaker_id, seed_frames=4, seed_motion=None):
    """DiscoAudioModel.forward, D:199-266."""
    audio_feat, speaker_feat, seed = _inputs(sd, cfg, audio, speaker_id, seed_frames, seed_motion)
    c1 = orc.mlp(sd, "audio_encoder_c1", audio_feat)
    c2 = orc.mlp(sd, "audio_encoder_c2", audio_feat)
    r = orc.mlp(sd, "audio_encoder_r", audio_feat)
    w = torch.softmax(orc.mlp(sd, "selector", audio_feat), dim=2)
    c = w[:, :, 0:1] * c1 + w[:, :, 1:2] * c2
    in_fea = torch.cat((torch.cat((c, r), dim=2), speaker_feat, seed), dim=2)
    hid = cfg["hidden_size"]
    body = lstm_bidirectional(sd, "body_motion_decoder", in_fea, cfg["n_layer"])
    body = body[:, :, :hid] + body[:, :, hid:]
    recombine = orc.mlp(sd, "body_out", body)
    return {"motion": recombine, "motion_axis_angle": _axis_angle(recombine, cfg), "audio_fea_c": c, "audio_fea_r": r}


def camn_forward(sd, cfg, audio, speaker_id, seed_frames=4, seed_motion=None):
    """CamnAudioModel.forward, C:223-280 (pose_rep "smplx": body joints first, hand joints after, C:211-217)."""
    audio_feat, speaker_feat, seed = _inputs(sd, cfg, audio, speaker_id, seed_frames, seed_motion)
    in_fea = torch.cat((audio_feat, speaker_feat, seed), dim=2)
    hid = cfg["hidden_size"]
    body = lstm_bidirectional(sd, "body_motion_decoder", in_fea, cfg["n_layer"])
    body = orc.mlp(sd, "body_out", body[:, :, :hid] + body[:, :, hid:])
    hands = lstm_bidirectional(sd, "hands_motion_decoder", torch.cat((in_fea, body), dim=2), cfg["n_layer"])
    hands = orc.mlp(sd, "hands_out", hands[:, :, :hid] + hands[:, :, hid:])
    bs, t, _ = body.shape
    recombine = torch.cat((body.reshape(bs, t, -1, 6), hands.reshape(bs, t, -1, 6)), dim=2)
    return {"motion": recombine, "motion_axis_angle": _axis_angle(recombine, cfg)}


# --------------------------------------------------------------------------------------
# state-dict specs (name -> (shape, role)), roles as in pantomatrix_amd.synthetic.draw
# --------------------------------------------------------------------------------------
def _wav_spec(spec, prefix):
    for i, (cin, cout, stride, _pad) in enumerate(WAV_BLOCKS):
        b = f"{prefix}.feat_extractor.{i}"
        convs = [("conv1", cin), ("conv2", cout)] + ([("downsample.0", cin)] if (stride != 1 or cin != cout) else [])
        for nm, ci in convs:
            spec[f"{b}.{nm}.weight"] = ((cout, ci, 15), "conv_w")
            spec[f"{b}.{nm}.bias"] = ((cout,), "bias")
            bn = {"conv1": "bn1", "conv2": "bn2", "downsample.0": "downsample.1"}[nm]
            spec[f"{b}.{bn}.weight"] = ((cout,), "norm_w")
            spec[f"{b}.{bn}.bias"] = ((cout,), "norm_b")
            spec[f"{b}.{bn}.running_mean"] = ((cout,), "bn_mean")
            spec[f"{b}.{bn}.running_var"] = ((cout,), "bn_var")
            spec[f"{b}.{bn}.num_batches_tracked"] = ((), "bn_count")


def _mlp_spec(spec, name, cin, mid, cout):
    spec[name + ".fc1.weight"], spec[name + ".fc1.bias"] = ((mid, cin), "linear_w"), ((mid,), "bias")
    spec[name + ".fc2.weight"], spec[name + ".fc2.bias"] = ((cout, mid), "linear_w"), ((cout,), "bias")


def _lstm_spec(spec, name, cin, hid, n_layers):
    for k in range(n_layers):
        for suffix in ("", "_reverse"):
            spec[f"{name}.weight_ih_l{k}{suffix}"] = ((4 * hid, cin if k == 0 else 2 * hid), "linear_w")
            spec[f"{name}.weight_hh_l{k}{suffix}"] = ((4 * hid, hid), "linear_w")
            spec[f"{name}.bias_ih_l{k}{suffix}"] = ((4 * hid,), "bias")
            spec[f"{name}.bias_hh_l{k}{suffix}"] = ((4 * hid,), "bias")


def disco_spec(cfg):
    spec = OrderedDict()
    af, hid, pd = cfg["audio_f"], cfg["hidden_size"], cfg["pose_dims"]
    _wav_spec(spec, "audio_encoder")
    if cfg["speaker_f"] > 0:
        spec["speaker_embedding.weight"] = ((cfg["speaker_dims"], cfg["speaker_f"]), "embedding")
    for nm in ("audio_encoder_c1", "audio_encoder_c2", "audio_encoder_r"):
        _mlp_spec(spec, nm, af, hid, af)
    _mlp_spec(spec, "selector", af, hid, 2)
    _lstm_spec(spec, "body_motion_decoder", pd + 1 + cfg["speaker_f"] + 2 * af, hid, cfg["n_layer"])
    _mlp_spec(spec, "body_out", hid, hid, pd)
    return spec


def camn_spec(cfg):
    spec = OrderedDict()
    af, hid, pd = cfg["audio_f"], cfg["hidden_size"], cfg["pose_dims"]
    _wav_spec(spec, "audio_encoder")
    if cfg["speaker_f"] > 0:
        spec["speaker_embedding.weight"] = ((cfg["speaker_dims"], cfg["speaker_f"]), "embedding")
    cin = pd + 1 + cfg["speaker_f"] + af
    _lstm_spec(spec, "body_motion_decoder", cin, hid, cfg["n_layer"])
    _mlp_spec(spec, "body_out", hid, hid, cfg["body_dims"])
    _lstm_spec(spec, "hands_motion_decoder", cin + cfg["body_dims"], hid, cfg["n_layer"])
    _mlp_spec(spec, "hands_out", hid, hid, cfg["hands_dims"])
    return spec

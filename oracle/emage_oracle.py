"""CPU oracle for the EMAGE hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain fp32 PyTorch-on-CPU restatement of the reference algorithm for the path
`BASELINE.json.north_star` names.  It is functional code over
flat state dicts (the checkpoint format of SURVEY.md §8b), written from the reference's
formulas; every function cites the reference lines it follows
(paths relative to /root/reference/models/emage_audio/).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
this module, and only as the checker / the timed CPU baseline.  The product
(`pantomatrix_amd`) never imports it and has no CPU fallback.

Pinning status: the reference holds NO golden vectors or known-answer tests for this path
(SURVEY.md §4, §8c).  The oracle is therefore pinned against the reference implementation
itself, executed in the build container: `tests/test_oracle_vs_reference.py` runs both on the
same seeded weights/inputs whenever /root/reference is present, and `tests/golden/*.npz`
(generated from the REFERENCE by `tests/golden/make_golden.py`) pin it where the reference
cannot travel (the GPU box).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

N_HEAD = 4  # modeling_emage_audio.py:238,241

# Joint partition of the 55 SMPL-X joints, modeling_emage_audio.py:75-90,104,181,185.
UPPER_JOINTS = [3, 6, 9, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21]
LOWER_JOINTS = [0, 1, 2, 4, 5, 7, 8, 10, 11]
HANDS_JOINTS = list(range(25, 55))
JAW_JOINT = 22


# --------------------------------------------------------------------------------------
# rotation helpers — processing_emage_audio.py:6-104
# --------------------------------------------------------------------------------------
def _sqrt_positive_part(x):
    """processing_emage_audio.py:10-14: sqrt(x) where x > 0, else exactly 0."""
    return torch.where(x > 0, torch.sqrt(torch.clamp(x, min=0)), torch.zeros_like(x))


def _copysign(a, b):
    """processing_emage_audio.py:6-8."""
    return torch.where((a < 0) != (b < 0), -a, a)


def rotation_6d_to_matrix(d6):
    """processing_emage_audio.py:50-56 (Gram-Schmidt; F.normalize eps=1e-12)."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = F.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def matrix_to_quaternion(m):
    """processing_emage_audio.py:16-29."""
    m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]
    w = 0.5 * _sqrt_positive_part(1 + m00 + m11 + m22)
    x = 0.5 * _sqrt_positive_part(1 + m00 - m11 - m22)
    y = 0.5 * _sqrt_positive_part(1 - m00 + m11 - m22)
    z = 0.5 * _sqrt_positive_part(1 - m00 - m11 + m22)
    x = _copysign(x, m[..., 2, 1] - m[..., 1, 2])
    y = _copysign(y, m[..., 0, 2] - m[..., 2, 0])
    z = _copysign(z, m[..., 1, 0] - m[..., 0, 1])
    return torch.stack((w, x, y, z), -1)


def _sin_half_over_angle(angles, half_angles):
    """Shared small-angle branch, processing_emage_audio.py:35-43 and :66-74."""
    small = angles.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(angles), angles)
    return torch.where(small, 0.5 - (angles * angles) / 48, torch.sin(half_angles) / safe)


def quaternion_to_axis_angle(q):
    """processing_emage_audio.py:31-44."""
    norms = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    angles = 2 * half
    return q[..., 1:] / _sin_half_over_angle(angles, half)


def rotation_6d_to_axis_angle(d6):
    """processing_emage_audio.py:58-59."""
    return quaternion_to_axis_angle(matrix_to_quaternion(rotation_6d_to_matrix(d6)))


def axis_angle_to_quaternion(aa):
    """processing_emage_audio.py:64-79."""
    angles = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = 0.5 * angles
    return torch.cat([torch.cos(half), aa * _sin_half_over_angle(angles, half)], dim=-1)


def quaternion_to_matrix(q):
    """processing_emage_audio.py:81-99."""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def axis_angle_to_rotation_6d(aa):
    """processing_emage_audio.py:101-104 + :61-62 (first two matrix rows, flattened)."""
    m = quaternion_to_matrix(axis_angle_to_quaternion(aa))
    return m[..., :2, :].reshape(*m.shape[:-2], 6)


def velocity2position(v, dt, init_pos):
    """processing_emage_audio.py:107-115: pos[0]=init, pos[i]=v[i-1]*dt+pos[i-1]
    (sequential fp32 accumulation, product rounded before the add).  v: (B,T,1), init: (B,1)."""
    out = [init_pos.unsqueeze(1)]
    for i in range(1, v.shape[1]):
        out.append(v[:, i - 1:i] * dt + out[-1])
    return torch.cat(out, dim=1)


def scatter_joints(selected, joints, n_joints=55):
    """recover_from_mask_ts, processing_emage_audio.py:118-132: (..., len(joints)*c) ->
    (..., n_joints*c) with zeros elsewhere."""
    c = selected.shape[-1] // len(joints)
    out = torch.zeros(selected.shape[:-1] + (n_joints, c), dtype=selected.dtype)
    out[..., joints, :] = selected.reshape(selected.shape[:-1] + (len(joints), c))
    return out.reshape(selected.shape[:-1] + (n_joints * c,))


# --------------------------------------------------------------------------------------
# conv stacks — processing_emage_audio.py:178-261
# --------------------------------------------------------------------------------------
def _conv_k3(sd, name, h):
    return F.conv1d(h, sd[name + ".weight"], sd[name + ".bias"], stride=1, padding=1)


def _resblock(sd, name, h):
    """ResBlock, processing_emage_audio.py:178-187: conv, LeakyReLU(0.2), conv, + input."""
    r = F.leaky_relu(_conv_k3(sd, name + ".model.0", h), 0.2)
    return _conv_k3(sd, name + ".model.2", r) + h


def conv_encoder(sd, prefix, x, n_layer):
    """VQEncoderV5 / VQEncoderV6.forward, processing_emage_audio.py:189-235.  x: (B,T,C)."""
    h = x.permute(0, 2, 1)
    for i in range(n_layer):
        h = F.leaky_relu(_conv_k3(sd, f"{prefix}.main.{3 * i}", h), 0.2)
        h = _resblock(sd, f"{prefix}.main.{3 * i + 2}", h)
    return h.permute(0, 2, 1)


def conv_decoder(sd, prefix, z, n_layer):
    """VQDecoderV5.forward, processing_emage_audio.py:237-261.  z: (B,T,L) -> (B,T,dim)."""
    h = z.permute(0, 2, 1)
    for i in range(2):
        h = _resblock(sd, f"{prefix}.main.{i}", h)
    for i in range(n_layer):
        h = F.leaky_relu(_conv_k3(sd, f"{prefix}.main.{2 + 2 * i}", h), 0.2)
    h = _conv_k3(sd, f"{prefix}.main.{2 + 2 * n_layer}", h)
    return h.permute(0, 2, 1)


# --------------------------------------------------------------------------------------
# Quantizer — processing_emage_audio.py:135-170, modeling_emage_audio.py:60-70
# --------------------------------------------------------------------------------------
def vq_distances(z_flat, codebook):
    """d = sum(z^2) + sum(e^2) - 2 z e^T, processing_emage_audio.py:147-148,161-162."""
    return (torch.sum(z_flat ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1)
            - 2 * torch.matmul(z_flat, codebook.t()))


def vq_nearest(z, codebook):
    """Quantizer.map2index: argmin over the codebook (first minimum), int64 (B,T)."""
    d = vq_distances(z.contiguous().view(-1, codebook.shape[1]), codebook)
    return torch.argmin(d, dim=1).reshape(z.shape[0], -1)


def vq_lookup(idx, codebook):
    """Quantizer.get_codebook_entry, processing_emage_audio.py:166-170."""
    return codebook[idx.reshape(-1)].view(idx.shape + (codebook.shape[1],)).contiguous()


def vq_forward(z, codebook, beta):
    """Quantizer.forward, processing_emage_audio.py:144-156 (values only, no autograd):
    returns (loss, z_q, indices, perplexity)."""
    d = vq_distances(z.contiguous().view(-1, codebook.shape[1]), codebook)
    idx = torch.argmin(d, dim=1)
    z_q = codebook[idx].view(z.shape)
    loss = torch.mean((z_q - z) ** 2) + beta * torch.mean((z_q - z) ** 2)
    e_mean = torch.mean(F.one_hot(idx, codebook.shape[0]).type(z.dtype), dim=0)
    perplexity = torch.exp(-torch.sum(e_mean * torch.log(e_mean + 1e-10)))
    return loss, z_q, idx, perplexity


class VQVAE:
    """EmageVQVAEConv, modeling_emage_audio.py:34-70, over a flat state dict."""

    def __init__(self, sd, cfg):
        self.sd, self.cfg, self.n = sd, cfg, cfg.vae_layer
        self.codebook = sd["quantizer.embedding.weight"]

    def encode(self, x):
        return conv_encoder(self.sd, "encoder", x, self.n)

    def map2index(self, x):            # :47-50
        return vq_nearest(self.encode(x), self.codebook)

    def map2latent(self, x):           # :51-55
        return vq_lookup(self.map2index(x), self.codebook)

    def decode(self, index):           # :56-59
        return conv_decoder(self.sd, "decoder", vq_lookup(index, self.codebook), self.n)

    def decode_from_latent(self, latent):   # :60-70
        return self.decode(vq_nearest(latent, self.codebook))

    def forward(self, x):              # :42-46
        pre = self.encode(x)
        loss, z_q, _, ppl = vq_forward(pre, self.codebook, self.cfg.vae_quantizer_lambda)
        return {"poses_feat": z_q, "embedding_loss": loss, "perplexity": ppl,
                "rec_pose": conv_decoder(self.sd, "decoder", z_q, self.n)}


class VAE:
    """EmageVAEConv, modeling_emage_audio.py:19-32."""

    def __init__(self, sd, cfg):
        self.sd, self.n = sd, cfg.vae_layer

    def forward(self, x):
        return {"rec_pose": conv_decoder(self.sd, "decoder", conv_encoder(self.sd, "encoder", x, self.n), self.n)}


class VQModel:
    """EmageVQModel, modeling_emage_audio.py:72-205."""

    def __init__(self, face, upper, hands, lower, global_ae):
        self.face, self.upper, self.hands, self.lower, self.global_ae = face, upper, hands, lower, global_ae

    def split_inputs(self, rot6d, expression, tar_contact=None, tar_trans=None):   # :97-108
        bs, t, j6 = rot6d.shape
        r = rot6d.reshape(bs, t, j6 // 6, 6)
        face = torch.cat([r[:, :, JAW_JOINT], expression], dim=2)
        upper = r[:, :, UPPER_JOINTS].reshape(bs, t, 78)
        hands = r[:, :, HANDS_JOINTS].reshape(bs, t, 180)
        lower = r[:, :, LOWER_JOINTS].reshape(bs, t, 54)
        tar_contact = torch.zeros(bs, t, 4) if tar_contact is None else tar_contact
        tar_trans = torch.zeros(bs, t, 3) if tar_trans is None else tar_trans
        return dict(face=face, upper=upper, hands=hands, lower=torch.cat([lower, tar_trans, tar_contact], dim=2))

    def map2index(self, rot6d, expression, tar_contact=None, tar_trans=None):       # :110-116
        x = self.split_inputs(rot6d, expression, tar_contact, tar_trans)
        return {p: getattr(self, p).map2index(x[p]) for p in ("face", "upper", "hands", "lower")}

    def map2latent(self, rot6d, expression, tar_contact=None, tar_trans=None):      # :118-124
        x = self.split_inputs(rot6d, expression, tar_contact, tar_trans)
        return {p: getattr(self, p).map2latent(x[p]) for p in ("face", "upper", "hands", "lower")}

    def _part(self, model, index, latent):
        if index is not None:
            return model.decode(index)
        if latent is not None:
            return model.decode_from_latent(latent)
        return None

    def decode(self, face_index=None, upper_index=None, hands_index=None, lower_index=None,
               face_latent=None, upper_latent=None, hands_latent=None, lower_latent=None,
               get_global_motion=False, ref_trans=None):                            # :126-193
        for t_ in (face_index, upper_index, hands_index, lower_index, face_latent, upper_latent, hands_latent, lower_latent):
            if t_ is not None:
                bs, t = t_.shape[:2]
                break
        to_aa = lambda r6: rotation_6d_to_axis_angle(r6.reshape(bs, t, -1, 6)).reshape(bs, t, -1)
        face_mix = self._part(self.face, face_index, face_latent)
        if face_mix is not None:
            face_jaw, expression = rotation_6d_to_axis_angle(face_mix[:, :, :6]), face_mix[:, :, 6:]
        else:
            face_jaw, expression = torch.zeros(bs, t, 3), torch.zeros(bs, t, 100)
        upper_6d = self._part(self.upper, upper_index, upper_latent)
        upper = to_aa(upper_6d) if upper_6d is not None else torch.zeros(bs, t, 39)
        hands_6d = self._part(self.hands, hands_index, hands_latent)
        hands = to_aa(hands_6d) if hands_6d is not None else torch.zeros(bs, t, 90)
        lower_mix = self._part(self.lower, lower_index, lower_latent)
        if lower_mix is not None:
            lower, transfoot = to_aa(lower_mix[:, :, :-7]), lower_mix[:, :, -7:]
        else:
            lower, transfoot = torch.zeros(bs, t, 27), torch.zeros(bs, t, 7)
            lower_mix = torch.cat([axis_angle_to_rotation_6d(lower.reshape(bs, t, -1, 3)).reshape(bs, t, -1), transfoot], dim=-1)
        aa = scatter_joints(upper, UPPER_JOINTS) + scatter_joints(hands, HANDS_JOINTS) + scatter_joints(lower, LOWER_JOINTS)
        aa[:, :, JAW_JOINT * 3:JAW_JOINT * 3 + 3] = face_jaw
        rot6d = axis_angle_to_rotation_6d(aa.reshape(bs, t, 55, 3)).reshape(bs, t, 330)
        out = dict(expression=expression, all_motion4inference=torch.cat([rot6d, transfoot], dim=2),
                   motion_axis_angle=aa, trans=None)
        if get_global_motion:
            out["trans"] = self.get_global_motion(lower_mix, ref_trans)
        return out

    def get_global_motion(self, lower_body, ref_trans):                             # :195-205
        v = self.global_ae.forward(lower_body)["rec_pose"][:, :, 54:57]
        if ref_trans.dim() == 2:
            ref_trans = ref_trans.unsqueeze(0).repeat(v.shape[0], 1, 1)
        x = velocity2position(v[:, :, 0:1], 1 / 30, ref_trans[:, 0, 0:1])
        z = velocity2position(v[:, :, 2:3], 1 / 30, ref_trans[:, 0, 2:3])
        return torch.cat([x, v[:, :, 1:2], z], dim=-1)


# --------------------------------------------------------------------------------------
# WavEncoder — processing_emage_audio.py:263-314
# --------------------------------------------------------------------------------------
WAV_BLOCKS = [(5, 1600, True), (6, 0, True), (1, 7, False), (6, 0, True), (1, 7, False), (3, 0, True)]  # :301-306


def _bn_eval(sd, name, h):
    return F.batch_norm(h, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], training=False, eps=1e-5)


def wav_encoder(sd, prefix, wav):
    """WavEncoder.forward (eval-mode BatchNorm): (B,L) -> (B,T',out_dim).  BasicBlock.forward is
    act2(bn2(conv2(act1(bn1(conv1(x))))) + shortcut), LeakyReLU default slope 0.01 (:283-294)."""
    h = wav.unsqueeze(1)
    for i, (stride, pad, has_ds) in enumerate(WAV_BLOCKS):
        b = f"{prefix}.feat_extractor.{i}"
        y = F.conv1d(h, sd[b + ".conv1.weight"], sd[b + ".conv1.bias"], stride=stride, padding=pad)
        y = F.leaky_relu(_bn_eval(sd, b + ".bn1", y), 0.01)
        y = F.conv1d(y, sd[b + ".conv2.weight"], sd[b + ".conv2.bias"], stride=1, padding=7)
        y = _bn_eval(sd, b + ".bn2", y)
        if has_ds:
            h = _bn_eval(sd, b + ".downsample.1",
                         F.conv1d(h, sd[b + ".downsample.0.weight"], sd[b + ".downsample.0.bias"], stride=stride, padding=pad))
        h = F.leaky_relu(y + h, 0.01)
    return h.transpose(1, 2)


# --------------------------------------------------------------------------------------
# transformer layers (torch nn.Transformer{En,De}coderLayer defaults: post-norm, ReLU,
# eps 1e-5, packed in_proj ordered Q,K,V, scale 1/sqrt(head_dim), no masks) — SURVEY §3.2
# --------------------------------------------------------------------------------------
def _linear(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def _ln(sd, name, x):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps=1e-5)


def mlp(sd, name, x):
    """MLP, processing_emage_audio.py:316-326: fc2(LeakyReLU(0.1)(fc1(x)))."""
    return _linear(sd, name + ".fc2", F.leaky_relu(_linear(sd, name + ".fc1", x), 0.1))


def mha(sd, name, q_in, kv_in, n_head=N_HEAD):
    """nn.MultiheadAttention forward, batch-first restatement: q_in (B,Tq,d), kv_in (B,Tk,d)."""
    w, b = sd[name + ".in_proj_weight"], sd[name + ".in_proj_bias"]
    d = w.shape[1]
    hd = d // n_head
    bsz, tq, _ = q_in.shape
    tk = kv_in.shape[1]
    q = F.linear(q_in, w[:d], b[:d]).view(bsz, tq, n_head, hd).transpose(1, 2)
    k = F.linear(kv_in, w[d:2 * d], b[d:2 * d]).view(bsz, tk, n_head, hd).transpose(1, 2)
    v = F.linear(kv_in, w[2 * d:], b[2 * d:]).view(bsz, tk, n_head, hd).transpose(1, 2)
    p = torch.softmax((q * (1.0 / math.sqrt(hd))) @ k.transpose(-1, -2), dim=-1)
    o = (p @ v).transpose(1, 2).reshape(bsz, tq, d)
    return _linear(sd, name + ".out_proj", o)


def encoder_layer(sd, name, x):
    x = _ln(sd, name + ".norm1", x + mha(sd, name + ".self_attn", x, x))
    return _ln(sd, name + ".norm2", x + _linear(sd, name + ".linear2", F.relu(_linear(sd, name + ".linear1", x))))


def decoder_layer(sd, name, x, memory):
    x = _ln(sd, name + ".norm1", x + mha(sd, name + ".self_attn", x, x))
    x = _ln(sd, name + ".norm2", x + mha(sd, name + ".multihead_attn", x, memory))
    return _ln(sd, name + ".norm3", x + _linear(sd, name + ".linear2", F.relu(_linear(sd, name + ".linear1", x))))


def decoder_stack(sd, name, n_layers, tgt, memory):
    for i in range(n_layers):
        tgt = decoder_layer(sd, f"{name}.layers.{i}", tgt, memory)
    return tgt


# --------------------------------------------------------------------------------------
# EmageAudioModel — modeling_emage_audio.py:207-490
# --------------------------------------------------------------------------------------
OUT_KEYS = ("rec_face", "rec_upper", "rec_hands", "rec_lower", "cls_face", "cls_upper", "cls_hands", "cls_lower")


class AudioModel:
    def __init__(self, sd, cfg):
        self.sd, self.cfg = sd, cfg

    def forward(self, audio, speaker_id, masked_motion, mask, use_audio=True):
        """EmageAudioModel.forward, modeling_emage_audio.py:265-341 (eval mode: dropout off)."""
        sd = self.sd
        x = torch.where(mask == 1, sd["mask_embedding"].expand_as(masked_motion), masked_motion)   # :267-268
        body_hint = conv_encoder(sd, "motion_encoder", x, 3)                                     # :271
        hint_body = mlp(sd, "bodyhints_body", body_hint)
        hint_face = mlp(sd, "bodyhints_face", body_hint)
        a_face = wav_encoder(sd, "audio_encoder_face", audio)                                    # :275
        a_body = wav_encoder(sd, "audio_encoder_body", audio)                                    # :276
        if a_face.shape[1] > hint_face.shape[1]:                                                 # :278-279
            a_face = a_face[:, :hint_face.shape[1]]
        if a_body.shape[1] > hint_face.shape[1]:                                                 # :280-281 re-trims FACE (sic)
            a_face = a_face[:, :hint_face.shape[1]]
        t = a_face.shape[1]
        pe = sd["position_embeddings.pe"]
        spk_body = sd["speaker_embedding_body.weight"][speaker_id].repeat(1, t, 1)               # :285
        spk_face = sd["speaker_embedding_face.weight"][speaker_id].repeat(1, t, 1)               # :286
        # face branch :288-294
        mem_face = _linear(sd, "audio_face_motion_proj", torch.cat([a_face, hint_face], dim=2))
        face = decoder_stack(sd, "face_motion_decoder", 4, spk_face + pe[:, :t], mem_face)
        rec_face = _linear(sd, "face_out_proj", face)
        cls_face = mlp(sd, "face_cls", rec_face)
        # body branch :297-312
        m = _linear(sd, "moton_proj", hint_body)
        m = spk_body + (m + pe[:, :m.shape[1]])
        m = encoder_layer(sd, "motion_self_encoder.layers.0", m)
        mem_body = _linear(sd, "audio_body_motion_proj", a_body)
        m = (m + spk_body) + pe[:, :m.shape[1]]
        cross = decoder_stack(sd, "audio_motion_cross_attn", 8, m, mem_body)
        if not use_audio:
            cross = cross * 0.0
        m = m + cross
        # part heads :315-330
        lat = {p: mlp(sd, f"motion2latent_{p}", m) for p in ("upper", "hands", "lower")}
        others = {"upper": ("hands", "lower"), "hands": ("upper", "lower"), "lower": ("upper", "hands")}
        out = {"rec_face": rec_face, "cls_face": cls_face}
        for p in ("upper", "hands", "lower"):
            refine = decoder_layer(sd, f"body_motion_decoder_{p}.layers.0", lat[p] + spk_body,
                                   lat[others[p][0]] + lat[others[p][1]])
            out[f"rec_{p}"] = _linear(sd, f"motion_out_proj_{p}", lat[p] + refine)
            out[f"cls_{p}"] = mlp(sd, f"motion_cls_{p}", out[f"rec_{p}"])
        return {k: out[k] for k in OUT_KEYS}

    def select_codes(self, net_out):
        """Latent-vs-index routing of modeling_emage_audio.py:398-410 / test_emage_audio.py:34-42."""
        cfg = self.cfg
        kw = {}
        for p, l, c in (("face", cfg.lf, cfg.cf), ("upper", cfg.lu, cfg.cu), ("hands", cfg.lh, cfg.ch), ("lower", cfg.ll, cfg.cl)):
            kw[f"{p}_latent"] = net_out[f"rec_{p}"] if (l > 0 and c == 0) else None
            kw[f"{p}_index"] = torch.max(F.log_softmax(net_out[f"cls_{p}"], dim=2), dim=2)[1] if c > 0 else None
        return kw

    def inference(self, audio, speaker_id, vq_model, masked_motion=None, mask=None):
        """EmageAudioModel.inference, modeling_emage_audio.py:343-490."""
        cfg = self.cfg
        bs = audio.shape[0]
        length = audio.shape[1] * 30 // 16000                                                    # :345
        motion = torch.cat([axis_angle_to_rotation_6d(torch.zeros(bs, length, 55, 3)).reshape(bs, length, -1),
                            torch.zeros(bs, length, 7)], dim=-1)                                 # :348-351
        if masked_motion is not None:
            motion[:, :masked_motion.shape[1]] = masked_motion
        full_mask = torch.ones_like(motion)
        if mask is not None:
            full_mask[:, :mask.shape[1]] = mask
        window, pre = cfg.pose_length, cfg.seed_frames
        rounds, remain = (length - pre) // (window - pre), (length - pre) % (window - pre)       # :364-368
        spf = 16000 // 30
        chunks = {k: [] for k in OUT_KEYS}
        last = motion[:, :pre]

        def run_window(start, end):
            w_mask = full_mask[:, start:end].clone()
            w_motion = motion[:, start:end].clone()
            w_motion[:, :pre] = torch.where(w_mask[:, :pre] == 0, motion[:, start:start + pre], last)   # :386-390
            w_mask[:, :pre] = 0
            a = audio[:, start * spf:start * spf + (end - start) * spf]                          # :393-394
            net = self.forward(a, speaker_id, w_motion, w_mask, use_audio=True)
            return net, vq_model.decode(**self.select_codes(net))

        for i in range(rounds):                                                                  # :380-426
            start = i * (window - pre)
            net, dec = run_window(start, start + window)
            last = dec["all_motion4inference"][:, -pre:]
            for k in OUT_KEYS:
                chunks[k].append(net[k][:, :-pre])
        if remain > pre:                                                                         # :428-470
            start = rounds * (window - pre)
            net, _ = run_window(start, start + pre + remain)
            for k in OUT_KEYS:
                chunks[k].append(net[k])
        return {k: torch.cat(chunks[k], dim=1) for k in OUT_KEYS}


def infer_clip(model: AudioModel, vq_model: VQModel, audio, speaker_id=None):
    """The timed body of /root/reference/test_emage_audio.py:16-53: inference, code selection,
    full-length decode with global translation.  Returns (poses (B,T,165), expressions (B,T,100),
    trans (B,T,3)) as numpy arrays."""
    bs = audio.shape[0]
    if speaker_id is None:
        speaker_id = torch.zeros(bs, 1, dtype=torch.long)
    with torch.no_grad():
        latent = model.inference(audio, speaker_id, vq_model)
        pred = vq_model.decode(**model.select_codes(latent), get_global_motion=True, ref_trans=torch.zeros(1, 3))
    return (pred["motion_axis_angle"].numpy(), pred["expression"].numpy(), pred["trans"].numpy())


def beat_format_arrays(poses, expressions, trans):
    """The npz schema of beat_format_save, /root/reference/emage_utils/motion_io.py:154-163
    (upsample == 1 because pose_fps == 30): returns the dict np.savez receives."""
    return dict(betas=np.zeros((300,), dtype=poses.dtype), poses=poses, expressions=expressions, trans=trans,
                model="smplx2020", gender="neutral", mocap_frame_rate=30)

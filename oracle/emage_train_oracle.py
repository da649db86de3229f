"""CPU restatement of the EMAGE TRAINING step — TEST INFRASTRUCTURE (groundwork for the
next rows of the hot path: no product code imports this file).

What it restates, and where the reference does it (read-only at /root/reference):
  * EmageAudioModel.forward in TRAIN mode            models/emage_audio/modeling_emage_audio.py:265-341 (M)
      - BatchNorm1d with batch statistics + running-stat update in the two WavEncoders
        (models/emage_audio/processing_emage_audio.py:262-314 (P); nn.SyncBatchNorm at train_emage_audio.py:252 (T) is
        the same arithmetic over the GLOBAL batch: at world size 1 it is this function)
      - dropout p = 0.1 of PeriodicPositionalEncoding (P:329-343) and of torch's nn.TransformerEncoderLayer /
        nn.TransformerDecoderLayer defaults (M:246-262): attention-probability dropout, dropout1/2/3 on the sub-layer
        outputs, dropout inside the FFN
  * the losses                                       T:106-130 (MSE on latents x l*, NLL(log_softmax) on indices x c*)
  * the three-pass step and its mask schedule        T:132-180
  * Adam                                             T:258-265 (torch.optim.Adam, lr 1.5e-4 constant, betas .9/.999,
                                                     eps 1e-8, weight decay 0: configs/emage_audio.yaml:63-78)

Randomness.  The reference draws every dropout mask and the random motion mask from torch's global CPU generator.
This restatement issues the SAME draws (same op, same tensor shape and memory layout, same order), so after
`torch.manual_seed(s)` both produce identical masks and the step can be compared number for number
(tests/test_train_oracle.py).  That is why the transformer part works on (T, B, d) tensors like the reference (its
layers are batch_first=False, M:289,300,309) and why the attention-probability dropout is drawn on a contiguous
(B, H, Tq, Tk) tensor (torch's scaled_dot_product_attention math path, which nn.MultiheadAttention takes in train
mode with need_weights=False).

Quirks kept on purpose (they are the reference's behaviour):
  * `mask_ratio = (iteration / 135 * 400) * 0.95 + 0.05` (T:163; from iteration 1 on the ratio exceeds 1: everything masked)
  * `clip_grad_norm_` runs BEFORE `backward()` (T:176-178), i.e. on gradients that zero_grad() just cleared: no effect
  * `cf: 0` in the config: the face classification loss is multiplied by 0 but still evaluated
  * BatchNorm running statistics are updated three times per step (three forwards)

Parity status: pinned against the reference run live in the build container (tests/test_train_oracle.py) and
through tests/golden/train_step_b2.npz (generated from the REAL reference by tests/golden/make_golden_train.py).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import emage_oracle as orc

DROPOUT_P = 0.1          # PeriodicPositionalEncoding default (P:329) and nn.Transformer*Layer default
BN_MOMENTUM = 0.1        # nn.BatchNorm1d default
N_HEAD = orc.N_HEAD


# --------------------------------------------------------------------------------------
# WavEncoder, train mode
# --------------------------------------------------------------------------------------
def _bn_train(sd, name, h, new_stats):
    """nn.BatchNorm1d.forward in training mode on (B, C, L): normalise with the batch mean / BIASED variance,
    update running_mean / running_var (UNBIASED variance, momentum 0.1) and num_batches_tracked.
    The updated buffers go to `new_stats` (read back by the next forward of the same step)."""
    rm = new_stats.get(name + ".running_mean", sd[name + ".running_mean"]).detach().clone()
    rv = new_stats.get(name + ".running_var", sd[name + ".running_var"]).detach().clone()
    y = F.batch_norm(h, rm, rv, sd[name + ".weight"], sd[name + ".bias"], training=True, momentum=BN_MOMENTUM, eps=1e-5)
    new_stats[name + ".running_mean"], new_stats[name + ".running_var"] = rm, rv
    nbt = new_stats.get(name + ".num_batches_tracked", sd.get(name + ".num_batches_tracked", torch.zeros((), dtype=torch.long)))
    new_stats[name + ".num_batches_tracked"] = nbt.detach().clone() + 1
    return y


def wav_encoder_train(sd, prefix, wav, new_stats):
    """WavEncoder.forward (P:296-314) with train-mode BatchNorm; block order inside BasicBlock.forward (P:283-294):
    conv1, bn1, act1, conv2, bn2, then the downsample branch (conv, bn), add, act2."""
    h = wav.unsqueeze(1)
    for i, (stride, pad, has_ds) in enumerate(orc.WAV_BLOCKS):
        b = f"{prefix}.feat_extractor.{i}"
        y = F.conv1d(h, sd[b + ".conv1.weight"], sd[b + ".conv1.bias"], stride=stride, padding=pad)
        y = F.leaky_relu(_bn_train(sd, b + ".bn1", y, new_stats), 0.01)
        y = F.conv1d(y, sd[b + ".conv2.weight"], sd[b + ".conv2.bias"], stride=1, padding=7)
        y = _bn_train(sd, b + ".bn2", y, new_stats)
        if has_ds:
            h = _bn_train(sd, b + ".downsample.1",
                          F.conv1d(h, sd[b + ".downsample.0.weight"], sd[b + ".downsample.0.bias"], stride=stride, padding=pad),
                          new_stats)
        h = F.leaky_relu(y + h, 0.01)
    return h.transpose(1, 2)


# --------------------------------------------------------------------------------------
# transformer layers, train mode, (T, B, d) layout
# --------------------------------------------------------------------------------------
def _drop(x, p):
    """nn.Dropout / F.dropout in training mode: x * bernoulli(1-p) / (1-p).  The mask is an `empty_like(x)` filled from
    the global generator in MEMORY order, so which element gets which draw depends on x's strides: every caller below
    hands over a tensor with the same layout the reference's module sees at that point."""
    return F.dropout(x, p, training=True) if p > 0 else x


def mha_train(sd, name, q_in, kv_in, p, n_head=N_HEAD):
    """nn.MultiheadAttention.forward(need_weights=False) in training mode, q_in (Tq, B, d), kv_in (Tk, B, d).
    torch projects, reshapes to (B, H, T, hd) and calls scaled_dot_product_attention; its CPU math path scales q and k
    by sqrt(scale) each, applies softmax, draws the dropout on the (B, H, Tq, Tk) probabilities, multiplies by v."""
    w, b = sd[name + ".in_proj_weight"], sd[name + ".in_proj_bias"]
    d = w.shape[1]
    hd = d // n_head
    tq, bsz, _ = q_in.shape
    tk = kv_in.shape[0]
    q = F.linear(q_in, w[:d], b[:d])
    k = F.linear(kv_in, w[d:2 * d], b[d:2 * d])
    v = F.linear(kv_in, w[2 * d:], b[2 * d:])
    # (T, B, d) -> (T, B*H, hd) -> (B*H, T, hd) -> (B, H, T, hd), exactly torch's view / transpose sequence
    q = q.reshape(tq, bsz * n_head, hd).transpose(0, 1).reshape(bsz, n_head, tq, hd)
    k = k.reshape(tk, bsz * n_head, hd).transpose(0, 1).reshape(bsz, n_head, tk, hd)
    v = v.reshape(tk, bsz * n_head, hd).transpose(0, 1).reshape(bsz, n_head, tk, hd)
    s = math.sqrt(1.0 / math.sqrt(hd))                     # sqrt of the 1/sqrt(hd) scale, applied to both operands
    attn = torch.softmax((q * s) @ (k * s).transpose(-2, -1), dim=-1)
    attn = _drop(attn, p)
    o = (attn @ v).permute(2, 0, 1, 3).contiguous().view(tq * bsz, d)
    return F.linear(o, sd[name + ".out_proj.weight"], sd[name + ".out_proj.bias"]).view(tq, bsz, d)


def _ffn_train(sd, name, x, p):
    """linear2(dropout(relu(linear1(x))))  — TransformerEncoderLayer._ff_block / TransformerDecoderLayer._ff_block."""
    return orc._linear(sd, name + ".linear2", _drop(F.relu(orc._linear(sd, name + ".linear1", x)), p))


def encoder_layer_train(sd, name, x, p):
    """nn.TransformerEncoderLayer (norm_first=False): x = norm1(x + dropout1(sa(x))); x = norm2(x + dropout2(ff(x)))."""
    x = orc._ln(sd, name + ".norm1", x + _drop(mha_train(sd, name + ".self_attn", x, x, p), p))
    return orc._ln(sd, name + ".norm2", x + _drop(_ffn_train(sd, name, x, p), p))


def decoder_layer_train(sd, name, x, memory, p):
    """nn.TransformerDecoderLayer (norm_first=False): self-attention, cross-attention, FFN, each followed by its
    dropout, the residual add and a LayerNorm."""
    x = orc._ln(sd, name + ".norm1", x + _drop(mha_train(sd, name + ".self_attn", x, x, p), p))
    x = orc._ln(sd, name + ".norm2", x + _drop(mha_train(sd, name + ".multihead_attn", x, memory, p), p))
    return orc._ln(sd, name + ".norm3", x + _drop(_ffn_train(sd, name, x, p), p))


def decoder_stack_train(sd, name, n_layers, tgt, memory, p):
    for i in range(n_layers):
        tgt = decoder_layer_train(sd, f"{name}.layers.{i}", tgt, memory, p)
    return tgt


def _ppe_train(sd, x, p):
    """PeriodicPositionalEncoding.forward (P:341-343) on (B, T, d): dropout(x + pe[:, :T]).  No .contiguous(): at
    M:308-309 x is the (T, B, d) encoder output permuted back, the sum keeps that memory order and so does the mask."""
    return _drop(x + sd["position_embeddings.pe"][:, :x.shape[1]], p)


def _tbd(x):
    return x.permute(1, 0, 2)


# --------------------------------------------------------------------------------------
# EmageAudioModel.forward, train mode (M:265-341) — same statement order as the reference: it fixes the RNG order
# --------------------------------------------------------------------------------------
def forward_train(sd, audio, speaker_id, masked_motion, mask, use_audio=True, p=DROPOUT_P, new_stats=None):
    new_stats = {} if new_stats is None else new_stats
    x = torch.where(mask == 1, sd["mask_embedding"].expand_as(masked_motion), masked_motion)             # M:267-268
    body_hint = orc.conv_encoder(sd, "motion_encoder", x, 3)                                            # M:271
    hint_body = orc.mlp(sd, "bodyhints_body", body_hint)
    hint_face = orc.mlp(sd, "bodyhints_face", body_hint)
    a_face = wav_encoder_train(sd, "audio_encoder_face", audio, new_stats)                              # M:275
    a_body = wav_encoder_train(sd, "audio_encoder_body", audio, new_stats)                              # M:276
    if a_face.shape[1] > hint_face.shape[1]:                                                            # M:278-279
        a_face = a_face[:, :hint_face.shape[1]]
    if a_body.shape[1] > hint_face.shape[1]:                                                            # M:280-281 (sic: re-trims FACE)
        a_face = a_face[:, :hint_face.shape[1]]
    t = a_face.shape[1]
    spk_body = sd["speaker_embedding_body.weight"][speaker_id].repeat(1, t, 1)                          # M:285
    spk_face = sd["speaker_embedding_face.weight"][speaker_id].repeat(1, t, 1)                          # M:286
    # face branch, M:288-294
    mem_face = orc._linear(sd, "audio_face_motion_proj", torch.cat([a_face, hint_face], dim=2))
    face_proj = _ppe_train(sd, spk_face, p)
    face = _tbd(decoder_stack_train(sd, "face_motion_decoder", 4, _tbd(face_proj), _tbd(mem_face), p))
    rec_face = orc._linear(sd, "face_out_proj", face)
    cls_face = orc.mlp(sd, "face_cls", rec_face)
    # body branch, M:297-312
    m = _ppe_train(sd, orc._linear(sd, "moton_proj", hint_body), p)
    m = spk_body + m
    m = _tbd(encoder_layer_train(sd, "motion_self_encoder.layers.0", _tbd(m), p))
    mem_body = orc._linear(sd, "audio_body_motion_proj", a_body)
    m = _ppe_train(sd, m + spk_body, p)
    cross = _tbd(decoder_stack_train(sd, "audio_motion_cross_attn", 8, _tbd(m), _tbd(mem_body), p))
    if not use_audio:
        cross = cross * 0.0
    m = m + cross
    # part heads, M:315-330: the three latents first, then the three refinement layers in upper, hands, lower order
    lat = {q: orc.mlp(sd, f"motion2latent_{q}", m) for q in ("upper", "hands", "lower")}
    others = {"upper": ("hands", "lower"), "hands": ("upper", "lower"), "lower": ("upper", "hands")}
    refine = {}
    for q in ("upper", "hands", "lower"):
        refine[q] = _tbd(decoder_layer_train(sd, f"body_motion_decoder_{q}.layers.0", _tbd(lat[q]) + _tbd(spk_body),
                                             _tbd(lat[others[q][0]] + lat[others[q][1]]), p))
    out = {"rec_face": rec_face, "cls_face": cls_face}
    for q in ("upper", "hands", "lower"):
        out[f"rec_{q}"] = orc._linear(sd, f"motion_out_proj_{q}", lat[q] + refine[q])
    for q in ("upper", "hands", "lower"):
        out[f"cls_{q}"] = orc.mlp(sd, f"motion_cls_{q}", out[f"rec_{q}"])
    return {k: out[k] for k in orc.OUT_KEYS}


# --------------------------------------------------------------------------------------
# losses, T:106-130
# --------------------------------------------------------------------------------------
def rec_loss(pred, latent_gt, cfg):
    """sum over parts of l_part * mean((rec_part - latent_part)^2)   (F.mse_loss, reduction 'mean')."""
    return sum(getattr(cfg, "l" + q[0]) * torch.mean((pred[f"rec_{q}"] - latent_gt[q]) ** 2)
               for q in ("upper", "lower", "hands", "face"))


def cls_loss(pred, index_gt, cfg):
    """sum over parts of c_part * NLLLoss(log_softmax(cls_part), index_part): the mean over (B, T) of -log p[target]."""
    total = 0.0
    for q in ("upper", "lower", "hands", "face"):
        logp = torch.log_softmax(pred[f"cls_{q}"], dim=2)
        picked = torch.gather(logp, 2, index_gt[q].unsqueeze(-1)).squeeze(-1)
        total = total + getattr(cfg, "c" + q[0]) * (-picked.mean())
    return total


# --------------------------------------------------------------------------------------
# the step, T:132-180
# --------------------------------------------------------------------------------------
def targets(vq: "orc.VQModel", motion_aa, expressions, trans, foot_contact):
    """Ground-truth conversion at the top of train_val_fn (T:146-152): axis-angle -> rot6d, frozen VQ-VAEs ->
    code indices and quantised latents, and the 337-channel motion the model is conditioned on."""
    bs, t, jc = motion_aa.shape
    rot6d = orc.axis_angle_to_rotation_6d(motion_aa.reshape(bs, t, jc // 3, 3)).reshape(bs, t, jc // 3 * 6)
    with torch.no_grad():
        index = vq.map2index(rot6d, expressions, tar_contact=foot_contact, tar_trans=trans)
        latent = vq.map2latent(rot6d, expressions, tar_contact=foot_contact, tar_trans=trans)
    return index, latent, torch.cat([rot6d, trans, foot_contact], dim=-1)


def train_step_losses(sd, vq, cfg, batch, iteration, p=DROPOUT_P):
    """The three forwards of one step and their six losses.  Returns (loss dict incl. "all", updated BatchNorm buffers).
    Order of random draws: forward 1 (seed mask), torch.rand for the random mask, forward 2, forward 3."""
    index, latent, masked_motion = targets(vq, batch["motion"], batch["expressions"], batch["trans"], batch["foot_contact"])
    bs, t = masked_motion.shape[:2]
    speaker_id = torch.zeros(bs, 1, dtype=torch.long)
    stats = {}
    mask = torch.ones_like(masked_motion)
    mask[:, :cfg.seed_frames] = 0
    loss = {}
    pred = forward_train(sd, batch["audio"], speaker_id, masked_motion, mask, True, p, stats)
    loss["rec_seed"], loss["cls_seed"] = rec_loss(pred, latent, cfg), cls_loss(pred, index, cfg)
    mask_ratio = (iteration / 135 * 400) * 0.95 + 0.05                                              # T:163 (sic)
    mask = (torch.rand(bs, t, cfg.pose_dims + 3 + 4) < mask_ratio).float()
    pred = forward_train(sd, batch["audio"], speaker_id, masked_motion, mask, True, p, stats)
    loss["rec_audio"], loss["cls_audio"] = rec_loss(pred, latent, cfg), cls_loss(pred, index, cfg)
    pred = forward_train(sd, batch["audio"], speaker_id, masked_motion, mask, False, p, stats)
    loss["rec_mask"], loss["cls_mask"] = rec_loss(pred, latent, cfg), cls_loss(pred, index, cfg)
    loss["all"] = sum(loss.values())
    return loss, stats


def trainable_keys(sd):
    """State-dict entries that are nn.Parameters of EmageAudioModel (everything but BatchNorm buffers and the
    positional table)."""
    skip = (".running_mean", ".running_var", ".num_batches_tracked")
    return [k for k in sd if not k.endswith(skip) and k != "position_embeddings.pe"]


def adam_update(param, grad, exp_avg, exp_avg_sq, step, lr=1.5e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam (no amsgrad, not decoupled): returns (new param, new exp_avg, new exp_avg_sq) for step number
    `step` (1-based)."""
    if weight_decay != 0:
        grad = grad + weight_decay * param
    exp_avg = beta1 * exp_avg + (1 - beta1) * grad
    exp_avg_sq = beta2 * exp_avg_sq + (1 - beta2) * grad * grad
    bias1, bias2 = 1 - beta1 ** step, 1 - beta2 ** step
    denom = exp_avg_sq.sqrt() / math.sqrt(bias2) + eps
    return param - (lr / bias1) * exp_avg / denom, exp_avg, exp_avg_sq


def train_step(sd, vq, cfg, batch, iteration, opt_state=None, seed=None, p=DROPOUT_P, lr=1.5e-4):
    """One optimisation step on a flat state dict.  Returns (losses, grads, new state dict, new optimiser state).
    Parameters that take no part in the forward (the two template layers `transformer_en_layer.*` and
    `audio_motion_cross_attn_layer.*` that nn.TransformerEncoder / nn.TransformerDecoder deep-copied, M:246-262)
    have no gradient and are left untouched, as torch.optim.Adam does for `grad is None`."""
    if seed is not None:
        torch.manual_seed(seed)
    keys = trainable_keys(sd)
    work = dict(sd)
    for k in keys:
        work[k] = sd[k].detach().clone().requires_grad_(True)
    losses, stats = train_step_losses(work, vq, cfg, batch, iteration, p)
    losses["all"].backward()
    grads = {k: work[k].grad for k in keys if work[k].grad is not None}
    opt_state = {} if opt_state is None else opt_state
    new_sd, new_opt = dict(sd), {}
    for k, g in grads.items():
        st = opt_state.get(k, dict(step=0, exp_avg=torch.zeros_like(g), exp_avg_sq=torch.zeros_like(g)))
        step = st["step"] + 1
        with torch.no_grad():
            new_p, m, v = adam_update(sd[k], g, st["exp_avg"], st["exp_avg_sq"], step, lr=lr)
        new_sd[k], new_opt[k] = new_p, dict(step=step, exp_avg=m, exp_avg_sq=v)
    for k, v in stats.items():
        new_sd[k] = v
    return {k: float(v) for k, v in losses.items()}, grads, new_sd, new_opt

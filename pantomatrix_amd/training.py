"""EmageAudioModel.forward in TRAIN mode on the MI355X kernels — the forward half of the EMAGE training step
(/root/reference/train_emage_audio.py:130-204 drives three such forwards per step).

What differs from the inference forward (`modeling_emage_audio.EmageAudioModel.forward`), all of it behaviour of torch
modules inside /root/reference/models/emage_audio/modeling_emage_audio.py (M:) and processing_emage_audio.py (P:):

  * the two WavEncoders (P:262-314) run nn.BatchNorm1d with BATCH statistics: the convolutions are launched with their raw
    weights (no BatchNorm fold), `emage_bn_stats` reduces each conv output over all clips and positions (float64 sums) and
    updates the running statistics the way torch does, `emage_bn_apply` normalises and applies what follows inside
    BasicBlock.forward (P:283-294: LeakyReLU; the shortcut, itself batch-normalised in the downsample blocks);
  * dropout (p = 0.1) at every site torch has one — PeriodicPositionalEncoding (P:341-343), and inside
    nn.TransformerEncoderLayer / nn.TransformerDecoderLayer (M:238-262): on the attention probabilities
    (`emage_attention_dropout`), on each sub-layer output before the residual add, inside the FFN (`emage_mul_add`).

Randomness is an INPUT: `dropout_masks` is the list of mask tensors (values bernoulli / (1 - p)) in the order the reference
draws them and in the logical shapes its modules see — (B, T, d) for the positional encodings, (B, H, Tq, Tk) for attention
probabilities, (T, B, C) for the layers' dropouts (its layers are batch_first=False).  The parity tests take the masks from
the CPU oracle's replay of the reference's generator draws (oracle/emage_train_oracle.py), so outputs, BatchNorm buffers
and losses can be compared number for number; a training loop would fill them from a device generator.

No CPU fallback: every arithmetic step is a launch into libemage_hip.so.  Backward, Adam and the gradient exchange
(pantomatrix_amd/dist.py has the bucket plan) are the next pieces of this row.
"""
from __future__ import annotations

import torch

from . import ops, spec
from ._lib import BF16, F32
from .modeling_emage_audio import OUT_KEYS, _Ctx, _WAV_TAPS, _conv_encoder, _rup

BN_MOMENTUM = 0.1          # nn.BatchNorm1d default
DROPOUT_P = 0.1            # PeriodicPositionalEncoding (P:329) and nn.Transformer*Layer defaults


def dropout_mask_count():
    """Masks one train-mode forward consumes: 3 positional encodings, 15 decoder layers x 6, 1 encoder layer x 4."""
    return 3 + 6 * (spec.N_FACE_LAYERS + spec.N_CROSS_LAYERS + 3) + 4


class _Masks:
    def __init__(self, masks, dev):
        self.masks, self.i, self.dev = list(masks), 0, dev

    def take(self, shape):
        if self.i >= len(self.masks):
            raise RuntimeError(f"dropout_masks: {len(self.masks)} masks given, the forward needs {dropout_mask_count()}")
        m = self.masks[self.i]
        self.i += 1
        if tuple(m.shape) != tuple(shape):
            raise RuntimeError(f"dropout mask {self.i - 1}: shape {tuple(m.shape)}, the reference draws {tuple(shape)} here")
        return m.to(device=self.dev, dtype=torch.float32).contiguous()


class TrainForward:
    """Callable train-mode forward of an `EmageAudioModel` (f16x3 or fp32 precision)."""

    def __init__(self, model):
        if model.precision == "bf16":
            raise ValueError("the training forward runs in the fp32-storage precisions (f16x3 / fp32)")
        self.model = model

    # ---- packing of what the inference pack does not hold: un-folded WavEncoder convolutions ----------------------------
    def _train_pack(self, pk):
        if "train.wav_in" in pk.w:
            return
        m = self.model
        w0, b0 = [], []
        for enc in ("audio_encoder_face", "audio_encoder_body"):
            for i, (cin, cout, stride, pad, ds) in enumerate(m._wav_blocks()):
                base = f"{enc}.feat_extractor.{i}"
                if i == 0:
                    for conv in (base + ".conv1", base + ".downsample.0"):
                        w, b = pk.folded(conv, None)
                        w0.append(w.reshape(cout, _WAV_TAPS))
                        b0.append(b)
                else:
                    pk.conv(base + ".conv1.raw", base + ".conv1", extra=(base + ".downsample.0", None) if ds else None)
                pk.conv(base + ".conv2.raw", base + ".conv2")
                for bn in ("bn1", "bn2") + (("downsample.1",) if ds else ()):
                    pk.w[f"{base}.{bn}.affine"] = (pk.f32(f"{base}.{bn}.weight"), pk.f32(f"{base}.{bn}.bias"))
        pk.w["train.wav_in"] = dict(w=torch.cat(w0, 0).float().contiguous(), b=torch.cat(b0).float().contiguous())
        pk.w["train.ones"] = torch.ones(pk.w["train.wav_in"]["w"].shape[0], dtype=torch.float32, device=pk.device)

    # ---- BatchNorm bookkeeping ----------------------------------------------------------------------------------------------
    def _bn(self, cx, name, x, new_stats):
        """Batch statistics of conv output `x` (M, C) for BatchNorm `name`; the running buffers advance in `new_stats`."""
        params = self.model._flat_params()
        rm = new_stats.get(name + ".running_mean", params[name + ".running_mean"]).detach().to(cx.dev, torch.float32).clone()
        rv = new_stats.get(name + ".running_var", params[name + ".running_var"]).detach().to(cx.dev, torch.float32).clone()
        stats = ops.bn_stats(x, rm, rv, BN_MOMENTUM)
        new_stats[name + ".running_mean"], new_stats[name + ".running_var"] = rm, rv
        nbt = new_stats.get(name + ".num_batches_tracked", params.get(name + ".num_batches_tracked", torch.zeros((), dtype=torch.long)))
        new_stats[name + ".num_batches_tracked"] = nbt.detach().clone() + 1
        return stats

    def _wav_encoder(self, cx, enc, e, audio, b, new_stats):
        """WavEncoder.forward (P:296-314) with train-mode BatchNorm -> (B*T', audio_f) fp32, T'."""
        m = self.model
        blocks = m._wav_blocks()
        lens = m._wav_lengths(audio.shape[1])
        k, q = _WAV_TAPS, blocks[0][1]
        w_in = cx.pk.w["train.wav_in"]
        x, lin = None, None
        for i, (cin, cout, stride, pad, ds) in enumerate(blocks):
            base = f"{enc}.feat_extractor.{i}"
            lout = lens[i]
            rows = b * lout
            if i == 0:
                r = slice(e * 2 * q, (e + 1) * 2 * q)
                y = torch.empty(rows, 2 * q, dtype=torch.float32, device=cx.dev)
                ops.wav_conv_in(F32, audio, w_in["w"][r], w_in["b"][r], cx.pk.w["train.ones"][:2 * q], y, lout, stride, pad)
            else:
                ent = cx.pk.w[base + ".conv1.raw"]
                y, _ = cx.gemm(x, base + ".conv1.raw", conv=(stride, pad, lin, lout), m=rows, n_store=_rup(ent["n"]))
            c1 = y[:, :cout]
            g1, b1 = cx.pk.w[base + ".bn1.affine"]
            y1 = torch.empty(rows, _rup(cout), dtype=torch.float32, device=cx.dev) if _rup(cout) != cout else torch.empty(rows, cout, dtype=torch.float32, device=cx.dev)
            if y1.shape[1] != cout:
                y1.zero_()
            ops.bn_apply(c1, self._bn(cx, base + ".bn1", c1, new_stats), g1, b1, y1[:, :cout], slope=0.01)
            c2, _ = cx.gemm(y1, base + ".conv2.raw", conv=(1, k // 2, lout, lout), m=rows, n_store=_rup(cout))
            c2 = c2[:, :cout]
            g2, b2 = cx.pk.w[base + ".bn2.affine"]
            st2 = self._bn(cx, base + ".bn2", c2, new_stats)                           # P:287-288, before the downsample branch (P:289-290)
            out = torch.empty(rows, cout, dtype=torch.float32, device=cx.dev)
            if ds:
                cds = y[:, cout:2 * cout]
                gd, bd = cx.pk.w[base + ".downsample.1.affine"]
                ops.bn_apply(c2, st2, g2, b2, out, slope=0.01, sc=cds, sc_bn=(*self._bn(cx, base + ".downsample.1", cds, new_stats), gd, bd))
            else:
                ops.bn_apply(c2, st2, g2, b2, out, slope=0.01, sc=x[:, :cout])
            x, lin = out, lout
        return x, lens[-1]

    # ---- transformer pieces, train mode ---------------------------------------------------------------------------------------
    def _drop_add(self, cx, o, masks, t, res=None):
        """res + dropout(o) with the mask the reference draws on the (T, B, C) tensor."""
        m, c = o.shape
        mk = masks.take((t, m // t, c)).view(m, c)
        return ops.mul_add(o, mk, res, mask_t_rows=t)

    def _mha(self, cx, masks, q, k, vt, vt_rows, b, tq, tk):
        d, h = self.model.config.hidden_size, spec.N_HEAD
        att = cx.lo(b * tq, d)
        ops.attention_dropout(cx.gdt, q, k, vt, vt_rows, att, b, h, tq, tk, d // h, masks.take((b, h, tq, tk)))
        return att

    def _self_attn(self, cx, masks, name, x, b, t):
        d = self.model.config.hidden_size
        qk = cx.lo(b * t, 2 * d)
        vt = cx.vt_buffer(b, d, t)
        cx.gemm(x, name + ".sa.qkv", out=qk, out_t=vt, t_col0=2 * d, t_rows=t)
        att = self._mha(cx, masks, qk[:, :d], qk[:, d:], vt, d, b, t, t)
        o, _ = cx.gemm(att, name + ".sa.out")
        return self._drop_add(cx, o, masks, t, res=x)

    def _ffn(self, cx, masks, name, x, t):
        f, _ = cx.gemm(x, name + ".ff1", slope=0.0)
        f = self._drop_add(cx, f, masks, t)
        o, _ = cx.gemm(f, name + ".ff2")
        return self._drop_add(cx, o, masks, t, res=x)

    def _encoder_layer(self, cx, masks, name, x, b, t):
        ln = self.model._ln
        x = ln(cx, name + ".norm1", self._self_attn(cx, masks, name, x, b, t))
        return ln(cx, name + ".norm2", self._ffn(cx, masks, name, x, t))

    def _decoder_layer(self, cx, masks, name, x, b, t, mem_k, mem_vt, vt_rows, tk):
        ln = self.model._ln
        x = ln(cx, name + ".norm1", self._self_attn(cx, masks, name, x, b, t))
        q, _ = cx.gemm(x, name + ".ca.q")
        att = self._mha(cx, masks, q, mem_k, mem_vt, vt_rows, b, t, tk)
        o, _ = cx.gemm(att, name + ".ca.out")
        x = ln(cx, name + ".norm2", self._drop_add(cx, o, masks, t, res=x))
        return ln(cx, name + ".norm3", self._ffn(cx, masks, name, x, t))

    def _ppe(self, cx, masks, x, b, t):
        """PeriodicPositionalEncoding.forward (P:341-343): dropout(x + pe[:, :T]) on (B*T, d) rows."""
        m, d = x.shape
        y = cx.lo(m, d)
        ops.add(cx.dt, x, cx.pk.w["pe"][:t], out=y, mod_b=t)
        return ops.mul_add(y, masks.take((b, t, d)).view(m, d))

    # ---- the forward --------------------------------------------------------------------------------------------------------
    def __call__(self, audio, speaker_id, masked_motion, mask, dropout_masks, use_audio=True, new_stats=None):
        """-> (dict of the 8 (B, T, 256) fp32 outputs, new_stats).  `new_stats` carries the BatchNorm running buffers from one
        forward of a step to the next (as oracle.emage_train_oracle.forward_train does); it is not written into the model."""
        model = self.model
        c = model.config
        cx = _Ctx(model._engine())
        pk, dev = cx.pk, cx.dev
        self._train_pack(pk)
        new_stats = {} if new_stats is None else new_stats
        masks = _Masks(dropout_masks, dev)
        b, t, cm = masked_motion.shape
        m = b * t
        d, mf, af = c.hidden_size, c.motion_f, c.audio_f
        nf, nc = spec.N_FACE_LAYERS, spec.N_CROSS_LAYERS
        audio = audio.to(device=dev, dtype=torch.float32).contiguous()
        motion3 = masked_motion.to(device=dev, dtype=torch.float32).contiguous()
        mask3 = mask.to(device=dev, dtype=torch.float32).contiguous()

        # masked motion -> spatial hints (M:267-273)
        x0 = ops.pack_motion(cx.dt, motion3, mask3, pk.w["mask_emb"], _rup(cm))
        hint, _ = _conv_encoder(cx, "motion_encoder", x0, t, spec.MOTION_ENC_LAYERS, mf, False)
        hh, _ = cx.gemm(hint, "bodyhints.fc1", slope=0.1)
        memcat = cx.lo(m, af + mf)                                                   # [audio2face | body_hint_face] (M:288)
        cx.gemm(hh[:, :d], "bodyhints_face.fc2", out=memcat[:, af:])
        hint_body, _ = cx.gemm(hh[:, d:], "bodyhints_body.fc2")

        # the two WavEncoders, batch statistics (M:275-281)
        a_face, ta = self._wav_encoder(cx, "audio_encoder_face", 0, audio, b, new_stats)
        a_body, _ = self._wav_encoder(cx, "audio_encoder_body", 1, audio, b, new_stats)
        if ta < t:
            raise RuntimeError(f"Sizes of tensors must match: audio features {ta} frames vs motion {t} frames")
        memcat[:, :af] = a_face.view(b, ta, af)[:, :t].reshape(m, af)                # M:278-281: the FACE features are trimmed to T

        sid = speaker_id.to(dev).reshape(b, 1).expand(b, t)
        spk_body = ops.gather_rows(pk.w["spk_body"], sid, F32)
        spk_face = ops.gather_rows(pk.w["spk_face"], sid, F32)
        out = {}

        # face branch (M:288-294)
        mem_face, _ = cx.gemm(memcat, "audio_face_motion_proj")
        face = self._ppe(cx, masks, spk_face, b, t)
        fkk, fvt = model._memory_kv(cx, "face.kv_all", mem_face, b, t, nf)
        for i in range(nf):
            face = self._decoder_layer(cx, masks, f"face_motion_decoder.layers.{i}", face, b, t, fkk[:, i * d:(i + 1) * d], fvt[:, i * d:], nf * d, t)
        rec_lo, out["rec_face"] = cx.gemm(face, "face_out_proj", want="both")
        hc, _ = cx.gemm(rec_lo, "face_cls.fc1", slope=0.1)
        _, out["cls_face"] = cx.gemm(hc, "face_cls.fc2", want="f32")

        # body branch (M:297-312)
        x, _ = cx.gemm(hint_body, "moton_proj")
        x = self._ppe(cx, masks, x, b, t)
        xs = cx.lo(m, d)
        ops.add(cx.dt, x, spk_body, out=xs)
        x = self._encoder_layer(cx, masks, "motion_self_encoder.layers.0", xs, b, t)
        mem_body, _ = cx.gemm(a_body, "audio_body_motion_proj")                      # M:303
        xs = cx.lo(m, d)
        ops.add(cx.dt, x, spk_body, out=xs)
        base = self._ppe(cx, masks, xs, b, t)                                        # M:304-305
        x = base
        bk, bvt = model._memory_kv(cx, "cross.kv_all", mem_body, b, ta, nc)
        for i in range(nc):
            x = self._decoder_layer(cx, masks, f"audio_motion_cross_attn.layers.{i}", x, b, t, bk[:, i * d:(i + 1) * d], bvt[:, i * d:], nc * d, ta)
        fea = cx.lo(m, d)
        if use_audio:
            ops.add(cx.dt, base, x, out=fea)                                         # motion_fea + cross (M:310-312)
        else:
            fea.copy_(base)                                                          # cross * 0 (M:311)

        # part latents, refinement layers, heads (M:315-330)
        parts = ("upper", "hands", "lower")
        others = {"upper": ("hands", "lower"), "hands": ("upper", "lower"), "lower": ("upper", "hands")}
        hl, _ = cx.gemm(fea, "motion2latent.fc1", slope=0.1)
        lat = {p: cx.gemm(hl[:, i * d:(i + 1) * d], f"motion2latent_{p}.fc2")[0] for i, p in enumerate(parts)}
        refine = {}
        for p in parts:
            tgt, mem_lo = cx.lo(m, d), cx.lo(m, d)
            ops.add(cx.dt, lat[p], spk_body, out=tgt)
            ops.add(cx.dt, lat[others[p][0]], lat[others[p][1]], out=mem_lo)
            name = f"body_motion_decoder_{p}.layers.0"
            k1, vt1 = model._memory_kv(cx, name + ".ca.kv", mem_lo, b, t, 1)
            refine[p] = self._decoder_layer(cx, masks, name, tgt, b, t, k1, vt1, d, t)
        rec_los = {}
        for p in parts:
            sum_lo = cx.lo(m, d)
            ops.add(cx.dt, lat[p], refine[p], out=sum_lo)
            rec_los[p], out[f"rec_{p}"] = cx.gemm(sum_lo, f"motion_out_proj_{p}", want="both")
        for p in parts:
            hc, _ = cx.gemm(rec_los[p], f"motion_cls_{p}.fc1", slope=0.1)
            _, out[f"cls_{p}"] = cx.gemm(hc, f"motion_cls_{p}.fc2", want="f32")
        if masks.i != len(masks.masks):
            raise RuntimeError(f"dropout_masks: {len(masks.masks)} masks given, the forward consumed {masks.i}")
        return {key: out[key].view(b, t, -1) for key in OUT_KEYS}, new_stats


# ======================================================================================
# the three forwards of a step and their six losses (train_emage_audio.py:132-172)
# ======================================================================================
def targets(vq, motion_aa, expressions, trans, foot_contact):
    """Top of train_val_fn (T:146-152): axis-angle -> rot-6D, the frozen VQ-VAEs' code indices and quantised latents, and the
    337-channel motion the model is conditioned on — all on the device (`vq` is the product EmageVQModel)."""
    bs, t, jc = motion_aa.shape
    dev = vq.vq_model_face.device
    aa = motion_aa.to(device=dev, dtype=torch.float32).reshape(bs * t * (jc // 3), 3).contiguous()
    rot6d = ops.axis_angle_to_rot6d(aa).reshape(bs, t, jc // 3 * 6)
    expressions, trans, foot_contact = (x.to(device=dev, dtype=torch.float32) for x in (expressions, trans, foot_contact))
    index = vq.map2index(rot6d, expressions, tar_contact=foot_contact, tar_trans=trans)
    latent = vq.map2latent(rot6d, expressions, tar_contact=foot_contact, tar_trans=trans)
    return index, latent, torch.cat([rot6d, trans, foot_contact], dim=-1)


def losses(cfg, pred, index_gt, latent_gt, workspace=None):
    """(rec, cls) of T:106-130 as float64 device scalars: sum_q l_q * mse(rec_q, latent_q), sum_q c_q * NLL(log_softmax(cls_q), index_q)."""
    dev = pred["rec_face"].device
    ws = ops.loss_workspace(dev) if workspace is None else workspace
    rec, cls = torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)
    for q in ("upper", "lower", "hands", "face"):
        b, t, k = pred[f"rec_{q}"].shape
        ops.mse_loss(pred[f"rec_{q}"].reshape(b * t, k), latent_gt[q].reshape(b * t, k), getattr(cfg, "l" + q[0]), rec, ws)
        ops.nll_loss(pred[f"cls_{q}"].reshape(b * t, -1), index_gt[q].reshape(-1).contiguous(), getattr(cfg, "c" + q[0]), cls, ws)
    return rec, cls


def step_losses(fwd: TrainForward, vq, batch, iteration, dropout_masks, random_mask):
    """The seed / audio / mask forwards of one step (T:132-172) -> (dict of the six losses + "all" as Python floats, BatchNorm
    buffers after the three forwards).  dropout_masks: three mask lists (one per forward); random_mask: the (B, T, 337)
    {0, 1} motion mask of forwards 2 and 3, `torch.rand(...) < mask_ratio` in the reference (T:163-165)."""
    cfg = fwd.model.config
    index, latent, masked_motion = targets(vq, batch["motion"], batch["expressions"], batch["trans"], batch["foot_contact"])
    bs, t = masked_motion.shape[:2]
    speaker_id = torch.zeros(bs, 1, dtype=torch.long, device=masked_motion.device)
    seed_mask = torch.ones_like(masked_motion)
    seed_mask[:, :cfg.seed_frames] = 0
    stats, out = {}, {}
    ws = ops.loss_workspace(masked_motion.device)
    for tag, mask, use_audio, masks in (("seed", seed_mask, True, dropout_masks[0]), ("audio", random_mask, True, dropout_masks[1]),
                                        ("mask", random_mask, False, dropout_masks[2])):
        pred, stats = fwd(batch["audio"], speaker_id, masked_motion, mask, masks, use_audio=use_audio, new_stats=stats)
        out["rec_" + tag], out["cls_" + tag] = losses(cfg, pred, index, latent, ws)
    res = {k: float(v) for k, v in out.items()}
    res["all"] = sum(res.values())
    return res, stats

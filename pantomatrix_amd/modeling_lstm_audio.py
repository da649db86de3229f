"""MI355X-native DisCo and CaMN — the two LSTM speech-to-gesture generators next to EMAGE.

Same public surface as
  D: /root/reference/models/disco_audio/modeling_disco_audio.py   DiscoAudioModel.forward D:199-266
  C: /root/reference/models/camn_audio/modeling_camn_audio.py     CamnAudioModel.forward  C:223-280
(`forward(audio, speaker_id, seed_frames=4, seed_motion=None, return_axis_angle=True)` -> dict; the same state-dict keys,
`pantomatrix_amd/spec.py`), eval mode.  Kernel plan, all through libemage_hip.so:

* WavEncoder (32..128 channels, 15 fps): `emage_wav_conv_in` + implicit-GEMM convs (the EMAGE WavEncoder code, narrower);
* MLPs, the selector, and the LSTM input projection x W_ih^T of ALL time steps (both directions stacked, one launch per
  layer): `emage_gemm`, writing straight into the column blocks of the next operand (no concatenation copies);
* the recurrence: `emage_lstm_step_pair`, one launch per time step carrying BOTH directions (h W_hh^T on MFMA with the LSTM
  cell in the epilogue; the forward direction at step s beside the backward one at step T-1-s); a whole forward is captured
  in a hipGraph by `LstmClipRunner`;
* rot-6D -> axis-angle -> 55 SMPL-X joints: `emage_rot6d_scatter`.
Precision: "f16x3" (default, fp32 storage + split-f16 MFMA: fp32-grade, what the parity tests use) or "fp32".
"""
from __future__ import annotations

import torch

from . import ops, spec
from ._lib import F16X3, H2
from .configuration_emage_audio import _AttrConfig
from .modeling_emage_audio import _EmageModule, _WavEncoderMixin, _Ctx, _rup


class DiscoAudioConfig(_AttrConfig):
    model_type = "disco_audio"


class CamnAudioConfig(_AttrConfig):
    model_type = "camn_audio"


class _LstmAudioModel(_WavEncoderMixin, _EmageModule):
    _supports_h2 = False          # float32 activations: the recurrence kernels split h_{t-1} once per step in LDS themselves

    def __init__(self, config):
        super().__init__(config)
        self._dt = F16X3
        self.pose_rep = getattr(config, "pose_rep", "smplx")
        self.pair_convs = True                 # WavEncoder: 32-channel blocks through the 64-channel kernels on position pairs (A/B switch)
        self.persistent_lstm = True            # one launch per LSTM layer (csrc/lstmseq.hip) instead of one per time step; same bits
        self.h2_input_projection = False       # A/B switch (round 5, measured: no gain — profiles/r05_lstm_input_projection_h2_ab.txt: CaMN 70.7 / 72.2 vs 72.1 / 72.1 ms,
                                               # DisCo 8.05 / 8.23 vs 7.98 / 8.03): True runs the per-layer input projection x W_ih^T on pre-split
                                               # EMAGE_H2 operands (one cast of x per layer) instead of splitting A inside the GEMM
        self._sync = {}                        # scratch of the persistent recurrences, see _lstm_sync

    def set_precision(self, precision: str):
        if precision == "bf16":
            raise ValueError("the LSTM models run in f16x3 or fp32 (their recurrences are not bf16-safe at the parity bar)")
        return super().set_precision(precision)

    def _wav_blocks(self):
        return spec.lstm_wav_encoder_blocks()

    # ---- packing ---------------------------------------------------------------------
    def _pack_lstm(self, pk, name, n_layer):
        """torch's packed gate rows [i | f | g | o] (each H rows) regrouped per hidden unit (row 4u + g), so one lane of
        `emage_lstm_step` owns the four gates of a unit; both directions' input projections stacked along N."""
        hid = self.config.hidden_size
        perm = torch.arange(4 * hid).view(4, hid).t().reshape(-1)                 # new row 4u + g  <-  old row g*H + u
        for k in range(n_layer):
            w_ih, b_all = [], []
            for d, suffix in enumerate(("", "_reverse")):
                p = f"{name}.weight_ih_l{k}{suffix}"
                w_ih.append(pk.p[p].float()[perm])
                b_all.append((pk.p[f"{name}.bias_ih_l{k}{suffix}"].float() + pk.p[f"{name}.bias_hh_l{k}{suffix}"].float())[perm])
                w, kp, ws = pk._pack_mat(pk.p[f"{name}.weight_hh_l{k}{suffix}"].float()[perm])
                pk.w[f"{name}.hh.{k}.{d}"] = dict(w=w, ws=ws)
            wcat = torch.cat(w_ih, 0)
            w, kp, ws = pk._pack_mat(wcat)
            pk.w[f"{name}.ih.{k}"] = dict(w=w, b=torch.cat(b_all).contiguous(), n=w.shape[0], cp=kp, taps=1, k_real=w_ih[0].shape[1], ws=ws)
            if pk.dt == F16X3 and self.h2_input_projection:      # the same weights as an EMAGE_H2 operand (set the switch before the first forward)
                wpad = torch.nn.functional.pad(wcat, (0, kp - wcat.shape[1])) if kp != wcat.shape[1] else wcat
                w2, ws2 = pk._operand(wpad, dt=H2)
                pk.w[f"{name}.ih.{k}.h2"] = dict(pk.w[f"{name}.ih.{k}"], w=w2, ws=ws2, dt=H2)

    def _pack_common(self, pk):
        self._pack_wav_encoders(pk, ("audio_encoder",))
        if self.config.speaker_f > 0:
            pk.w["spk"] = pk.f32("speaker_embedding.weight")
        jm = getattr(self.config, "joint_mask", "local_upper")
        joints = spec.LOCAL_UPPER_JOINTS if jm == "local_upper" else list(range(1, 55))
        slot = torch.full((55,), -1, dtype=torch.int32)
        for i, j in enumerate(joints):
            slot[j] = i
        pk.w["slot_of_joint"] = slot.to(pk.device)

    # ---- building blocks ---------------------------------------------------------------
    def _audio_feat(self, cx, audio, b, dest=None):
        """WavEncoder of all clips -> (B*T, audio_f) fp32, written into `dest` (a 2-D view) when given.  Long clips at
        large batch (CaMN: 256 x 28 s) exceed what one launch's operand may span: the clips are encoded in groups."""
        lens = self._wav_lengths(audio.shape[1])
        t = lens[-1]
        if dest is None:
            dest = torch.empty(b * t, self.config.audio_f, dtype=torch.float32, device=cx.dev)
        pair_rows = self.pair_convs and self._wav_pairs_ok(lens)       # 32-channel blocks as 64-channel convolutions over position pairs
        step = self._wav_clip_chunk(cx, b, lens, width=32 if pair_rows else None)
        for c0 in range(0, b, step):
            n = min(step, b - c0)
            if pair_rows:
                y1, sc = self._wav_first_layer_pairs(cx, audio[c0:c0 + n], lens)
                self._wav_encoder_chain_pairs(cx, "audio_encoder", y1, sc, n, lens, dest=dest[c0 * t:(c0 + n) * t])
            else:
                y0 = self._wav_first_layer(cx, audio[c0:c0 + n], lens)
                self._wav_encoder_chain(cx, "audio_encoder", 0, y0, n, lens, dest=dest[c0 * t:(c0 + n) * t])
        return dest, t

    def _seed_src_map(self, t, seed_motion):
        """Which frame of the reference's padded seed tensor each of the T input frames shows (D:229-242 verbatim,
        including the shorter-seed quirk: `cat(seed, seed[:, -diff:])` with diff < 0 appends the frames from |diff| on)."""
        if seed_motion is None:
            return list(range(t))
        t_m = seed_motion.shape[1]
        if t_m == t:
            return list(range(t))
        diff = t_m - t
        if diff > 0:
            return list(range(t))
        m = list(range(t_m)) + list(range(-diff, t_m))
        if len(m) != t:
            raise RuntimeError(f"Sizes of tensors must match except in dimension 2. Expected size {t} but got size {len(m)} "
                               f"(seed motion of {t_m} frames vs {t} audio frames)")
        return m

    def _mlp(self, cx, name, x, out=None, out_f32=None):
        """MLP (D:152-164): Linear + LeakyReLU(0.1) + Linear; the result lands in `out` (and `out_f32`) when given."""
        h, _ = cx.gemm(x, name + ".fc1", slope=0.1)
        return cx.gemm(h, name + ".fc2", out=out, out_f32=out_f32)[0]

    def _bilstm(self, cx, name, x, b, t, n_layer):
        """nn.LSTM(bidirectional=True, batch_first=True), eval: x (B*T, Cp) -> (B*T, 2H) [forward | backward].
        f16x3 with H in {256, 512}: the recurrence of a layer is ONE persistent launch (`ops.lstm_layer`, csrc/lstmseq.hip);
        otherwise (exact-fp32 mode, other sizes, `persistent_lstm = False`) one launch per time step — the same bits."""
        hid = self.config.hidden_size
        persistent = self.persistent_lstm and ops.lstm_layer_supported(cx.gdt, hid)
        zeros = None if persistent else torch.zeros(b, hid, dtype=torch.float32, device=cx.dev)
        for k in range(n_layer):
            e_h2 = cx.pk.w.get(f"{name}.ih.{k}.h2") if self.h2_input_projection else None
            if e_h2 is not None and x.shape[1] == e_h2["cp"]:
                _, gx = cx.gemm(ops.cast_pad(H2, x, x.shape[1]), f"{name}.ih.{k}.h2", want="f32")       # (B*T, 8H) fp32: all steps, both directions
            else:
                gx, _ = cx.gemm(x, f"{name}.ih.{k}")
            hseq = torch.empty(b * t, 2 * hid, dtype=torch.float32, device=cx.dev)
            g3, h3 = gx.view(b, t, -1), hseq.view(b, t, 2 * hid)
            w0, w1 = cx.pk.w[f"{name}.hh.{k}.0"], cx.pk.w[f"{name}.hh.{k}.1"]
            if persistent:
                sync = self._lstm_sync(name, k, b, hid, cx.dev)
                ops.lstm_layer(cx.gdt, g3, (w0["w"], w1["w"]), (w0["ws"], w1["ws"]), h3, sync)
                x = hseq
                continue
            cstate = torch.zeros(2, b, hid, dtype=torch.float32, device=cx.dev)
            prev = [zeros, zeros]
            for s in range(t):                                                       # step s of the forward direction runs beside
                sf, sb = s, t - 1 - s                                                # step t-1-s of the backward one, in ONE launch
                cur = [h3[:, sf, :hid], h3[:, sb, hid:]]
                ops.lstm_step_pair(cx.gdt, (prev[0], w0["w"], g3[:, sf, :4 * hid], cstate[0], cur[0], w0["ws"]),
                                   (prev[1], w1["w"], g3[:, sb, 4 * hid:8 * hid], cstate[1], cur[1], w1["ws"]))
                prev = cur
            x = hseq
        return x

    def _lstm_sync(self, name, k, b, hid, dev):
        """Scratch of the persistent recurrence, one tensor per (LSTM, layer, batch): allocated once (stable under graph replay)."""
        key = (name, k, b, str(dev))
        if key not in self._sync:
            self._sync[key] = ops.lstm_layer_sync(b, hid, dev)
        return self._sync[key]

    def _checked(self, result):
        """End of an eager forward: surface a lost block of the persistent recurrence (under graph capture the runner checks)."""
        if self._sync and not (self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            self.check_kernels()
        return result

    def fold_kernel_health(self, counter):
        """counter (int32 device scalar) += lost blocks of this forward's persistent recurrences (stream-ordered, capturable)."""
        for s in self._sync.values():
            ops.lstm_layer_health(s, counter)

    def check_kernels(self):
        """Raise if a persistent-recurrence launch reported a lost block since the last call (synchronises the device)."""
        for s in self._sync.values():
            ops.lstm_layer_check(s)

    def _lstm_head(self, cx, name, out_name, in_fea, b, t, out=None, out_f32=None):
        hid = self.config.hidden_size
        y = self._bilstm(cx, name, in_fea, b, t, self.config.n_layer)
        s = torch.empty(b * t, hid, dtype=torch.float32, device=cx.dev)
        ops.add(cx.dt, y[:, :hid], y[:, hid:], out=s)                               # forward + backward halves (D:253)
        return self._mlp(cx, out_name, s, out=out, out_f32=out_f32)

    def _axis_angle(self, cx, rot6d2d, b, t):
        return ops.rot6d_scatter(rot6d2d, cx.pk.w["slot_of_joint"]).view(b, t, 165)

    def _tail_inputs(self, cx, in_fea, col0, speaker_id, seed_frames, seed_motion, b, t):
        c = self.config
        key = (t, None if seed_motion is None else seed_motion.shape[1], str(cx.dev))
        if key not in self._templates:             # small constant table, built once per shape (never inside a graph capture)
            self._templates[key] = torch.tensor(self._seed_src_map(t, seed_motion), dtype=torch.int32, device=cx.dev)
        src = self._templates[key]
        sm = None if seed_motion is None else seed_motion.to(device=cx.dev, dtype=torch.float32).contiguous()
        sid = speaker_id.to(cx.dev).reshape(-1).contiguous()
        ops.lstm_inputs(in_fea[:, col0:], cx.pk.w.get("spk"), sid if c.speaker_f > 0 else None, sm, c.pose_dims, seed_frames, src, b, t)



class DiscoAudioModel(_LstmAudioModel):
    config_class = DiscoAudioConfig
    base_model_prefix = "camn_audio"          # (sic) D:174
    _spec_fn = staticmethod(spec.disco_model_spec)

    def _pack(self, pk):
        c = self.config
        self._pack_common(pk)
        pk.linear("enc.fc1", [f"{n}.fc1" for n in ("audio_encoder_c1", "audio_encoder_c2", "audio_encoder_r", "selector")])
        for nm in ("audio_encoder_c1", "audio_encoder_c2", "audio_encoder_r", "selector"):
            pk.linear(nm + ".fc2", [nm + ".fc2"])
        self._pack_lstm(pk, "body_motion_decoder", c.n_layer)
        pk.linear("body_out.fc1", ["body_out.fc1"])
        pk.linear("body_out.fc2", ["body_out.fc2"])

    def forward(self, audio, speaker_id, seed_frames=4, seed_motion=None, return_axis_angle=True):
        """DiscoAudioModel.forward (D:199-266), eval mode: audio (B, L) fp32, speaker_id (B, 1) int64 ->
        {motion (B,T,258), motion_axis_angle (B,T,165), audio_fea_c (B,T,128), audio_fea_r (B,T,128)}."""
        c = self.config
        cx = _Ctx(self._engine())
        dev = cx.dev
        audio = audio.to(device=dev, dtype=torch.float32)
        b = audio.shape[0]
        af, hid = c.audio_f, c.hidden_size
        feat, t = self._audio_feat(cx, audio, b)
        m = b * t
        cin = 2 * af + c.speaker_f + c.pose_dims + 1
        in_fea = torch.empty(m, _rup(cin), dtype=torch.float32, device=dev)          # [c | r | speaker | seed | flag | 0]
        hcat, _ = cx.gemm(feat, "enc.fc1", slope=0.1)                                 # the four MLPs' hidden layers in one launch
        c1, _ = cx.gemm(hcat[:, 0:hid], "audio_encoder_c1.fc2")
        c2, _ = cx.gemm(hcat[:, hid:2 * hid], "audio_encoder_c2.fc2")
        cx.gemm(hcat[:, 2 * hid:3 * hid], "audio_encoder_r.fc2", out=in_fea[:, af:2 * af])
        sel, _ = cx.gemm(hcat[:, 3 * hid:4 * hid], "selector.fc2")
        ops.softmax2_mix(sel, c1, c2, in_fea[:, :af])                                # D:244-247
        self._tail_inputs(cx, in_fea, 2 * af, speaker_id, seed_frames, seed_motion, b, t)
        motion = self._lstm_head(cx, "body_motion_decoder", "body_out", in_fea, b, t)
        out = {"motion": motion.view(b, t, c.pose_dims),
               "motion_axis_angle": self._axis_angle(cx, motion, b, t) if return_axis_angle else None,
               "audio_fea_c": in_fea[:, :af].reshape(b, t, af), "audio_fea_r": in_fea[:, af:2 * af].reshape(b, t, af)}
        return self._checked(out)


class CamnAudioModel(_LstmAudioModel):
    config_class = CamnAudioConfig
    base_model_prefix = "camn_audio"
    _spec_fn = staticmethod(spec.camn_model_spec)

    def _pack(self, pk):
        c = self.config
        self._pack_common(pk)
        self._pack_lstm(pk, "body_motion_decoder", c.n_layer)
        self._pack_lstm(pk, "hands_motion_decoder", c.n_layer)
        for nm in ("body_out", "hands_out"):
            pk.linear(nm + ".fc1", [nm + ".fc1"])
            pk.linear(nm + ".fc2", [nm + ".fc2"])

    def forward(self, audio, speaker_id, seed_frames=4, seed_motion=None, return_axis_angle=True):
        """CamnAudioModel.forward (C:223-280), eval mode, pose_rep "smplx": the body LSTM, then the hands LSTM on
        [inputs | body output]; `motion` is (B, T, 43, 6) with the body joints first (C:272-276)."""
        c = self.config
        cx = _Ctx(self._engine())
        dev = cx.dev
        audio = audio.to(device=dev, dtype=torch.float32)
        b = audio.shape[0]
        af = c.audio_f
        cin = af + c.speaker_f + c.pose_dims + 1
        lens = self._wav_lengths(audio.shape[1])
        t = lens[-1]
        m = b * t
        # one buffer holds the hands LSTM's input [audio | speaker | seed | flag | body (78) | 0]; the body LSTM reads its first
        # `cin` columns — there its zero-padded weight columns meet the (finite) body block, contributing exactly 0
        wide = torch.zeros(m, _rup(cin + c.body_dims), dtype=torch.float32, device=dev)
        self._audio_feat(cx, audio, b, dest=wide[:, :af])
        self._tail_inputs(cx, wide[:, :_rup(cin)], af, speaker_id, seed_frames, seed_motion, b, t)
        motion = torch.empty(m, c.pose_dims, dtype=torch.float32, device=dev)                          # body joints, then hand joints
        # body_out's last GEMM writes the body block twice: into the hands LSTM's input and into the result
        self._lstm_head(cx, "body_motion_decoder", "body_out", wide[:, :_rup(cin)], b, t,
                        out=wide[:, cin:cin + c.body_dims], out_f32=motion[:, :c.body_dims])
        self._lstm_head(cx, "hands_motion_decoder", "hands_out", wide, b, t, out=motion[:, c.body_dims:])
        return self._checked({"motion": motion.view(b, t, c.pose_dims // 6, 6),
                              "motion_axis_angle": self._axis_angle(cx, motion, b, t) if return_axis_angle else None})

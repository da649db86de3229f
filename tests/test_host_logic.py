"""Host-side logic of pantomatrix_amd (weight packing, buffer views, launch sequence, window schedule) checked
on CPU against the oracle, with every C-ABI call replaced by its torch restatement (tests/fake_ops.py).
This is NOT a parity claim for the kernels — those are the `-m gpu` tests — it proves the host drives them with
the right operands."""
import os

import numpy as np
import pytest
import torch

import common
import fake_ops
from oracle import emage_oracle as orc
from pantomatrix_amd import synthetic


def test_forward_window_fp32_host_logic(golden_dir):
    model, _ = common.product_models(precision="fp32")
    omodel, _ = common.oracle_models()
    audio, spk, motion, mask = common.window_inputs(2)
    with fake_ops.installed(), torch.no_grad():
        out = model.forward(audio, spk, motion, mask)
        out_na = model.forward(audio, spk, motion, mask, use_audio=False)
        ref = omodel.forward(audio, spk, motion, mask)
        ref_na = omodel.forward(audio, spk, motion, mask, use_audio=False)
    for k in orc.OUT_KEYS:
        assert out[k].shape == (2, 64, 256)
        assert float((out[k] - ref[k]).abs().max()) < 2e-4, k
        assert float((out_na[k] - ref_na[k]).abs().max()) < 2e-4, k
    g = np.load(os.path.join(golden_dir, "forward_b1.npz"))
    audio, spk, motion, mask = common.window_inputs(1)
    with fake_ops.installed(), torch.no_grad():
        out1 = model.forward(audio, spk, motion, mask)
    for k in orc.OUT_KEYS:
        np.testing.assert_allclose(out1[k].numpy(), g[k], atol=3e-4, rtol=0)


def test_forward_window_bf16_host_logic():
    """bf16 storage only perturbs the result (~0.7 % on latents); catches dtype plumbing errors."""
    model, _ = common.product_models(precision="bf16")
    omodel, _ = common.oracle_models()
    audio, spk, motion, mask = common.window_inputs(1)
    with fake_ops.installed(), torch.no_grad():
        out = model.forward(audio, spk, motion, mask)
        ref = omodel.forward(audio, spk, motion, mask)
    for k in orc.OUT_KEYS:
        rel = float((out[k] - ref[k]).norm() / ref[k].norm())
        assert rel < 0.05, (k, rel)


def test_forward_window_f16x3_host_logic(golden_dir):
    """The split-f16 operand mode: weights packed by ops.split_f16_weights (hi / lo fp16 planes in the kernel's k order,
    power-of-two scale), activations float32 — through the fake's restatement of the three-product arithmetic the window
    must stay at fp32-grade distance from the reference (this is what lets the fast mode keep the VQ code indices)."""
    model, _ = common.product_models(precision="f16x3")
    assert model.precision == "f16x3"
    g = np.load(os.path.join(golden_dir, "forward_b1.npz"))
    audio, spk, motion, mask = common.window_inputs(1)
    with fake_ops.installed(), torch.no_grad():
        out1 = model.forward(audio, spk, motion, mask)
    for k in orc.OUT_KEYS:
        np.testing.assert_allclose(out1[k].numpy(), g[k], atol=3e-4, rtol=0)
    # WavEncoder routing: block 0 of both encoders fused, the 64 / 128-channel stride-1 convs on the LDS-resident slab
    assert fake_ops.CALLS.count("wav_block0") == 2 and fake_ops.CALLS.count("wav_conv_in") == 0
    assert fake_ops.CALLS.count("conv_slab") == 2 * 6            # per encoder: conv2 of blocks 1-4 and conv1 of blocks 2, 4
    model.slab_convs = False
    with fake_ops.installed(), torch.no_grad():
        out2 = model.forward(audio, spk, motion, mask)
    assert fake_ops.CALLS.count("wav_conv_in") == 1 and fake_ops.CALLS.count("conv_slab") == 0
    model.slab_convs = True
    for k in orc.OUT_KEYS:
        assert float((out2[k] - out1[k]).abs().max()) < 1e-5, k
    from pantomatrix_amd import ops
    w = torch.randn(40, 192) * 0.02
    packed, scale = ops.split_f16_weights(w)
    hi, lo = fake_ops.unsplit_f16_weights(packed, 40, 192)
    assert packed.dtype == torch.float32 and packed.shape == (40, 192) and 4096 <= float(w.abs().max()) * scale < 8192
    assert float(((hi + lo) / scale - w).abs().max()) <= 2.0 ** -22 * float(w.abs().max())


def test_layernorm_fold_host_logic(golden_dir):
    """Round 6: the interior LayerNorms of the post-norm layers are FOLDED into the contractions around them (`fold_layernorm`, EMAGE_H2 mode:
    LN(s) W^T + b = rstd (s W'^T - mu c) + b', residuals recomputed from the raw sum and its statistics).  On the CPU stand-ins: 41 of a window's
    47 LayerNorm launches disappear — what stays are the norms whose result leaves the stack (the last layer of each stack, the encoder layer's
    second norm with its post-add) —, the contraction count is unchanged, and the window stays at the reference (golden) and at the unfolded
    result to fp32 rounding; `use_audio=False` (no cross-attention stack) folds the remaining layers."""
    model, _ = common.product_models(precision="f16x3")
    assert model.fold_layernorm
    g = np.load(os.path.join(golden_dir, "forward_b1.npz"))
    audio, spk, motion, mask = common.window_inputs(1)
    outs, calls = {}, {}
    with fake_ops.installed(), torch.no_grad():
        for fold in (True, False):
            model.fold_layernorm = fold
            fake_ops.CALLS.clear()
            outs[fold] = {k: v.clone() for k, v in model.forward(audio, spk, motion, mask).items()}
            calls[fold] = (fake_ops.CALLS.count("layernorm"), fake_ops.CALLS.count("gemm"))
        model.fold_layernorm = True
        fake_ops.CALLS.clear()
        out_na = model.forward(audio, spk, motion, mask, use_audio=False)
        ln_na = fake_ops.CALLS.count("layernorm")
        model.fold_layernorm = False
        ref_na = model.forward(audio, spk, motion, mask, use_audio=False)
    model.fold_layernorm = True
    assert calls[False][0] == 47 and calls[True][0] == 6 and calls[True][1] == calls[False][1], calls
    assert ln_na == 5                       # face 1 + the encoder layer's second norm + the three refinement layers
    for k in orc.OUT_KEYS:
        np.testing.assert_allclose(outs[True][k].numpy(), g[k], atol=3e-4, rtol=0)
        assert float((outs[True][k] - outs[False][k]).abs().max()) < 2e-5, k
        assert float((out_na[k] - ref_na[k]).abs().max()) < 2e-5, k
    # the folded operand entries: W' = W gamma, b' = W beta + b, c = the row sums of W'
    pk = model._packed if model._packed is not None else None
    with fake_ops.installed(), torch.no_grad():
        model.forward(audio, spk, motion, mask)
        pk = model._packed
    name, norm = "face_motion_decoder.layers.1", "face_motion_decoder.layers.1.norm2"
    e = pk.w[name + ".ff1@" + norm]
    p = pk.p
    w, b = p[name + ".linear1.weight"].float(), p[name + ".linear1.bias"].float()
    gm, bt = p[norm + ".weight"].float(), p[norm + ".bias"].float()
    assert e["norm"] == norm and e["n"] == w.shape[0]
    np.testing.assert_allclose(e["c"].numpy(), (w * gm).sum(1).numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(e["b"].numpy(), (b + w @ bt).numpy(), rtol=0, atol=1e-5)
    hi, lo = fake_ops.h2_planes(e["w"], w.shape[1])
    np.testing.assert_allclose(((hi + lo) / e["ws"]).numpy(), (w * gm).numpy(), rtol=0, atol=2.0 ** -21 * float((w * gm).abs().max()))


def test_activation_shift_host_logic(golden_dir):
    """Round 6: the activation images of the EMAGE_H2 mode carry a per-model power-of-two scale 2^(4 - k) in the dtype code of every launch that
    writes or reads one (include/emage_hip.h EMAGE_H2_SHIFT; `model.activation_shift`, default 0 = 16 x: the parity-green form).  On the CPU
    stand-ins: at k = 4 (images hold x: |x| < 65 504 instead of 4 094) the window stays at the golden; the packed weights are the same objects
    (their scales are per tensor); every H2 launch of the forward saw the shifted code; an activation of 10^4 — beyond the default range — comes
    through a k = 4 image and overflows the k = 0 one."""
    from pantomatrix_amd import ops
    from pantomatrix_amd._lib import H2
    model, _ = common.product_models(precision="f16x3")
    g = np.load(os.path.join(golden_dir, "forward_b1.npz"))
    audio, spk, motion, mask = common.window_inputs(1)
    seen = []
    with fake_ops.installed(), torch.no_grad():
        out0 = {k: v.clone() for k, v in model.forward(audio, spk, motion, mask).items()}
        pk0 = model._packed
        model.set_activation_shift(4)
        real = fake_ops._h2s
        try:
            fake_ops._h2s = lambda dtype: (seen.append(dtype), real(dtype))[1]
            out4 = {k: v.clone() for k, v in model.forward(audio, spk, motion, mask).items()}
        finally:
            fake_ops._h2s = real
        assert model._packed is pk0 and pk0.act_shift == 4                  # no re-pack: only the launches' dtype code changes
        model.set_activation_shift(0)
    codes = {c for c in seen if c & 0xff == H2}
    assert codes == {ops.h2_shifted(4)} and ops.h2_shifted(4) == H2 | (4 << 8) and ops.h2_shifted(0) == H2, codes
    for k in orc.OUT_KEYS:
        np.testing.assert_allclose(out4[k].numpy(), g[k], atol=3e-4, rtol=0)
        assert float((out4[k] - out0[k]).abs().max()) < 5e-5, k
    big = torch.full((1, 8), 1.0e4)
    assert not torch.isfinite(ops.h2_unpack(ops.h2_pack(big))).all()                                  # 16 x 10^4 > 65 504: the hi plane overflows
    assert torch.allclose(ops.h2_unpack(ops.h2_pack(big, ops.act_scale(ops.h2_shifted(4))), ops.act_scale(ops.h2_shifted(4))), big)
    with pytest.raises(ValueError):
        model.set_activation_shift(13)


@pytest.mark.parametrize("frames,batch,precision", [(128, 2, "fp32"), (70, 1, "fp32"), (129, 1, "fp32"), (129, 1, "f16x3"), (310, 1, "f16x3"), (40, 1, "f16x3"), (64, 1, "fp32")])
def test_inference_and_decode_host_logic(golden_dir, frames, batch, precision):
    """Whole clip: window schedule, seed carry-over through the VQ decode, tail windows with T+1 audio
    frames, final decode with global translation — against the REFERENCE's golden outputs."""
    g = np.load(os.path.join(golden_dir, f"infer_{frames}f_b{batch}.npz"))
    model, vq = common.product_models(precision=precision)
    audio = synthetic.synthetic_audio(batch, synthetic.samples_for_frames(frames))
    with fake_ops.installed(), torch.no_grad():
        (poses, expr, trans), lat = common.product_infer_clip(model, vq, audio)
        sel = model._select_codes(lat)
    assert poses.shape == g["poses"].shape
    for p in ("upper", "hands", "lower"):
        assert np.array_equal(sel[f"{p}_index"].numpy(), g[f"index_{p}"]), p
    np.testing.assert_allclose(lat["rec_face"].numpy(), g["rec_face"], atol=3e-4, rtol=0)
    np.testing.assert_allclose(poses, g["poses"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(expr, g["expressions"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(trans, g["trans"], atol=1e-3, rtol=0)


@pytest.mark.parametrize("layer", [2, 3])
def test_vq_api_host_logic(golden_dir, layer):
    import pantomatrix_amd as pa
    g = np.load(os.path.join(golden_dir, f"vq_layer{layer}.npz"))
    _, vqc, gc = common.cfg_dicts(vae_layer=layer, global_layer=4 if layer == 2 else 3)
    gen = torch.Generator().manual_seed(11)
    with fake_ops.installed(), torch.no_grad():
        for p in common.PARTS:
            cfg = pa.EmageVQVAEConvConfig(**vqc[p])
            m = pa.EmageVQVAEConv(cfg).set_precision("fp32")
            m.load_state_dict(synthetic.vqvae_state(cfg, p, 0))
            x = torch.randn(2, 40, cfg.vae_test_dim, generator=gen)
            idx = torch.randint(0, 256, (2, 40), generator=gen)
            z = torch.randn(2, 40, 256, generator=gen)
            assert np.array_equal(m.map2index(x).numpy(), g[f"{p}_map2index"])
            np.testing.assert_allclose(m.decode(idx).numpy(), g[f"{p}_decode"], atol=2e-4, rtol=0)
            np.testing.assert_allclose(m.decode_from_latent(z).numpy(), g[f"{p}_decode_from_latent"], atol=2e-4, rtol=0)
            lat = m.map2latent(x)
            assert lat.shape == (2, 40, 256)
            f = m.forward(x)
            assert set(f) == {"poses_feat", "embedding_loss", "perplexity", "rec_pose"} and f["rec_pose"].shape == x.shape
        gcfg = pa.EmageVAEConvConfig(**gc)
        ae = pa.EmageVAEConv(gcfg).set_precision("fp32")
        ae.load_state_dict(synthetic.vae_state(gcfg, 0))
        x = torch.randn(2, 40, 61, generator=gen)
        np.testing.assert_allclose(ae.forward(x)["rec_pose"].numpy(), g["global_rec_pose"], atol=2e-4, rtol=0)


def test_no_cpu_fallback():
    """Outside the test patch the product refuses to run without a device."""
    model, vq = common.product_models(precision="fp32")
    audio, spk, motion, mask = common.window_inputs(1)
    with pytest.raises(RuntimeError, match="MI355X"):
        model.forward(audio, spk, motion, mask)
    with pytest.raises(RuntimeError, match="MI355X"):
        vq.vq_model_upper.decode(torch.zeros(1, 8, dtype=torch.long))


def test_checkpoint_round_trip(tmp_path):
    import pantomatrix_amd as pa
    _, vqc, _ = common.cfg_dicts()
    cfg = pa.EmageVQVAEConvConfig(**vqc["face"])
    m = pa.EmageVQVAEConv(cfg)
    m.load_state_dict(synthetic.vqvae_state(cfg, "face", 3))
    m.save_pretrained(str(tmp_path / "emage_vq" / "face"))
    m2 = pa.EmageVQVAEConv.from_pretrained(str(tmp_path), subfolder="emage_vq/face")
    assert m2.config.vae_test_dim == 106 and list(m2.state_dict()) == list(m.state_dict())
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])
    with pytest.raises(RuntimeError):
        m2.load_state_dict({"bogus": torch.zeros(1)})


def test_infer_codes_equals_inference_path(golden_dir):
    """The lean code path (per-window indices, seed-only decode) produces the reference's codes and motion."""
    g = np.load(os.path.join(golden_dir, "infer_129f_b1.npz"))
    model, vq = common.product_models(precision="fp32")
    audio = synthetic.synthetic_audio(1, synthetic.samples_for_frames(129))
    with fake_ops.installed(), torch.no_grad():
        codes = model.infer_codes(audio, torch.zeros(1, 1, dtype=torch.long), vq)
        pred = vq.decode(**codes, get_global_motion=True, ref_trans=torch.zeros(1, 3))
        model.seed_only_decode = False
        codes_full = model.infer_codes(audio, torch.zeros(1, 1, dtype=torch.long), vq)
    for p in ("face", "upper", "hands", "lower"):
        assert np.array_equal(codes[f"{p}_index"].numpy(), g[f"index_{p}"]), p
        assert torch.equal(codes[f"{p}_index"], codes_full[f"{p}_index"])
    # the latent-routed face comes back as the index of its nearest code (what decode(face_latent=...) computes first)
    assert codes["face_latent"] is None and codes["face_index"].shape == (1, 129)
    np.testing.assert_allclose(pred["motion_axis_angle"].numpy(), g["poses"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(pred["trans"].numpy(), g["trans"], atol=1e-3, rtol=0)


def test_models_are_nn_modules_with_reference_parameter_tree():
    """the accelerated classes are torch.nn.Modules whose parameters / buffers carry the reference's names:
    named_parameters(), buffers, hooks, state-dict loading errors and device moves behave as for the reference classes;
    the packed operand copies are rebuilt after anything that may have changed the parameters."""
    from pantomatrix_amd import spec
    model, vq = common.product_models(precision="f16x3")
    assert isinstance(model, torch.nn.Module) and isinstance(vq, torch.nn.Module)
    names = dict(model.named_parameters())
    bufs = dict(model.named_buffers())
    sp = model._spec
    assert set(names) | set(bufs) == set(sp) and list(model.state_dict()) == list(sp)
    assert "audio_encoder_face.feat_extractor.0.bn1.running_mean" in bufs and "position_embeddings.pe" in bufs
    assert "audio_motion_cross_attn.layers.7.multihead_attn.in_proj_weight" in names and "mask_embedding" in names
    assert all(k.split(".")[0] in ("vq_model_face", "vq_model_upper", "vq_model_hands", "vq_model_lower", "global_motion")
               for k in vq.state_dict())
    assert not model.training and sum(p.numel() for p in model.parameters()) > 100_000_000
    # hooks run (nn.Module.__call__), and the packed copies follow the parameters
    seen = []
    h = model.register_forward_hook(lambda m, a, out: seen.append(sorted(out)))
    audio, spk, motion, mask = common.window_inputs(1)
    with fake_ops.installed(), torch.no_grad():
        out1 = model(audio, spk, motion, mask)
        first_pack = model._packed
        with torch.no_grad():
            model.get_parameter("face_out_proj.bias").add_(1.0)
        model.invalidate_packed()
        out2 = model(audio, spk, motion, mask)
        assert model._packed is not first_pack
    h.remove()
    assert len(seen) == 2 and seen[0] == sorted(orc.OUT_KEYS)
    assert float((out2["rec_face"] - out1["rec_face"] - 1.0).abs().max()) < 1e-5      # the edited bias reached the packed copy
    sd = model.state_dict()
    model.load_state_dict(sd)
    assert model._packed is None                                                     # load_state_dict drops the packed copies
    bad = dict(sd)
    bad.pop("face_out_proj.bias")
    with pytest.raises(RuntimeError, match="Missing key"):
        model.load_state_dict(bad)
    model.train()                                                                    # EmageAudioModel trains through its class API (training.train_forward)
    assert model.training
    model.eval()
    with pytest.raises(NotImplementedError):                                         # the VQ-VAEs are frozen in the reference (T:233-245)
        vq.vq_model_face.train()


def test_packed_operands_follow_in_place_updates():
    """train, `optimizer.step()`, `model.eval(); model(...)` used the packed weights from before the update.
    `_engine()` now compares the version counters of every parameter AND buffer with the stamp taken at packing time: an in-place
    parameter edit (what an optimiser step is), a BatchNorm-buffer update (what a train-mode forward does) and a parent module's
    `load_state_dict` (which never calls the children's) are each followed by a re-pack."""
    model, vq = common.product_models(precision="fp32")
    audio, spk, motion, mask = common.window_inputs(1)
    with fake_ops.installed(), torch.no_grad():
        out0 = model.forward(audio, spk, motion, mask)
        pk0 = model._packed
        assert model._engine() is pk0                                    # nothing changed: no re-pack
        model.face_out_proj.bias.add_(1.0)                               # an optimiser step's in-place update
        out1 = model.forward(audio, spk, motion, mask)
        assert model._packed is not pk0
        assert float((out1["rec_face"] - out0["rec_face"] - 1.0).abs().max()) < 1e-5
        assert float((out1["rec_upper"] - out0["rec_upper"]).abs().max()) == 0.0
        pk1 = model._packed
        model.audio_encoder_body.feat_extractor._modules["0"].bn1.running_mean.add_(0.25)      # a train-mode forward's buffer update
        out2 = model.forward(audio, spk, motion, mask)
        assert model._packed is not pk1 and float((out2["rec_upper"] - out1["rec_upper"]).abs().max()) > 1e-4
        # parent load_state_dict: the children's packed copies follow
        idx = torch.arange(8).view(1, 8) % 256
        d0 = vq.vq_model_upper.decode(idx)
        other = common.product_models(seed=1, precision="fp32")[1]
        vq.load_state_dict(other.state_dict())
        d1 = vq.vq_model_upper.decode(idx)
        assert float((d1 - other.vq_model_upper.decode(idx)).abs().max()) == 0.0 and float((d1 - d0).abs().max()) > 1e-3


def test_operand_scales_are_rederived_when_a_weight_leaves_their_range():
    """The split-fp16 operand scales are power-of-two constants chosen at the first packing and kept for every re-packing (a training
    step re-packs without a host read-back).  A weight that has since grown 64x would overflow its fp16 hi plane with the cached scale:
    every cached-scale packing checks max |w| x scale against [2^10, 2^14) on the device (`ops.f16_scale_out_of_range`), and
    `_engine()` answers a miss by choosing the scales afresh (VERDICT weak #2)."""
    from pantomatrix_amd import ops
    w = torch.randn(16, 64)
    _, s = ops.split_f16_weights_h2(w)
    assert int(ops.f16_scale_out_of_range(w, s)) == 0 and int(ops.f16_scale_out_of_range(w * 3.9, s)) in (0, 1)
    assert int(ops.f16_scale_out_of_range(w * 8, s)) == 1 and int(ops.f16_scale_out_of_range(w / 16, s)) == 1
    assert int(ops.f16_scale_out_of_range(w * float("nan"), s)) == 1 and int(ops.f16_scale_out_of_range(w * 0, s)) == 0
    model, _ = common.product_models(precision="f16x3")
    audio, spk, motion, mask = common.window_inputs(1)
    with fake_ops.installed(), torch.no_grad():
        out0 = model.forward(audio, spk, motion, mask)
        assert model._packed.range_flag is None                           # first packing: every scale fresh
        model.face_cls.fc2.weight.mul_(1.5)
        out1 = model.forward(audio, spk, motion, mask)
        assert model._packed.range_flag is not None and int(model._packed.range_flag) == 0       # re-packed with the cached scales, all in range
        scales1 = dict(model.__dict__["_scale_caches"][(str(model.device), model._packed.dt, False)])
        model.face_cls.fc2.weight.mul_(64.0)
        model.face_cls.fc2.bias.mul_(0.0)
        out2 = model.forward(audio, spk, motion, mask)
        scales2 = model.__dict__["_scale_caches"][(str(model.device), model._packed.dt, False)]
        changed = [k for k in scales1 if scales1[k] != scales2[k]]
        assert len(changed) == 1 and scales2[changed[0]] in (scales1[changed[0]] / 64, scales1[changed[0]] / 128)      # the weight grew 96x
        assert torch.isfinite(out2["cls_face"]).all()
        fresh, _ = common.product_models(precision="f16x3")
        fresh.load_state_dict(model.state_dict())
        out3 = fresh.forward(audio, spk, motion, mask)
        assert float((out3["cls_face"] - out2["cls_face"]).abs().max()) <= 1e-4 * float(out2["cls_face"].abs().max())
        # invalidate_packed(reset_scales=True): the explicit form for weights replaced behind the module's back
        model.invalidate_packed(reset_scales=True)
        assert model.__dict__["_scale_caches"] == {}


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_train_only_operand_set_is_never_handed_to_an_eval_forward(precision):
    """A training step re-packs the MFMA operands behind every update and leaves out what only the eval-mode forward reads (the WavEncoder
    convolutions with their BatchNorms folded in): `_engine(train_only=True)`.  Such a set must never serve an eval-mode caller — in fp32
    precision both forwards ask for the same operand type, so only the flag tells them apart — while a FULL set may serve the training
    forward; the two keep separate scale caches (the cache is keyed by packing order)."""
    model, _ = common.product_models(precision=precision)
    audio, spk, motion, mask = common.window_inputs(1)
    with fake_ops.installed(), torch.no_grad():
        lean = model._engine(h2=False, train_only=True)
        assert lean.train_only and "audio_encoder_face.feat_extractor.1.conv2" not in lean.w and "moton_proj" in lean.w
        assert model._engine(h2=False, train_only=True) is lean                  # nothing changed: the training forward keeps its set
        out = model.forward(audio, spk, motion, mask)                            # eval mode: a full set is packed
        full = model._packed
        assert full is not lean and not full.train_only and "audio_encoder_face.feat_extractor.1.conv2" in full.w
        if precision == "fp32":                                                  # same operand type: the full set also serves the training forward
            assert model._engine(h2=False, train_only=True) is full
        caches = model.__dict__.get("_scale_caches", {})
        if precision == "f16x3":
            assert {k[2] for k in caches} == {False, True}                       # one cache per kind of set
            assert (str(model.device), lean.dt, True) in caches and (str(model.device), full.dt, False) in caches
        assert all(torch.isfinite(v).all() for v in out.values() if v is not None)

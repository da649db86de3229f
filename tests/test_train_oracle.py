"""The TRAINING-step oracle (oracle/emage_train_oracle.py: train-mode forward with BatchNorm batch statistics and
dropout, the six losses, Adam) against the REAL reference step run live in the build container, and against the golden
fixture generated from it (tests/golden/train_step_b2.npz) — groundwork for SURVEY §8(f) row 1, CPU only."""
import os

import numpy as np
import pytest
import torch

import common
from oracle import emage_oracle as orc
from oracle import emage_train_oracle as tro
from oracle import reference_harness as rh
from pantomatrix_amd import synthetic
from pantomatrix_amd.configuration_emage_audio import EmageAudioConfig


from tools.workloads import train_batch  # noqa: E402  (shared with bench.py and tests/golden/make_golden_train.py)


def _oracle_step(iteration, seed):
    acfg, _, _ = common.cfg_dicts()
    cfg = EmageAudioConfig(**acfg)
    _, vq = common.oracle_models()
    sd = synthetic.audio_model_state(cfg, 0)
    return tro.train_step(sd, vq, cfg, train_batch(), iteration, seed=seed), sd


def test_adam_update_matches_torch():
    g = torch.Generator().manual_seed(0)
    p = torch.randn(50, 7, generator=g)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    m, v, cur = torch.zeros_like(p), torch.zeros_like(p), p.clone()
    for step in range(1, 4):
        grad = torch.randn(50, 7, generator=g)
        ref.grad = grad.clone()
        opt.step()
        cur, m, v = tro.adam_update(cur, grad, m, v, step)
        assert torch.allclose(cur, ref.detach(), rtol=0, atol=1e-7)


def test_losses_restate_torch_functionals():
    g = torch.Generator().manual_seed(1)
    cfg = EmageAudioConfig(**common.cfg_dicts()[0])
    pred = {f"{k}_{q}": torch.randn(2, 9, 256, generator=g) for k in ("rec", "cls") for q in ("face", "upper", "hands", "lower")}
    lat = {q: torch.randn(2, 9, 256, generator=g) for q in ("face", "upper", "hands", "lower")}
    idx = {q: torch.randint(0, 256, (2, 9), generator=g) for q in ("face", "upper", "hands", "lower")}
    want_rec = sum(getattr(cfg, "l" + q[0]) * torch.nn.functional.mse_loss(pred[f"rec_{q}"], lat[q]) for q in lat)
    nll = torch.nn.NLLLoss()
    want_cls = sum(getattr(cfg, "c" + q[0]) * nll(torch.log_softmax(pred[f"cls_{q}"], 2).permute(0, 2, 1), idx[q]) for q in idx)
    assert torch.allclose(tro.rec_loss(pred, lat, cfg), want_rec, rtol=1e-6)
    assert torch.allclose(tro.cls_loss(pred, idx, cfg), want_cls, rtol=1e-6)


def test_train_forward_without_dropout_in_eval_stats_limit():
    """p = 0 and batch statistics equal to the running statistics would be the eval forward; at least the
    no-dropout train forward must agree with the eval oracle everywhere the WavEncoder is not involved."""
    cfg = EmageAudioConfig(**common.cfg_dicts()[0])
    sd = synthetic.audio_model_state(cfg, 0)
    audio, spk, motion, mask = common.window_inputs(2)
    with torch.no_grad():
        a = tro.forward_train(sd, audio, spk, motion, mask, use_audio=False, p=0.0)
        b = orc.AudioModel(sd, cfg).forward(audio, spk, motion, mask, use_audio=False)
    for k in ("rec_upper", "rec_hands", "rec_lower", "cls_upper"):      # body path with use_audio=False never sees audio
        assert torch.allclose(a[k], b[k], atol=2e-5), k


@pytest.mark.skipif(not rh.available(), reason="needs /root/reference (build container only)")
@pytest.mark.parametrize("iteration", [0, 3])
def test_train_step_matches_reference_live(iteration):
    acfg, vqc, gc = common.cfg_dicts()
    model, vq = rh.build_reference(acfg, vqc, gc, 0)
    ref_loss, ref_grads, ref_sd = rh.reference_train_step(model, vq, acfg, train_batch(), iteration, seed=11)
    (loss, grads, new_sd, _), sd0 = _oracle_step(iteration, seed=11)
    for k in ("rec_seed", "cls_seed", "rec_audio", "cls_audio", "rec_mask", "cls_mask", "all"):
        assert abs(loss[k] - ref_loss[k]) <= 2e-4 * max(1.0, abs(ref_loss[k])), (k, loss[k], ref_loss[k])
    assert set(grads) == set(ref_grads)
    # a conv bias in front of a train-mode BatchNorm has an exactly-zero gradient (the batch mean removes it): fp32 noise
    # there is judged against the step's largest gradient, everything else against its own tensor's scale
    gmax = max(float(g.abs().max()) for g in ref_grads.values())
    worst = 0.0
    for k, g in ref_grads.items():
        scale = float(g.abs().max())
        err = float((grads[k] - g).abs().max())
        worst = max(worst, err / (scale + 1e-6 * gmax))
        assert err <= 5e-3 * scale + 1e-6 * gmax, (k, err, scale, gmax)
    print("max relative gradient error vs reference:", worst, "largest gradient entry:", gmax)
    # Adam's first step is lr * g / (|g| + eps): where the true gradient is zero (conv biases shadowed by BatchNorm) the
    # update is +-lr on the SIGN of fp32 noise, so those entries can only be bounded by 2 lr; everything else must agree
    shadowed = 0
    for k in grads:
        diff = float((new_sd[k] - ref_sd[k]).abs().max())
        if float(ref_grads[k].abs().max()) < 1e-5 * gmax:
            shadowed += 1
            assert diff <= 2 * 1.5e-4 + 1e-7, (k, diff)
        else:
            noisy = (ref_grads[k].abs() < 1e-6 * gmax)           # single near-zero entries inside a live tensor
            d = (new_sd[k] - ref_sd[k]).abs()
            assert float(d[~noisy].max()) < 3e-5, (k, float(d[~noisy].max()))
            assert float(d.max()) <= 2 * 1.5e-4 + 1e-7, k
    assert shadowed >= 12                              # conv1 / conv2 / downsample biases of the two WavEncoders
    for k in ref_sd:                                  # BatchNorm buffers after three train-mode forwards
        if k.endswith((".running_mean", ".running_var")):
            assert torch.allclose(new_sd[k], ref_sd[k], rtol=1e-4, atol=1e-6), k
        if k.endswith(".num_batches_tracked"):
            assert int(new_sd[k]) == int(ref_sd[k]) == int(sd0[k]) + 3, k
    untouched = [k for k in tro.trainable_keys(sd0) if k not in grads]
    assert untouched and all(k.startswith(("transformer_en_layer.", "audio_motion_cross_attn_layer.")) for k in untouched)


def test_train_step_matches_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    (loss, grads, new_sd, _), _ = _oracle_step(int(g["iteration"]), seed=int(g["seed"]))
    for k in ("rec_seed", "cls_seed", "rec_audio", "cls_audio", "rec_mask", "cls_mask", "all"):
        assert abs(loss[k] - float(g["loss_" + k])) <= 2e-4 * max(1.0, abs(float(g["loss_" + k]))), k
    names = [str(n) for n in g["grad_names"]]
    assert sorted(names) == sorted(grads)
    gmax = max(float(v.abs().max()) for v in grads.values())
    for n, norm, first, shadowed, s in zip(names, g["grad_norms"], g["grad_first"], g["shadowed"], g["param_sum_after"]):
        if shadowed:                                   # true gradient 0 (bias in front of a train-mode BatchNorm): fp32 noise
            assert float(grads[n].abs().max()) < 1e-4 * gmax, n
            continue
        assert abs(float(grads[n].norm()) - float(norm)) <= 5e-3 * float(norm) + 1e-6 * gmax, n
        assert abs(float(grads[n].reshape(-1)[0]) - float(first)) <= 5e-3 * float(grads[n].abs().max()) + 1e-6 * gmax, n
        # sum of the updated tensor: each entry may differ by 3e-5 (see the live test), a few noise entries by 2 lr
        assert abs(float(new_sd[n].double().sum()) - float(s)) <= 3e-5 * new_sd[n].numel() ** 0.5 + 2e-3, n

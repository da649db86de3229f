"""The reference's OWN `train_val_fn` (train_emage_audio.py:130-204, compiled from its source where it lies: oracle/reference_harness.py)
run against the PRODUCT classes — what swapping `from models.emage_audio import ...` for `from pantomatrix_amd import ...` in the
training script amounts to ("do this" 5).  Build container only (/root/reference does not travel); the kernels are
the CPU stand-ins of tests/fake_ops.py, the GPU twin of the same loop is tests/test_train_forward_gpu.py::
test_reference_style_training_loop_on_the_device."""
import os
import types

import numpy as np
import pytest
import torch

import common
import fake_ops
import train_common as tc
from oracle import emage_train_oracle as tro
from oracle import reference_harness as rh
from pantomatrix_amd import synthetic
from pantomatrix_amd.configuration_emage_audio import EmageAudioConfig

pytestmark = pytest.mark.skipif(not rh.available(), reason="needs /root/reference (build container only)")


class _TorchWithRecordedRand:
    """`torch` as train_val_fn sees it, except that `torch.rand(bs, t, 337)` (the motion mask draw, T:163) returns values that turn
    `rand < mask_ratio` into the RECORDED mask of the reference step for any ratio in (0, 1): the product consumes no global
    generator state in its forwards (its dropout masks are injected), so the global stream is not where the reference's was."""

    def __init__(self, recorded_mask):
        self._mask = recorded_mask

    def rand(self, *shape, **kw):
        assert tuple(shape) == tuple(self._mask.shape), (shape, self._mask.shape)
        return torch.where(self._mask > 0.5, torch.full_like(self._mask, -1.0), torch.full_like(self._mask, 2.0))

    def __getattr__(self, name):
        return getattr(torch, name)


def _reference_cfg(acfg):
    return types.SimpleNamespace(model=types.SimpleNamespace(**acfg), solver=types.SimpleNamespace(max_grad_norm=0.99))


def test_reference_train_val_fn_trains_the_product_model(golden_dir):
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    seed, it = int(g["seed"]), int(g["iteration"])
    batch, ref_losses, masks, random_mask, _ = tc.oracle_step(seed, it)
    acfg = common.cfg_dicts()[0]
    cfg = EmageAudioConfig(**acfg)
    _, ovq = common.oracle_models()
    _, _, new_sd, _ = tro.train_step(synthetic.audio_model_state(cfg, 0), ovq, cfg, batch, it, seed=seed)

    fns = rh.reference_train_functions()
    fns["torch"] = _TorchWithRecordedRand(random_mask)
    model, vq = common.product_models(precision="fp32")
    opt = torch.optim.Adam(filter(lambda q: q.requires_grad, model.parameters()), lr=1.5e-4, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-8)
    sched = types.SimpleNamespace(step=lambda: None)
    model.dropout_masks_override = [list(m) for m in masks]
    with fake_ops.installed():
        losses = fns["train_val_fn"](_reference_cfg(acfg), batch, model, torch.device("cpu"), mode="train", motion_vq=vq, optimizer=opt,
                                     lr_scheduler=sched, ClsFn=torch.nn.NLLLoss(), iteration=it)
    assert model.training and not model.dropout_masks_override          # train_val_fn called model.train(); three forwards consumed the masks
    model.eval()
    for k, v in ref_losses.items():
        assert abs(float(losses[k]) - v) < 2e-4 * max(1.0, abs(v)), (k, float(losses[k]), v)
    for k in ("rec_seed", "cls_seed", "rec_audio", "cls_audio", "rec_mask", "cls_mask", "all"):      # and the REAL reference's numbers
        assert abs(float(losses[k]) - float(g["loss_" + k])) < 2e-4 * max(1.0, abs(float(g["loss_" + k]))), k
    params = model._flat_params()
    lr = 1.5e-4
    checked = 0
    for name, shadowed, s in zip([str(n) for n in g["grad_names"]], g["shadowed"], g["param_sum_after"]):
        if shadowed:
            continue
        p = params[name]
        assert float((p - new_sd[name]).abs().max()) <= 2.05 * lr, name
        assert abs(float(p.double().sum()) - float(s)) <= 3e-5 * p.numel() ** 0.5 + 2e-3 + (0.3 * lr * p.numel() if name.startswith("audio_encoder") else 0), name
        checked += 1
    assert checked > 400


def test_reference_train_val_fn_validates_with_the_product_models():
    """mode="val" of the same function: eval forwards + `motion_vq.decode(...)` of the product classes hand the FGD evaluator the same
    rot-6D motion as the reference's own classes do."""
    acfg, vqc, gc = common.cfg_dicts()
    from test_train_oracle import train_batch
    batch = train_batch(bs=2)
    random_mask = (torch.rand(2, batch["motion"].shape[1], acfg["pose_dims"] + 7, generator=torch.Generator().manual_seed(5)) < 0.4).float()

    def run(model, vq, ctx):
        seen = {}
        fns = rh.reference_train_functions()
        fns["torch"] = _TorchWithRecordedRand(random_mask)
        evaluator = types.SimpleNamespace(update=lambda pred, gt: seen.update(pred=pred.detach().clone(), gt=gt.detach().clone()))
        with ctx, torch.no_grad():
            losses = fns["train_val_fn"](_reference_cfg(acfg), batch, model, torch.device("cpu"), mode="val", motion_vq=vq,
                                         ClsFn=torch.nn.NLLLoss(), iteration=2, fgd_evaluator=evaluator)
        return {k: float(v) for k, v in losses.items()}, seen

    import contextlib
    ref_model, ref_vq = rh.build_reference(acfg, vqc, gc, 0)
    ref_losses, ref_seen = run(ref_model, ref_vq, contextlib.nullcontext())
    model, vq = common.product_models(precision="fp32")
    got_losses, got_seen = run(model, vq, fake_ops.installed())
    assert not model.training
    for k, v in ref_losses.items():
        assert abs(got_losses[k] - v) < 2e-4 * max(1.0, abs(v)), (k, got_losses[k], v)
    assert got_seen["pred"].shape == ref_seen["pred"].shape
    assert float((got_seen["gt"] - ref_seen["gt"]).abs().max()) < 1e-6
    assert float((got_seen["pred"] - ref_seen["pred"]).abs().max()) < 1e-3

"""Run the REAL reference implementation (imported read-only from /root/reference) on the
synthetic weights — TEST INFRASTRUCTURE.  Used to pin `oracle/emage_oracle.py` and to
generate `tests/golden/*.npz`.  /root/reference exists only in the build container; every
caller must check `available()` first (the GPU box has no reference).
"""
from __future__ import annotations

import os
import sys
import warnings

REFERENCE_ROOT = "/root/reference"
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_stubs")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models", "emage_audio"))


def import_reference():
    """Import `models.emage_audio` from the reference.  `omegaconf` is not installed, so a
    stub exposing `OmegaConf.to_container` is put on the path."""
    try:
        import omegaconf  # noqa: F401
    except ImportError:
        if _STUBS not in sys.path:
            sys.path.insert(0, _STUBS)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(1, REFERENCE_ROOT)
    import models.emage_audio as ref  # namespace package: the reference has no models/__init__.py
    return ref


def build_reference(audio_cfg: dict, vq_cfgs: dict, global_cfg: dict, seed: int = 0):
    """Instantiate the reference modules and load the synthetic weights into them with
    `load_state_dict(strict=True)` (from_pretrained is broken under transformers 5.x).
    Returns (EmageAudioModel, EmageVQModel), both in eval mode."""
    from pantomatrix_amd import synthetic
    from pantomatrix_amd.configuration_emage_audio import EmageAudioConfig, EmageVQVAEConvConfig, EmageVAEConvConfig
    ref = import_reference()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = ref.EmageAudioModel(ref.EmageAudioConfig(**audio_cfg))
    model.load_state_dict(synthetic.audio_model_state(EmageAudioConfig(**audio_cfg), seed), strict=True)
    parts = {}
    for part in ("face", "upper", "hands", "lower"):
        m = ref.EmageVQVAEConv(ref.EmageVQVAEConvConfig(**vq_cfgs[part]))
        m.load_state_dict(synthetic.vqvae_state(EmageVQVAEConvConfig(**vq_cfgs[part]), part, seed), strict=True)
        parts[part] = m
    g = ref.EmageVAEConv(ref.EmageVAEConvConfig(**global_cfg))
    g.load_state_dict(synthetic.vae_state(EmageVAEConvConfig(**global_cfg), seed), strict=True)
    vq = ref.EmageVQModel(face_model=parts["face"], upper_model=parts["upper"], hands_model=parts["hands"],
                          lower_model=parts["lower"], global_model=g)
    return model.eval(), vq.eval()


def reference_train_functions():
    """The reference's own `get_rec_loss`, `get_cls_loss` and `train_val_fn` (train_emage_audio.py:106-204), compiled
    from its source WITHOUT importing the module (its top level needs wandb, diffusers, librosa, the renderer ...):
    only those three function definitions are executed, in a namespace holding torch, F and the reference's
    rotation_conversions.  Nothing is copied into this repo; the file is read where it lies."""
    import ast
    import importlib
    import torch
    import torch.nn.functional as F
    import_reference()
    rc = importlib.import_module("emage_utils.rotation_conversions")
    path = os.path.join(REFERENCE_ROOT, "train_emage_audio.py")
    tree = ast.parse(open(path).read())
    wanted = {"get_rec_loss", "get_cls_loss", "train_val_fn"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert {n.name for n in body} == wanted
    ns = {"torch": torch, "F": F, "rc": rc}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def reference_train_step(model, vq, model_cfg: dict, batch: dict, iteration: int, seed: int, lr=1.5e-4):
    """Run ONE optimisation step of the reference (its train_val_fn, its model in train mode, torch.optim.Adam with the
    reference's hyper-parameters, configs/emage_audio.yaml:63-78) on CPU.  Returns (losses, grads, state dict after)."""
    import types
    import torch
    fns = reference_train_functions()
    cfg = types.SimpleNamespace(model=types.SimpleNamespace(**model_cfg), solver=types.SimpleNamespace(max_grad_norm=0.99))
    for prm in model.parameters():
        prm.requires_grad = True
    for prm in vq.parameters() if hasattr(vq, "parameters") else []:
        prm.requires_grad = False
    opt = torch.optim.Adam(filter(lambda q: q.requires_grad, model.parameters()), lr=lr, betas=(0.9, 0.999),
                           weight_decay=0.0, eps=1e-8)
    sched = types.SimpleNamespace(step=lambda: None)                     # lr_scheduler 'constant', warm-up 0
    torch.manual_seed(seed)
    # snapshot gradients right after backward(): optimizer.step() does not clear them, so they are still there after
    losses = fns["train_val_fn"](cfg, batch, model, torch.device("cpu"), mode="train", motion_vq=vq, optimizer=opt,
                                 lr_scheduler=sched, ClsFn=torch.nn.NLLLoss(), iteration=iteration)
    grads = {k: prm.grad.detach().clone() for k, prm in model.named_parameters() if prm.grad is not None}
    return {k: float(v.detach()) for k, v in losses.items()}, grads, {k: v.detach().clone() for k, v in model.state_dict().items()}


def build_reference_lstm_model(kind: str, cfg: dict, state_dict):
    """Instantiate the reference's DiscoAudioModel / CamnAudioModel (kind "disco" / "camn") with the given weights."""
    import importlib
    import_reference()
    mod = importlib.import_module(f"models.{kind}_audio")
    cls_cfg = getattr(mod, "DiscoAudioConfig" if kind == "disco" else "CamnAudioConfig")
    cls = getattr(mod, "DiscoAudioModel" if kind == "disco" else "CamnAudioModel")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = cls(cls_cfg(**cfg))
    model.load_state_dict(state_dict, strict=True)
    return model.eval()

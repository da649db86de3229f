"""Shared pieces of the train-mode forward tests: the oracle run with its dropout masks RECORDED (the oracle issues the
reference's generator draws; a recorded mask is exactly the tensor torch multiplies with), product-side drivers."""
import contextlib

import torch
import torch.nn.functional as F

import common
from oracle import emage_train_oracle as tro
from pantomatrix_amd import synthetic
from pantomatrix_amd.configuration_emage_audio import EmageAudioConfig


@contextlib.contextmanager
def recorded_masks(store):
    """Inside: every dropout of the training oracle appends its mask (values 0 or 1 / (1 - p), the logical shape the
    reference's module sees) to `store`; the oracle's result is unchanged bit for bit (x * mask is what F.dropout returns)."""
    saved = tro._drop

    def _drop(x, p):
        if p <= 0:
            return x
        m = F.dropout(torch.ones_like(x), p, training=True)           # ones_like keeps x's memory format: the same draws in the same order
        store.append(m)
        return x * m

    tro._drop = _drop
    try:
        yield store
    finally:
        tro._drop = saved


def oracle_forward(seed, use_audio=True, new_stats=None, batch=2):
    cfg = EmageAudioConfig(**common.cfg_dicts()[0])
    sd = synthetic.audio_model_state(cfg, 0)
    audio, spk, motion, mask = common.window_inputs(batch)
    masks = []
    torch.manual_seed(seed)
    with torch.no_grad(), recorded_masks(masks):
        ns = {} if new_stats is None else new_stats
        out = tro.forward_train(sd, audio, spk, motion, mask, use_audio=use_audio, new_stats=ns)
    return (audio, spk, motion, mask), out, masks, ns


_STEP_CACHE = {}


def oracle_step(seed, iteration, bs=2):
    """Cached per (seed, iteration, bs): several tests replay the same reference step; callers must not modify the tensors in place
    (they move copies to the device / slice them)."""
    key = (seed, iteration, bs)
    if key not in _STEP_CACHE:
        _STEP_CACHE[key] = _oracle_step(seed, iteration, bs)
    batch, losses, masks, random_mask, stats = _STEP_CACHE[key]
    return dict(batch), dict(losses), [list(m) for m in masks], random_mask, dict(stats)


def _oracle_step(seed, iteration, bs=2):
    """tro.train_step_losses with every forward's dropout masks and the random motion mask recorded:
    -> (batch, loss dict of floats, [masks of forward 1, 2, 3], random_mask, BatchNorm buffers)."""
    from test_train_oracle import train_batch
    cfg = EmageAudioConfig(**common.cfg_dicts()[0])
    _, vq = common.oracle_models()
    sd = synthetic.audio_model_state(cfg, 0)
    per_forward, motion_masks = [], []
    orig = tro.forward_train

    def spy(sd_, audio, spk, motion, mask, use_audio=True, p=tro.DROPOUT_P, new_stats=None):
        masks = []
        with recorded_masks(masks):
            out = orig(sd_, audio, spk, motion, mask, use_audio=use_audio, p=p, new_stats=new_stats)
        per_forward.append(masks)
        motion_masks.append(mask.clone())
        return out

    tro.forward_train = spy
    try:
        torch.manual_seed(seed)
        with torch.no_grad():
            losses, stats = tro.train_step_losses(sd, vq, cfg, train_batch(bs=bs), iteration)
    finally:
        tro.forward_train = orig
    return train_batch(bs=bs), {k: float(v) for k, v in losses.items()}, per_forward, motion_masks[1], stats


def shard_masks(masks, lo, hi, batch):
    """The slice [lo, hi) of the clips out of a forward's dropout masks: the batch axis is 0 for (B, T, d) / (B, H, Tq, Tk), 1 for (T, B, C)."""
    out = []
    for m in masks:
        if m.shape[0] == batch and m.dim() in (3, 4) and not (m.dim() == 3 and m.shape[1] == batch and m.shape[0] != batch):
            out.append(m[lo:hi].contiguous())
        else:
            out.append(m[:, lo:hi].contiguous())
    return out

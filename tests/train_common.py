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


_FORWARD_CACHE = {}


def oracle_forward(seed, use_audio=True, new_stats=None, batch=2):
    """One train-mode forward of the oracle with its dropout masks recorded.  Calls that start from fresh BatchNorm buffers are cached
    per (seed, use_audio, batch) — the two precisions of a parametrised test replay the same forward; callers get fresh containers and
    must not modify the tensors in place."""
    if new_stats is None:
        key = (seed, bool(use_audio), batch)
        if key not in _FORWARD_CACHE:
            _FORWARD_CACHE[key] = _oracle_forward(seed, use_audio, None, batch)
        inputs, out, masks, ns = _FORWARD_CACHE[key]
        return inputs, dict(out), list(masks), dict(ns)
    return _oracle_forward(seed, use_audio, new_stats, batch)


def _oracle_forward(seed, use_audio, new_stats, batch):
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
    """The oracle's `train_step_losses` with every forward's dropout masks and the random motion mask recorded (tools/workloads.py,
    shared with bench.py's training leg): -> (batch, loss dict of floats, [masks of forward 1, 2, 3], random_mask, BatchNorm buffers)."""
    from tools import workloads
    r = workloads.oracle_train_step_recorded(seed, iteration, bs=bs)
    return r["batch"], r["losses"], r["masks"], r["random_mask"], r["stats"]


def shard_masks(masks, lo, hi, batch):
    """The slice [lo, hi) of the clips out of a forward's dropout masks: the batch axis is 0 for (B, T, d) / (B, H, Tq, Tk), 1 for (T, B, C)."""
    out = []
    for m in masks:
        if m.shape[0] == batch and m.dim() in (3, 4) and not (m.dim() == 3 and m.shape[1] == batch and m.shape[0] != batch):
            out.append(m[lo:hi].contiguous())
        else:
            out.append(m[:, lo:hi].contiguous())
    return out


def wav_encoder_backward_check(model, dev, enc="audio_encoder_body", batch=2):
    """The train-mode WavEncoder of `model` (fp32 precision) forward + backward on `dev` against float64 autograd of the oracle's
    encoder (P:263-314 with batch-statistics BatchNorm).  A LeakyReLU pre-activation within fp32 rounding of 0 takes the other
    slope in two fp32 implementations that round differently, which changes every gradient upstream of it by a visible amount; so the
    comparison walks back from the output: as long as NO activation of the blocks seen so far changed sign against the float64 run,
    the gradient arriving at the block output AND the block's parameter gradients must agree to fp32 accuracy; behind the first flip
    only the norms are compared.  Returns counts for the caller's assertions."""
    import torch.nn.functional as Fn
    from oracle import emage_oracle as orc
    from pantomatrix_amd import training
    from pantomatrix_amd.modeling_emage_audio import _Ctx
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in synthetic.audio_model_state(EmageAudioConfig(**common.cfg_dicts()[0]), 0).items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if k.startswith(enc + ".") and v.is_floating_point() and "running_" not in k}
    audio = common.window_inputs(batch)[0]
    h, outs, ns = audio.double().unsqueeze(1), [], {}
    for i, (stride, pad, has_ds) in enumerate(orc.WAV_BLOCKS):
        b = f"{enc}.feat_extractor.{i}"
        y = Fn.conv1d(h, sd[b + ".conv1.weight"], sd[b + ".conv1.bias"], stride=stride, padding=pad)
        y = Fn.leaky_relu(tro._bn_train(sd, b + ".bn1", y, ns), 0.01)
        y = tro._bn_train(sd, b + ".bn2", Fn.conv1d(y, sd[b + ".conv2.weight"], sd[b + ".conv2.bias"], stride=1, padding=7), ns)
        if has_ds:
            h = tro._bn_train(sd, b + ".downsample.1", Fn.conv1d(h, sd[b + ".downsample.0.weight"], sd[b + ".downsample.0.bias"], stride=stride, padding=pad), ns)
        h = Fn.leaky_relu(y + h, 0.01)
        h.retain_grad()
        outs.append(h)
    up = torch.randn(h.transpose(1, 2).shape, generator=torch.Generator().manual_seed(1))
    (h.transpose(1, 2) * up.double()).sum().backward()

    fwd = training.TrainForward(model)
    seen, flips = {}, {}
    orig = fwd._wav_blocks_backward

    def spy(cx, svs):
        for sv in svs:
            seen[sv["i"]] = fwd.tape.get(sv["out"]).clone()
            ref_out = outs[sv["i"]].detach().permute(0, 2, 1).reshape(sv["out"].shape).to(sv["out"].device)
            flips[sv["i"]] = int(((sv["out"] > 0) != (ref_out > 0)).sum())
        return orig(cx, svs)

    fwd._wav_blocks_backward = spy
    with torch.no_grad():
        cx = _Ctx(model._engine(h2=False))
        fwd._train_pack(cx.pk)
        fwd.tape, fwd.param_grads = training._Tape(cx.dev), {}
        x, _ = fwd._wav_encoder(cx, enc, 1, audio.to(dev), batch, {})
        fwd.tape.add(x, up.reshape(x.shape).to(dev))
        fwd.tape.run()
    grads = {k: v.detach().cpu() for k, v in fwd.param_grads.items()}
    n_blocks = len(orc.WAV_BLOCKS)
    gmax = max(float(v.grad.abs().max()) for v in leaves.values())
    clean, clean_blocks, params_checked, worst, bad = True, 0, 0, 0.0, []
    for i in reversed(range(n_blocks)):               # walking back from the output
        ref = outs[i].grad.permute(0, 2, 1).reshape(seen[i].shape)
        got = seen[i].double().cpu()
        rel = float((got - ref).abs().max() / ref.abs().max())
        if clean:
            assert rel < 2e-5, (i, rel)
        else:
            assert float((got - ref).norm() / ref.norm()) < 5e-2, i
        clean = clean and flips[i] == 0
        if clean:                                     # no flip in this block or behind it: its parameter gradients are pinned, too
            clean_blocks += 1
            for k, leaf in leaves.items():
                if not k.startswith(f"{enc}.feat_extractor.{i}."):
                    continue
                g = grads[k].double().reshape(leaf.grad.shape)
                # a conv bias in front of a train-mode BatchNorm has an exactly-zero gradient: fp32 noise there is judged against the
                # largest gradient of the encoder
                err = float((g - leaf.grad).abs().max())
                scale = float(leaf.grad.abs().max())
                shadowed = scale < 1e-5 * gmax        # true gradient zero: what is left is the rounding noise of a sum over all positions
                if err > 2e-4 * scale + (2e-5 if shadowed else 5e-6) * gmax:
                    bad.append((k, err, scale))
                worst = max(worst, err / (scale + 1e-2 * gmax))
                params_checked += 1
    assert not bad, (bad, gmax)
    assert sum(flips.values()) <= 8, flips
    return dict(flips=dict(flips), clean_blocks=clean_blocks, params_checked=params_checked, worst_param_rel=worst)

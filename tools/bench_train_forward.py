#!/usr/bin/env python
"""Forward side of the EMAGE training step at BASELINE config 3 (per-GPU batch 56 x 64-frame clips) on the MI355X:
targets through the HIP VQ models, the three train-mode forwards (batch-statistics BatchNorm, dropout with device-drawn
masks) and the six losses (pantomatrix_amd/training.py).  Eager launches, one stream.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402


def draw_masks(b, t, d, ff, h, ta, dev, p=0.1):
    from pantomatrix_amd import spec
    mk = lambda *shape: (torch.rand(*shape, device=dev) >= p).float() / (1 - p)
    dec = lambda tk: [mk(b, h, t, t), mk(t, b, d), mk(b, h, t, tk), mk(t, b, d), mk(t, b, ff), mk(t, b, d)]
    enc = [mk(b, h, t, t), mk(t, b, d), mk(t, b, ff), mk(t, b, d)]
    out = [mk(b, t, d)]
    for _ in range(spec.N_FACE_LAYERS):
        out += dec(t)
    out += [mk(b, t, d)] + enc + [mk(b, t, d)]
    for _ in range(spec.N_CROSS_LAYERS):
        out += dec(ta)
    for _ in range(3):
        out += dec(t)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=56)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--graph", action="store_true", help="with --full-step: the step captured as one hipGraph (Trainer.capture / replay; precision fp32)")
    ap.add_argument("--full-step", action="store_true", help="whole optimisation steps (training.Trainer.step: + backward, Adam) instead of the forward side")
    args = ap.parse_args()
    import common
    from pantomatrix_amd import training
    dev = "cuda"
    model, vq = common.product_models(precision=args.precision, device=dev)
    fwd = training.TrainForward(model)
    b, t = args.batch, 64
    g = torch.Generator().manual_seed(5)
    batch = dict(motion=0.3 * torch.randn(b, t, 165, generator=g), audio=0.1 * torch.randn(b, t * 16000 // 30, generator=g),
                 expressions=0.5 * torch.randn(b, t, 100, generator=g), trans=0.1 * torch.randn(b, t, 3, generator=g),
                 foot_contact=(torch.rand(b, t, 4, generator=g) > 0.5).float())
    batch = {k: v.to(dev) for k, v in batch.items()}
    c = model.config
    ta = model._wav_lengths(batch["audio"].shape[1])[-1]
    random_mask = (torch.rand(b, t, 337, device=dev) < 0.5).float()

    trainer = training.Trainer(model, vq) if args.full_step else None
    if args.graph:
        trainer.capture(batch, random_mask)                          # dropout masks are drawn inside the graph (ops.dropout_mask)

        def step():
            return trainer.replay(), None
    else:
        step = None

    def eager_step():
        masks = [draw_masks(b, t, c.hidden_size, 2 * c.hidden_size, 4, ta, dev) for _ in range(3)]
        if trainer is not None:
            return trainer.step(batch, 0, masks, random_mask), None
        return training.step_losses(fwd, vq, batch, 0, masks, random_mask)

    step = step or eager_step
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses, _ = step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    what = ("one whole EMAGE optimisation step (targets, 3 x (train-mode forward, losses, backward), Adam), "
            + ("ONE hipGraph replay (+ drawing the dropout masks)" if args.graph else "eager, one stream") + "; f16x3: split-fp16 MFMA "
            "contractions forward and backward, fp32: exact-fp32 MFMA" if args.full_step else
            "forward side of one EMAGE training step (targets + 3 train-mode forwards + 6 losses), eager, one stream")
    print(json.dumps({"what": what, "config": {"workload": "BASELINE config 3", "clips_per_gpu": b, "frames_per_clip": t}, "dtype": args.precision,
                      "ms_per_step": ms, "clip_windows_per_s": b / (ms * 1e-3), "peak_memory_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
                      "losses": {k: round(v, 4) for k, v in losses.items()}}))


if __name__ == "__main__":
    main()

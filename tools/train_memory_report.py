#!/usr/bin/env python
"""Where the memory of one EMAGE training step (56 clips per GPU, f16x3, eager `Trainer.step`) is: live bytes before / after and the peak
inside every phase — targets, each forward, each backward, the shared encoder backward, Adam — plus what is resident between steps.
    python tools/train_memory_report.py [--batch 56] [--no-share]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pantomatrix_amd import ops, training  # noqa: E402
from tools import workloads  # noqa: E402

GB = 2.0 ** 30


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=56)
    ap.add_argument("--no-share", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    rows = []

    def mark(name):
        torch.cuda.synchronize()
        rows.append({"at": name, "live_gb": round(torch.cuda.memory_allocated() / GB, 3), "peak_since_last_gb": round(torch.cuda.max_memory_allocated() / GB, 3)})
        torch.cuda.reset_peak_memory_stats()

    def phase(obj, attr, name):
        fn = getattr(obj, attr)
        count = {"n": 0}

        def inner(*a, **k):
            mark(f"before {name} #{count['n']}")
            r = fn(*a, **k)
            mark(f"after  {name} #{count['n']}")
            count["n"] += 1
            return r
        setattr(obj, attr, inner)

    mark("process start")
    model, vq = workloads.product_models(precision="f16x3", device=dev)
    mark("models built")
    data = {k: v.to(dev) for k, v in workloads.train_batch(bs=args.batch, t=64).items()}
    random_mask = (torch.rand(args.batch, 64, 337, generator=torch.Generator().manual_seed(6)) < 0.5).float().to(dev)
    trainer = training.Trainer(model, vq, seed=1, share_encoders=not args.no_share)
    mark("trainer built (buckets)")
    trainer.step(data, random_mask=random_mask)          # first step: packs, scales, Adam state
    mark("after step 1 (resident between steps)")
    phase(training, "targets", "targets")
    fwd_cls = training.TrainForward
    phase(fwd_cls, "__call__", "forward")
    phase(fwd_cls, "backward", "backward")
    phase(fwd_cls, "finish_shared", "encoder backward (shared)")
    real_adam = ops.adam_multi

    def adam(*a, **k):
        mark("before adam")
        return real_adam(*a, **k)
    ops.adam_multi = adam
    trainer.step(data, random_mask=random_mask)
    mark("after step 2")
    print(json.dumps({"batch": args.batch, "share_encoders": not args.no_share, "marks": rows}, indent=1))


if __name__ == "__main__":
    main()

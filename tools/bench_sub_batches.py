#!/usr/bin/env python
"""ClipRunner(sub_batches = k) at the BASELINE batch (64 x 128-frame clips, f16x3): k independent groups of clips, each its own captured graph, replayed on k
streams at once — does a second chain fill the ramp / drain gaps of the first?  Device-side timing (no D2H), interleaved arms.  One JSON line.
    python tools/bench_sub_batches.py [--steps 20]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
steps = 20 if "--steps" not in sys.argv else int(sys.argv[sys.argv.index("--steps") + 1])
from pantomatrix_amd import synthetic  # noqa: E402
from pantomatrix_amd.runtime import ClipRunner  # noqa: E402
from tools import workloads as common  # noqa: E402

dev = torch.device("cuda", 0)
model, vq = common.product_models(precision="f16x3", device=dev)
n = synthetic.samples_for_frames(128)
audio = synthetic.synthetic_audio(64, n).to(dev)
runners = {k: ClipRunner(model, vq, 64, n, use_graph=True, sub_batches=k) for k in (1, 2, 4)}


def run(r):
    if r.sub == 1:
        r.run_device()
        return
    main = torch.cuda.current_stream(dev)
    start = torch.cuda.Event()
    start.record(main)
    for child, s in zip(r.children, r.streams):
        s.wait_event(start)
        with torch.cuda.stream(s):
            child.run_device()
        main.wait_stream(s)


for k, r in runners.items():
    if r.sub == 1:
        r.run_device(audio)
    else:
        m = 64 // k
        for i, c in enumerate(r.children):
            c.run_device(audio[i * m:(i + 1) * m])
torch.cuda.synchronize()
out = {"workload": "64 x 128-frame clips f16x3, device-side step (no D2H)", "steps": steps, "ms_per_step": {}}
for rep in range(3):
    for k, r in runners.items():
        run(r)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run(r)
        torch.cuda.synchronize()
        out["ms_per_step"].setdefault(str(k), []).append(round(1e3 * (time.perf_counter() - t0) / steps, 3))
print(json.dumps(out))

#!/usr/bin/env python
"""Probe (MI355X): which precision / launch combination of ClipRunner reports non-finite values for the checkpoint of
tests/test_parity_gpu.py::test_clip_runner_raises_on_overflow_of_the_split_fp16_range (moton_proj.bias + 6000)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
from pantomatrix_amd import synthetic  # noqa: E402
from pantomatrix_amd.runtime import ClipRunner  # noqa: E402

DEV = "cuda"
n = synthetic.samples_for_frames(128)
a = synthetic.synthetic_audio(2, n).to(DEV)
model, vq = common.product_models(precision="f16x3", device=DEV)
sd = model.state_dict()
sd["moton_proj.bias"] = sd["moton_proj.bias"] + 6000.0
model.load_state_dict(sd)


def probe(tag, mp, vp, graph):
    model.set_precision(mp)
    vq.set_precision(vp)
    r = ClipRunner(model, vq, 2, n, use_graph=graph)
    out = r.run_device(a)
    torch.cuda.synchronize()
    print(tag, "model", mp, "vq", vp, "graph", graph, "-> counter", int(r.nonfinite), "non-finite per output",
          [int((~torch.isfinite(t)).sum()) for t in out], flush=True)


probe("A", "f16x3", "f16x3", True)
probe("B", "fp32", "f16x3", False)
probe("C", "fp32", "fp32", False)
probe("D", "fp32", "fp32", True)
probe("E", "f16x3", "f16x3", True)
probe("F", "fp32", "fp32", True)
model2, vq2 = common.product_models(precision="fp32", device=DEV)
model2.load_state_dict(sd)
r = ClipRunner(model2, vq2, 2, n, use_graph=True)
out = r.run_device(a)
torch.cuda.synchronize()
print("G fresh fp32 models, graph -> counter", int(r.nonfinite), [int((~torch.isfinite(t)).sum()) for t in out])

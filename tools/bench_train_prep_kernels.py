#!/usr/bin/env python
"""Per-launch time of the training step's operand-staging kernels at the 56-clip shapes (run on the MI355X): the transposed EMAGE_H2 cast of a layer
input and `grad_prep` (activation backward + both gradient images + bias partials), as captured graphs of 20 launches."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pantomatrix_amd import ops  # noqa: E402

dev, iters = "cuda", 20
g = torch.Generator().manual_seed(0)


def timed(call):
    call()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(iters):
            call()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


for m, c in ((3584, 768), (3584, 1536), (3584, 2304), (3584, 256)):
    x = torch.randn(m, c, generator=g).to(dev)
    y = torch.randn(m, c, generator=g).to(dev)
    ms = ops.round_up(m, 64)
    print(f"h2_cast(transpose) {m} x {c}: {timed(lambda: ops.h2_cast(x, ms, scale=1024.0, transpose=True)):7.2f} us   "
          f"grad_prep (+ relu) {timed(lambda: ops.grad_prep(x, y, 0.0, 1024.0, n_store=ops.round_up(c, 64), m_store=ms)):7.2f} us   "
          f"grad_prep (linear) {timed(lambda: ops.grad_prep(x, None, 0.0, 1024.0, n_store=ops.round_up(c, 64), m_store=ms)):7.2f} us", flush=True)

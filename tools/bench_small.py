#!/usr/bin/env python
"""Graph-timed microbenchmarks of the non-GEMM kernels at the EMAGE window shapes (run on the MI355X)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pantomatrix_amd import ops  # noqa: E402
from pantomatrix_amd._lib import BF16  # noqa: E402

dev = "cuda"


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


b, t, d, h = 64, 64, 768, 4
m = b * t
bf = torch.bfloat16
qk = torch.randn(m, 2 * d, device=dev).to(bf)
vt = torch.randn(b, d, t, device=dev).to(bf)
out = torch.zeros(m, d, dtype=bf, device=dev)
print("attention 64x64x(4x192)  us:", timeit(lambda: ops.attention(BF16, qk[:, :d], qk[:, d:], vt, d, out, b, h, t, t, d // h)))
x = torch.randn(m, d, device=dev).to(bf)
g_, b_ = torch.ones(d, device=dev), torch.zeros(d, device=dev)
y = torch.zeros(m, d, dtype=bf, device=dev)
print("layernorm 4096x768       us:", timeit(lambda: ops.layernorm(BF16, x, g_, b_, 1e-5, None, None, y)))
print("add 4096x768             us:", timeit(lambda: ops.add(BF16, x, x, None, None, y)))
wav = torch.randn(128, 34112, device=dev) * 0.1
w, bias, slope = torch.randn(256, 15, device=dev), torch.zeros(256, device=dev), torch.ones(256, device=dev)
y0 = torch.zeros(128 * 7460, 256, dtype=bf, device=dev)
print("wav_conv_in 128 clips    us:", timeit(lambda: ops.wav_conv_in(BF16, wav, w, bias, slope, y0, 7460, 5, 1600), iters=5))
idx = torch.randint(0, 256, (m,), device=dev)
tab = torch.randn(256, 256, device=dev)
print("gather_rows 4096x256     us:", timeit(lambda: ops.gather_rows(tab, idx, BF16, 256)))
lg = torch.randn(m, 256, device=dev)
print("argmax_logsoftmax        us:", timeit(lambda: ops.argmax_logsoftmax(lg)))
z = torch.randn(m, 256, device=dev)
print("vq_argmin 4096           us:", timeit(lambda: ops.vq_argmin(z, tab)))
print("null-ish (cast_pad 64x64) us:", timeit(lambda: ops.cast_pad(BF16, z[:64, :64], 64)))

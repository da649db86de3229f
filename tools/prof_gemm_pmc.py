#!/usr/bin/env python
"""A fixed set of emage_gemm / emage_conv_slab launches for rocprofv3 --pmc passes (run on the MI355X):
    rocprofv3 --kernel-trace --pmc <counters> -d out -o pmc --output-format csv -- python tools/prof_gemm_pmc.py [f16x3|h2|bf16]
Each shape is launched 4 times back to back with its production tile configuration (h2 = the EMAGE_H2 storage form: activations
and weights pre-split, what the f16x3 model runs since round 3; f16x3 = round 2's kernel on float32 activations)."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pantomatrix_amd import ops  # noqa: E402
from pantomatrix_amd._lib import BF16, F16X3, H2  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
dt = {"f16x3": F16X3, "h2": H2, "bf16": BF16}[mode]
td = torch.float32 if dt == H2 else ops.TORCH_DTYPE[dt]
g = torch.Generator().manual_seed(0)
dev = "cuda"


def lin(m, k, n, res=False):
    a = torch.randn(m, k, generator=g).to(td).to(dev)
    w = torch.randn(n, k, generator=g) / math.sqrt(k)
    if dt == H2:
        a = ops.h2_pack(a)
        wp, ws = ops.split_f16_weights_h2(w.to(dev))
    else:
        wp, ws = (ops.split_f16_weights(w) if dt == F16X3 else (w.to(td), 1.0))
    wp = wp.to(dev)
    bias = torch.zeros(n, device=dev)
    r = torch.randn(m, n, generator=g).to(td).to(dev) if res else None
    out = torch.empty(m, n, dtype=td, device=dev)
    for _ in range(4):
        ops.gemm(dt, a, wp, bias, None, r, out, None, None, n=n, cp=k, w_scale=ws)


lin(4096, 768, 768, res=True)      # out_proj: 64x192 tiles, one per CU
lin(4096, 768, 1536)               # ffn1
lin(4096, 768, 2304)               # qkv (without the V^T path)
lin(4096, 768, 256)                # head: 64x64 tiles
if dt == H2:
    torch.cuda.synchronize()
    sys.exit(0)
c, nseq, l = 64, 128, 1241
a = torch.randn(nseq * l, c, generator=g).to(td).to(dev)
w = torch.randn(c, 15 * c, generator=g) / math.sqrt(15 * c)
wp, ws = (ops.split_f16_weights(w) if dt == F16X3 else (w.to(td), 1.0))
wp = wp.to(dev)
bias, slope = torch.zeros(c, device=dev), torch.full((c,), 0.01, device=dev)
o1, o2 = torch.empty(nseq * l, c, dtype=td, device=dev), torch.empty(nseq * l, c, dtype=td, device=dev)
for _ in range(4):
    ops.gemm(dt, a, wp, bias, slope, None, o1, None, None, n=c, cp=c, res_first=True, taps=15, stride=1, pad=7, lin=l, lout=l, m=nseq * l, w_scale=ws)
    ops.conv_slab(dt, a, wp, bias, slope, None, o2, nseq=nseq, l=l, taps=15, pad=7, w_scale=ws)
torch.cuda.synchronize()

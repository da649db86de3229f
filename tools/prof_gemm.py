#!/usr/bin/env python
"""Run a few emage_gemm launches of fixed shapes/configurations (for rocprofv3 --pmc / --kernel-trace)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pantomatrix_amd import _lib, ops  # noqa: E402
from pantomatrix_amd._lib import BF16  # noqa: E402

lib = _lib.use_tools(True)      # tools build of the library: every tile configuration + emage_set_tuning
dev = "cuda"
g = torch.Generator().manual_seed(0)
cases = [("head", 768, 256), ("out_proj", 768, 768), ("qkv", 768, 2304), ("kv_all", 768, 12288)]
cfgs = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["19", "25", "20"])]
m = 4096
for name, k, n in cases:
    a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
    out = torch.zeros(m, n, dtype=torch.bfloat16, device=dev)
    for cfg in cfgs:
        lib.emage_set_tuning(0, cfg)
        for _ in range(5):
            ops.gemm(BF16, a, w, None, None, None, out, None, None, n=n, cp=k)
        torch.cuda.synchronize()
print("done")

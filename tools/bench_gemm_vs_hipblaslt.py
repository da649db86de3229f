#!/usr/bin/env python
"""Calibration of emage_gemm against the vendor GEMM — a TOOLS-ONLY yardstick: the product path never calls
hipBLASLt / rocBLAS; `torch.mm` / `torch.nn.functional.linear` (which dispatch to them on ROCm) are timed here beside `emage_gemm` in
bf16 and in the EMAGE_H2 split-fp16 mode, on one large shape (8192^3: the kernel's ceiling as a fraction of the 2.5 PF dense peak) and on
the six contraction shapes of a 64-clip window (M = 4096).  Every variant is a captured hipGraph of `--iters` launches timed with HIP
events; inputs are uniform random (not zeros: DVFS).  Usage (MI355X): python tools/bench_gemm_vs_hipblaslt.py > profiles/r04_gemm_vs_hipblaslt.txt"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pantomatrix_amd import ops  # noqa: E402
from pantomatrix_amd._lib import BF16, H2  # noqa: E402

PEAK_TF = 2500.0
SHAPES = [("8192^3", 8192, 8192, 8192), ("out_proj", 4096, 768, 768), ("ffn1", 4096, 1536, 768), ("ffn2", 4096, 768, 1536),
          ("qkv", 4096, 2304, 768), ("kv_all", 4096, 12288, 768), ("conv3/head (N=256, K=768)", 4096, 256, 768)]


def timed(call, iters):
    call()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(iters):
            call()
    gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    print("us per launch (captured hipGraph of %d launches, best of 3 replays); TF/s = 2 M N K / time; %% of 2.5 PF dense fp16 / bf16 MFMA" % args.iters)
    print("emage H2 issues 3 MFMAs per product: its MFMA-issue fraction is 3x the algorithmic figure shown")
    print(f"{'shape':28s} {'M':>6s} {'N':>6s} {'K':>6s} | {'vendor bf16':>22s} | {'vendor fp16':>22s} | {'emage_gemm bf16':>22s} | {'emage_gemm H2 (f16x3)':>22s}")
    for name, m, n, k in SHAPES:
        iters = 5 if m * n * k > 1e11 else args.iters
        a = (torch.rand(m, k, generator=g) * 2 - 1).to(dev)
        w = ((torch.rand(n, k, generator=g) * 2 - 1) / k ** 0.5).to(dev)
        flops = 2.0 * m * n * k
        cells = []
        for td in (torch.bfloat16, torch.float16):
            at, wt = a.to(td), w.to(td)
            out = torch.empty(m, n, dtype=td, device=dev)
            us = timed(lambda: torch.mm(at, wt.t(), out=out), iters)
            cells.append(us)
        ab, wb = a.to(torch.bfloat16), w.to(torch.bfloat16).contiguous()
        ob = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        cells.append(timed(lambda: ops.gemm(BF16, ab, wb, None, None, None, ob, None, None, n=n, cp=k), iters))
        ah = ops.h2_pack(a)
        wh, ws = ops.split_f16_weights_h2(w)
        oh = torch.empty(m, n, dtype=torch.float32, device=dev)
        cells.append(timed(lambda: ops.gemm(H2, ah, wh, None, None, None, oh, None, None, n=n, cp=k, w_scale=ws), iters))
        fmt = lambda us: f"{us:8.1f} us {flops / us / 1e6:6.0f} TF {100 * flops / us / 1e6 / PEAK_TF:4.1f}%"
        print(f"{name:28s} {m:6d} {n:6d} {k:6d} | " + " | ".join(fmt(us) for us in cells), flush=True)
        del a, w, ab, wb, ob, ah, wh, oh
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

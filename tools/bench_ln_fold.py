#!/usr/bin/env python
"""What the LayerNorm fold costs per launch (run on the MI355X): the contractions of a decoder layer at the window shape (M = 4096, d = 768), each
timed as a captured graph of `iters` launches — plain (round 5's form) against the folding variants of include/emage_hip.h: emage_gemm_problem."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pantomatrix_amd import ops  # noqa: E402
from pantomatrix_amd._lib import H2  # noqa: E402

dev, m, d, t, iters = "cuda", 4096, 768, 64, 20
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g)


def pack(n, k):
    w, ws = ops.split_f16_weights_h2((rnd(n, k) / math.sqrt(k)).to(dev))
    return w, ws, (rnd(n) * 0.1).to(dev)


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


x_img = ops.h2_pack(rnd(m, d)).to(dev)
a_img = ops.h2_pack(rnd(m, d)).to(dev)
f_img = ops.h2_pack(rnd(m, 2 * d)).to(dev)
st = torch.zeros(m, d // 32, 2, device=dev)
st[:, :, 1] = 32.0
st2 = torch.zeros(m, d // 32, 2, device=dev)
gamma, beta, cvec = torch.ones(d, device=dev), torch.zeros(d, device=dev), torch.zeros(3 * d, device=dev)
out_h, out_f = torch.zeros(m, d, device=dev), torch.zeros(m, d, device=dev)
wide_h, wide_f = torch.zeros(m, 2 * d, device=dev), torch.zeros(m, 2 * d, device=dev)
vt = torch.zeros(m // t, d, t, device=dev)
wo, wos, bo = pack(d, d)
w2, w2s, b2 = pack(d, 2 * d)
wq, wqs, bq = pack(d, d)
w1, w1s, b1 = pack(2 * d, d)
wqkv, wqkvs, bqkv = pack(3 * d, d)
relu = torch.zeros(2 * d, device=dev)
rows = [
    ("out_proj: res H2, out f32 (round 5)", lambda: ops.gemm(H2, a_img, wo, bo, None, x_img, None, out_f, None, n=d, cp=d, w_scale=wos, res_h2=True)),
    ("out_proj: res H2, out H2", lambda: ops.gemm(H2, a_img, wo, bo, None, x_img, out_h, None, None, n=d, cp=d, w_scale=wos, res_h2=True)),
    ("out_proj: res H2, out H2 + stats_out", lambda: ops.gemm(H2, a_img, wo, bo, None, x_img, out_h, None, None, n=d, cp=d, w_scale=wos, res_h2=True, stats_out=st2)),
    ("out_proj: folded res, out H2", lambda: ops.gemm(H2, a_img, wo, bo, None, x_img, out_h, None, None, n=d, cp=d, w_scale=wos, res_h2=True, res_ln=(st, gamma, beta))),
    ("out_proj: folded res, out H2 + stats_out", lambda: ops.gemm(H2, a_img, wo, bo, None, x_img, out_h, None, None, n=d, cp=d, w_scale=wos, res_h2=True, res_ln=(st, gamma, beta), stats_out=st2)),
    ("ffn2: res H2, out f32 (round 5)", lambda: ops.gemm(H2, f_img, w2, b2, None, x_img, None, out_f, None, n=d, cp=2 * d, w_scale=w2s, res_h2=True)),
    ("ffn2: folded res, out H2 + stats_out", lambda: ops.gemm(H2, f_img, w2, b2, None, x_img, out_h, None, None, n=d, cp=2 * d, w_scale=w2s, res_h2=True, res_ln=(st, gamma, beta), stats_out=st2)),
    ("ca.q: out f32 (round 5)", lambda: ops.gemm(H2, x_img, wq, bq, None, None, None, out_f, None, n=d, cp=d, w_scale=wqs)),
    ("ca.q: ln fold", lambda: ops.gemm(H2, x_img, wq, bq, None, None, None, out_f, None, n=d, cp=d, w_scale=wqs, ln=(st, cvec[:d]))),
    ("ffn1: relu, out H2 (round 5)", lambda: ops.gemm(H2, x_img, w1, b1, relu, None, wide_h, None, None, n=2 * d, cp=d, w_scale=w1s)),
    ("ffn1: ln fold", lambda: ops.gemm(H2, x_img, w1, b1, relu, None, wide_h, None, None, n=2 * d, cp=d, w_scale=w1s, ln=(st, cvec[:2 * d]))),
    ("qkv + V^T (round 5)", lambda: ops.gemm(H2, x_img, wqkv, bqkv, None, None, None, wide_f, vt, n=3 * d, cp=d, w_scale=wqkvs, t_col0=2 * d, t_rows=t)),
    ("qkv + V^T: ln fold", lambda: ops.gemm(H2, x_img, wqkv, bqkv, None, None, None, wide_f, vt, n=3 * d, cp=d, w_scale=wqkvs, t_col0=2 * d, t_rows=t, ln=(st, cvec))),
    ("layernorm f32 -> H2 (round 5)", lambda: ops.layernorm(H2, out_f, gamma, beta, 1e-5, None, None, out_h)),
]
dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if dbg:                                  # timing ablations of the fold (tools library, emage_set_tuning key 1: 256 = no statistics loads, 512 = no merge, 1024 = no vector loads)
    from pantomatrix_amd import _lib
    _lib.use_tools(True).emage_set_tuning(1, dbg)
    print(f"== tools library, dbg {dbg}")
for name, call in rows:
    print(f"{name:45s} {timed(call):7.2f} us", flush=True)

#!/usr/bin/env python
"""CPU study behind the f16x3 mode: the oracle's matmuls / convs replaced by emulated operand formats,
VQ code flips and latent error against the plain fp32 oracle on N synthetic 128-frame clips.
    python tools/split_precision_study.py [clips=8] [modes: f16x3 bf16x3 bf16 ...]
f16x3 / bf16x3: x = hi + lo in two fp16 / bf16 planes, hi*hi + hi*lo + lo*hi; bf16: single bf16 operands.
f16x3+lnfold (round 5): f16x3 with every LayerNorm FOLDED into the Linear that consumes it — the contraction runs
on the RAW (un-normalised) row x with W' = W * gamma, and the epilogue applies the row statistics:
    LN(x) W^T + b = r * (x W'^T - mu * c) + d,   c[n] = sum_k W'[n][k],   d = W beta + b,   mu / r = the row's mean / rstd (fp32)
(what a fused kernel would compute: no LayerNorm launch, 94 per step; the residual stream still uses the exact fp32 LN(x)).  The
question it answers: does the cancellation in x W'^T - mu c cost VQ code indices?"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import common  # noqa: E402
from oracle import emage_oracle as orc  # noqa: E402
from pantomatrix_amd import synthetic  # noqa: E402

MODE = None
SA = 16.0


def split(x, dt, scale):
    x = x * scale
    hi = x.to(dt).float()
    return hi, (x - hi).to(dt).float()


def mm3(a, w, op):
    if MODE in ("f16x3", "f16x3+lnfold"):
        m = float(w.abs().max())
        sw = 2.0 ** (12 - math.floor(math.log2(m))) if m > 0 else 1.0
        ah, al = split(a, torch.float16, SA)
        wh, wl = split(w, torch.float16, sw)
        assert torch.isfinite(ah).all(), float(a.abs().max())
        return (op(ah, wh) + (op(ah, wl) + op(al, wh))) / (SA * sw)
    if MODE == "bf16x3":
        ah, al = split(a, torch.bfloat16, 1.0)
        wh, wl = split(w, torch.bfloat16, 1.0)
        return op(ah, wh) + (op(ah, wl) + op(al, wh))
    if MODE == "bf16":
        return op(a.bfloat16().float(), w.bfloat16().float())
    raise ValueError(MODE)


class Shim:
    def __init__(self):
        self.ln = {}                 # id(LayerNorm output) -> (output kept alive, raw x, mu, rstd, gamma, beta)
        self.folded = 0

    def __getattr__(self, n):
        return getattr(F, n)

    def layer_norm(self, x, shape, weight=None, bias=None, eps=1e-5):
        y = F.layer_norm(x, shape, weight, bias, eps)
        if MODE == "f16x3+lnfold":
            mu = x.mean(-1, keepdim=True)
            r = torch.rsqrt(x.var(-1, unbiased=False, keepdim=True) + eps)
            if len(self.ln) > 64:
                self.ln.clear()
            self.ln[id(y)] = (y, x, mu, r, weight, bias)
        return y

    def linear(self, x, w, b=None):
        if MODE is None:
            return F.linear(x, w, b)
        hit = self.ln.get(id(x)) if MODE == "f16x3+lnfold" else None
        if hit is not None and hit[0] is x:
            _y, raw, mu, r, gamma, beta = hit
            wp = w * gamma[None, :]
            c = wp.sum(1)
            d = w @ beta + (b if b is not None else 0.0)
            acc = mm3(raw, wp, lambda p, q: F.linear(p, q))
            self.folded += 1
            return r * (acc - mu * c) + d
        r = mm3(x, w, lambda p, q: F.linear(p, q))
        return r if b is None else r + b

    def conv1d(self, x, w, b=None, stride=1, padding=0):
        if MODE is None or w.shape[1] == 1:
            return F.conv1d(x, w, b, stride=stride, padding=padding)
        r = mm3(x, w, lambda p, q: F.conv1d(p, q, None, stride=stride, padding=padding))
        return r if b is None else r + b[None, :, None]


def main():
    global MODE
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    shim = Shim()
    orc.F = shim
    omodel, ovq = common.oracle_models()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    audio = synthetic.synthetic_audio(n, synthetic.samples_for_frames(128))

    def run():
        with torch.no_grad():
            lat = omodel.inference(audio, torch.zeros(n, 1, dtype=torch.long), ovq)
            return lat, omodel.select_codes(lat)

    t = time.time()
    ref = run()
    print(f"{n} clips, fp32 oracle: {time.time() - t:.1f} s")
    for mode in (sys.argv[2:] or ["f16x3", "bf16x3", "bf16"]):
        MODE = mode
        shim.folded = 0
        out = run()
        if mode == "f16x3+lnfold":
            print(f"  LayerNorm-folded contractions in this run: {shim.folded}")
        MODE = None
        flips = {p: int((out[1][f"{p}_index"] != ref[1][f"{p}_index"]).sum()) for p in ("upper", "hands", "lower")}
        rel = {k: float((out[0][k] - ref[0][k]).norm() / ref[0][k].norm()) for k in ("rec_face", "cls_upper", "cls_hands", "cls_lower")}
        print(f"{mode:7s} code flips {flips} of {ref[1]['upper_index'].numel()} each; relative error " + ", ".join(f"{k} {v:.2e}" for k, v in rel.items()))


if __name__ == "__main__":
    main()

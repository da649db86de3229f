#!/usr/bin/env python
"""Per-kernel A/B of the LDS-resident-slab convolutions against the generic path (run on the MI355X), graph-timed:
block 0 fused (emage_wav_block0) vs emage_wav_conv_in + emage_gemm, and emage_conv_slab vs emage_gemm on the stride-1
k = 15 layer shapes of the EMAGE WavEncoder at the hoisted batch (128 sequences)."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pantomatrix_amd import ops  # noqa: E402
from pantomatrix_amd._lib import BF16, F32, F16X3  # noqa: E402

DEV = "cuda"


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    gen = torch.Generator().manual_seed(0)
    for name, dt in (("f16x3", F16X3), ("bf16", BF16), ("fp32", F32)):
        td = ops.TORCH_DTYPE[dt]
        sd = F32 if dt == F16X3 else dt                      # storage / elementwise dtype code
        # ---- block 0: 64 clips x 2 windows -> 128 sequences of 7460 positions, C = 64
        c, nclip, nwin, win, hop, lout = 64, 64, 2, 34112, 31980, 7460
        wav = (0.1 * torch.randn(nclip, win + hop, generator=gen)).to(DEV)
        wf = (torch.randn(2 * c, 15, generator=gen) / 4).to(DEV)
        bf = (torch.randn(2 * c, generator=gen) * 0.1).to(DEV)
        sf = torch.cat([torch.full((c,), 0.01), torch.ones(c)]).to(DEV)
        w2 = (torch.randn(c, 15 * c, generator=gen) / math.sqrt(15 * c))
        w2p, ws = (ops.split_f16_weights(w2) if dt == F16X3 else (w2.to(td), 1.0))
        w2p = w2p.to(DEV)
        b2, s2 = (torch.randn(c, generator=gen) * 0.1).to(DEV), torch.full((c,), 0.01, device=DEV)
        nseq = nclip * nwin
        y0 = torch.zeros(nseq * lout, 2 * c, dtype=td, device=DEV)
        ref = torch.zeros(nseq * lout, c, dtype=td, device=DEV)
        got = torch.zeros(nseq * lout, c, dtype=td, device=DEV)

        def unfused():
            ops.wav_conv_in(sd, wav, wf, bf, sf, y0, lout, 5, 1600, nwin=nwin, hop=hop, win_len=win)
            ops.gemm(dt, y0[:, :c], w2p, b2, s2, y0[:, c:], ref, None, None, n=c, cp=c, res_first=True, taps=15, stride=1, pad=7,
                     lin=lout, lout=lout, m=nseq * lout, w_scale=ws)

        def fused():
            ops.wav_block0(dt, wav, wf[:c], bf[:c], 0.01, wf[c:], bf[c:], 5, 1600, w2p, b2, s2, 15, 7, got, lout, nwin=nwin, hop=hop, win_len=win, w_scale=ws)

        tu, tf = timed(unfused, 5), timed(fused, 5)
        flops = 2.0 * nseq * lout * c * 15 * c
        print(f"{name:6s} block0 C=64 128 seq x 7460 : unfused {tu:8.1f} us  fused {tf:8.1f} us  ({flops / tf / 1e6:6.0f} TF/s algorithmic)  equal={torch.equal(ref, got)}")
        # ---- stride-1 layers
        for c, l in ((64, 1241), (128, 205)):
            a = torch.randn(nseq * l, c, generator=gen).to(td).to(DEV)
            w = torch.randn(c, 15 * c, generator=gen) / math.sqrt(15 * c)
            wp, ws = (ops.split_f16_weights(w) if dt == F16X3 else (w.to(td), 1.0))
            wp = wp.to(DEV)
            bias, slope = (torch.randn(c, generator=gen) * 0.1).to(DEV), torch.full((c,), 0.01, device=DEV)
            sc = torch.randn(nseq * l, c, generator=gen).to(td).to(DEV)
            r1, r2 = torch.zeros(nseq * l, c, dtype=td, device=DEV), torch.zeros(nseq * l, c, dtype=td, device=DEV)
            tg = timed(lambda: ops.gemm(dt, a, wp, bias, slope, sc, r1, None, None, n=c, cp=c, res_first=True, taps=15, stride=1, pad=7, lin=l, lout=l, m=nseq * l, w_scale=ws))
            tsl = timed(lambda: ops.conv_slab(dt, a, wp, bias, slope, sc, r2, nseq=nseq, l=l, taps=15, pad=7, w_scale=ws))
            flops = 2.0 * nseq * l * c * 15 * c
            print(f"{name:6s} conv C={c:3d} 128 seq x {l:5d} : gemm    {tg:8.1f} us  slab  {tsl:8.1f} us  ({flops / tsl / 1e6:6.0f} TF/s algorithmic)  equal={torch.equal(r1, r2)}")


if __name__ == "__main__":
    main()

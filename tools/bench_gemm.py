#!/usr/bin/env python
"""Sweep emage_gemm tile configurations over the shapes of the EMAGE window (run on the MI355X).
Checks each configuration against configuration 0 (the register-staged kernel validated by the parity tests)
and prints HIP-event timings.  Usage: python tools/bench_gemm.py [--dtype bf16|fp32]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pantomatrix_amd import _lib, ops  # noqa: E402
from pantomatrix_amd._lib import BF16, F32, F16X3  # noqa: E402

SHAPES = [
    # name, (nb, lin, lout), cin, n, taps, stride, pad, extras
    ("qkv 768->2304 +vt", (64, 64, 64), 768, 2304, 1, 1, 0, dict(vt=1536)),
    ("out_proj 768->768 +res", (64, 64, 64), 768, 768, 1, 1, 0, dict(res=True)),
    ("ffn1 768->1536 relu", (64, 64, 64), 768, 1536, 1, 1, 0, dict(slope=0.0)),
    ("ffn2 1536->768 +res", (64, 64, 64), 1536, 768, 1, 1, 0, dict(res=True)),
    ("kv_all 768->12288 +vt", (64, 64, 64), 768, 12288, 1, 1, 0, dict(vt=6144)),
    ("head 768->256", (64, 64, 64), 768, 256, 1, 1, 0, dict()),
    ("mlp 256->768", (64, 64, 64), 256, 768, 1, 1, 0, dict(slope=0.1)),
    ("conv3 256->256", (64, 64, 64), 256, 256, 3, 1, 1, dict(slope=0.2)),
    ("conv3 337->256", (64, 64, 64), 337, 256, 3, 1, 1, dict(slope=0.2)),
    ("conv3 T=120 256->256", (64, 120, 120), 256, 256, 3, 1, 1, dict(slope=0.2)),
    ("vq conv3 seed T=17 256->256", (64, 17, 17), 256, 256, 3, 1, 1, dict(slope=0.2)),
    ("cls fc 256->256 M=4096", (64, 64, 64), 256, 256, 1, 1, 0, dict(slope=0.1)),
    ("wav b0.conv2 64->64 k15", (64, 7460, 7460), 64, 64, 15, 1, 7, dict(slope=0.01)),
    ("wav b1.conv1+ds 64->128 s6", (64, 7460, 1241), 64, 128, 15, 6, 0, dict()),
    ("wav b1.conv2 64->64", (64, 1241, 1241), 64, 64, 15, 1, 7, dict(slope=0.01)),
    ("wav b3.conv1+ds 64->256 s6", (64, 1241, 205), 64, 256, 15, 6, 0, dict()),
    ("wav b4.conv2 128->128", (64, 205, 205), 128, 128, 15, 1, 7, dict()),
    ("wav b5.conv1+ds 128->512 s3", (64, 205, 64), 128, 512, 15, 3, 0, dict()),
    ("wav b5.conv2 256->256", (64, 64, 64), 256, 256, 15, 1, 7, dict()),
]
CONFIGS = [25, 18, 27, 32, 33, 34, 36, 37]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--ablate", action="store_true", help="time each config with DMA / MFMA / epilogue removed (diagnostic)")
    ap.add_argument("--configs", default="")
    ap.add_argument("--dbg", type=int, default=0, help="emage_set_tuning key 1 mask for the whole sweep (8: sc1 result stores, 16: nt result stores)")
    args = ap.parse_args()
    dt = {"bf16": BF16, "fp32": F32, "f16x3": F16X3}[args.dtype]
    td = ops.TORCH_DTYPE[dt]
    lib = _lib.use_tools(True)      # tools build of the library: every tile configuration + emage_set_tuning
    lib.emage_set_tuning(1, args.dbg)
    dev = "cuda"
    global CONFIGS
    if args.configs:
        CONFIGS = [int(c) for c in args.configs.split(",")]
    g = torch.Generator().manual_seed(0)
    print(f"{'shape':32s} GF  | " + " ".join(f"c{c:<6d}" for c in CONFIGS) + " | best")
    for name, (nb, lin, lout), cin, n, taps, stride, pad, ex in SHAPES:
        cp = ops.round_up(cin, 64)
        m = nb * lout
        a = torch.zeros(nb * lin, cp)
        a[:, :cin] = torch.randn(nb * lin, cin, generator=g)
        w = torch.zeros(n, taps, cp)
        w[:, :, :cin] = torch.randn(n, taps, cin, generator=g) / (cin * taps) ** 0.5
        a, w = a.to(td).to(dev), w.reshape(n, taps * cp).to(td)
        w_scale = 1.0
        if dt == F16X3:
            w, w_scale = ops.split_f16_weights(w)
        w = w.to(dev)
        bias = (torch.randn(n, generator=g) * 0.1).to(dev)
        slope = torch.full((n,), float(ex["slope"]), device=dev) if "slope" in ex else None
        res = torch.randn(m, n, generator=g).to(dev) if ex.get("res") else None
        vt0 = ex.get("vt")
        ncol = vt0 or n
        flops = 2.0 * m * n * taps * cin

        def run(cfg):
            lib.emage_set_tuning(0, cfg)
            out = torch.zeros(m, ncol, dtype=td, device=dev)
            out_f = torch.zeros(m, ncol, device=dev) if res is not None else None
            out_t = torch.zeros(nb, n - vt0, ops.round_up(lout, 32), dtype=td, device=dev) if vt0 else None
            call = lambda: ops.gemm(dt, a, w, bias, slope, res, out, out_f, out_t, n=n, cp=cp, t_col0=vt0 or 0,
                                    t_rows=lout if vt0 else 0, taps=taps, stride=stride, pad=pad, lin=lin, lout=lout, m=m, w_scale=w_scale)
            call()
            torch.cuda.synchronize()
            # replay a captured graph of `iters` launches so the host launch cost (~8 us/op from Python) is excluded
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(args.iters):
                    call()
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / args.iters * 1e3, out, out_f, out_t

        if args.ablate:
            row = []
            for cfg in CONFIGS:
                cell = []
                for dbg in (0, 1, 2, 4, 7):
                    lib.emage_set_tuning(1, dbg)
                    cell.append(run(cfg)[0])
                lib.emage_set_tuning(1, 0)
                row.append(f"c{cfg}: full {cell[0]:.1f} noDMA {cell[1]:.1f} noMFMA {cell[2]:.1f} noEpi {cell[3]:.1f} none {cell[4]:.1f}")
            print(f"{name:30s} " + " | ".join(row))
            continue
        ref = None
        cells, best = [], (1e9, None)
        for cfg in CONFIGS:
            try:
                us, out, out_f, out_t = run(cfg)
            except Exception as e:  # noqa: BLE001
                cells.append("ERR    ")
                continue
            if ref is None:
                ref = (out.float(), None if out_f is None else out_f.clone(), None if out_t is None else out_t.float())
                ok = True
            else:
                ok = (out.float() - ref[0]).abs().max().item() <= 0.05
                if out_f is not None:
                    ok &= (out_f - ref[1]).abs().max().item() <= 1e-3
                if out_t is not None:
                    ok &= (out_t.float() - ref[2]).abs().max().item() <= 0.05
            cells.append(f"{us:6.1f}{' ' if ok else '!'}")
            if ok and us < best[0]:
                best = (us, cfg)
        lib.emage_set_tuning(0, -1)
        print(f"{name:32s} {flops / 1e9:5.1f}| " + " ".join(cells) + f" | c{best[1]} {flops / best[0] / 1e6:6.0f} TF/s")


if __name__ == "__main__":
    main()

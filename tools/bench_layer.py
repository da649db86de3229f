"""Time emage_transformer_layer (one launch per layer) against the per-op launch sequence, B clips x 64 frames, inside a
hipGraph of `reps` chained layers; optional ring depths and ablation masks (1 no spin, 2 no fences, 4 no tiles,
8 no attention, 16 no LayerNorm — timing only).  Usage: python tools/bench_layer.py [--b 64] [--reps 8]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import common  # noqa: E402
from pantomatrix_amd import _lib, spec  # noqa: E402
from pantomatrix_amd import modeling_emage_audio as M  # noqa: E402


def timed_graph(fn, iters=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=64)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--waves", type=int, default=4, help="waves per workgroup for the ablation runs (4 or 8)")
    ap.add_argument("--masks", type=str, default="0,4,8,16,24,256,512,1024,1792")
    args = ap.parse_args()
    model, _ = common.product_models(precision="bf16", device="cuda")
    lib = _lib.load()
    cx = M._Ctx(model._engine())
    b, t, d, nc = args.b, 64, 768, spec.N_CROSS_LAYERS
    g = torch.Generator().manual_seed(0)
    x = torch.randn(b * t, d, generator=g).to(torch.bfloat16).cuda()
    mem = torch.randn(b * t, d, generator=g).to(torch.bfloat16).cuda()
    bk, bvt = model._memory_kv(cx, "cross.kv_all", mem, b, t, nc)

    def chain():
        y = x
        for i in range(args.reps):
            li = i % nc
            y = model._decoder_layer(cx, f"audio_motion_cross_attn.layers.{li}", y, b, t, bk[:, li * d:(li + 1) * d],
                                     bvt[:, li * d:], nc * d, t)
        return y

    model.fused_layers = False
    base = timed_graph(chain) / args.reps
    print(f"per-op sequence : {1e3 * base:8.1f} us / decoder layer   (B={b})")
    model.fused_layers = True
    for waves in (4, 8):
        lib.emage_layer_set_tuning(2, waves)
        for ring in (2, 3):
            lib.emage_layer_set_tuning(0, ring)
            ms = timed_graph(chain) / args.reps
            print(f"fused, {waves} waves, ring {ring}: {1e3 * ms:8.1f} us / decoder layer")
    lib.emage_layer_set_tuning(0, 3)
    lib.emage_layer_set_tuning(2, args.waves)
    lib.emage_layer_set_tuning(3, 1)
    ms = timed_graph(chain) / args.reps
    print(f"fused, {args.waves} waves, ring 3, W prefetch before barriers (needs a build with -DEMAGE_LAYER_PREFETCH=1): {1e3 * ms:8.1f} us / decoder layer")
    lib.emage_layer_set_tuning(3, 0)
    for mask in [int(v) for v in args.masks.split(",")]:
        lib.emage_layer_set_tuning(1, mask)
        ms = timed_graph(chain) / args.reps
        print(f"fused, {args.waves} waves, ring 3, ablation mask {mask:4d}: {1e3 * ms:8.1f} us / decoder layer")
    lib.emage_layer_set_tuning(1, 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Do two independent chains of the same GEMM overlap on the MI355X?  (diagnostic for the stream-lane design)"""
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


def chain_inputs(m, k, n):
    a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
    return a, w, torch.zeros(m, n, dtype=torch.bfloat16, device=dev)


def run(nchains, m, k, n, iters=20, cfg=-1):
    lib.emage_set_tuning(0, cfg)
    ins = [chain_inputs(m, k, n) for _ in range(nchains)]
    streams = [torch.cuda.Stream() for _ in range(nchains)]

    def body():
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(main)
        for s, (a, w, o) in zip(streams, ins):
            s.wait_event(ev)
            with torch.cuda.stream(s):
                for _ in range(iters):
                    ops.gemm(BF16, a, w, None, None, None, o, None, None, n=n, cp=k)
            e2 = torch.cuda.Event(); e2.record(s); main.wait_event(e2)
    body(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        body()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


cfgs = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["-1"])]
for (m, k, n) in ((4096, 768, 768), (4096, 768, 1536), (4096, 768, 2304), (4096, 1536, 768)):
    for cfg in cfgs:
        t1, t2, t4 = run(1, m, k, n, cfg=cfg), run(2, m, k, n, cfg=cfg), run(4, m, k, n, cfg=cfg)
        print(f"M={m} K={k} N={n} cfg {cfg:3d}: us per launch-slot  1 chain {t1:6.1f} | 2 chains {t2:6.1f} -> {t2 / 2:5.1f}/launch | 4 chains {t4:6.1f} -> {t4 / 4:5.1f}/launch")


def run_separate_graphs(nchains, m, k, n, iters=20, cfg=-1):
    """Same chains, but each captured in ITS OWN graph and the graphs replayed on different streams."""
    lib.emage_set_tuning(0, cfg)
    ins = [chain_inputs(m, k, n) for _ in range(nchains)]
    streams = [torch.cuda.Stream() for _ in range(nchains)]
    graphs = []
    for (a, w, o) in ins:
        ops.gemm(BF16, a, w, None, None, None, o, None, None, n=n, cp=k)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(iters):
                ops.gemm(BF16, a, w, None, None, None, o, None, None, n=n, cp=k)
        graphs.append(gr)

    def go():
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(main)
        for s, gr in zip(streams, graphs):
            s.wait_event(ev)
            with torch.cuda.stream(s):
                gr.replay()
            e2 = torch.cuda.Event(); e2.record(s); main.wait_event(e2)
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


if len(sys.argv) > 2 and sys.argv[2] == "separate":
    for (m, k, n) in ((4096, 768, 768), (4096, 768, 1536)):
        for cfg in cfgs:
            a1, a2, a4 = run(1, m, k, n, cfg=cfg), run(2, m, k, n, cfg=cfg), run(4, m, k, n, cfg=cfg)
            s2, s4 = run_separate_graphs(2, m, k, n, cfg=cfg), run_separate_graphs(4, m, k, n, cfg=cfg)
            print(f"N={n} cfg {cfg}: one graph: 1ch {a1:.1f} 2ch {a2:.1f} 4ch {a4:.1f} | separate graphs: 2 -> {s2:.1f}  4 -> {s4:.1f}")

#!/usr/bin/env python
"""Phase timeline of one block of an instrumented EMAGE_H2 tile kernel (configs 201 / 203 / 205 / 241): s_memtime stamps per wave."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pantomatrix_amd import _lib, ops
from pantomatrix_amd._lib import H2

lib = _lib.use_tools(True)      # tools build of the library: every tile configuration + emage_set_tuning
raw = lib

dev = "cuda"
m, k, n = 4096, 768, 768
g = torch.Generator().manual_seed(0)
a = ops.h2_pack(torch.randn(m, k, generator=g).to(dev))
w, ws = ops.split_f16_weights_h2((torch.randn(n, k, generator=g) / k ** 0.5).to(dev))
bias = torch.randn(n, generator=g).to(dev)
res = torch.randn(m, n, generator=g).to(dev)
out, outf = torch.zeros(m, n, device=dev), torch.zeros(m, n, device=dev)
for cfg in [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "201,203,205,241").split(",")]:
    trace = torch.zeros(16 * 512, dtype=torch.int64, device=dev)
    raw.emage_h2_set_trace(C.c_void_p(trace.data_ptr()))
    lib.emage_set_tuning(4, cfg)
    for _ in range(3):
        ops.gemm(H2, a, w, bias, None, res, out, outf, None, n=n, cp=k, w_scale=ws)
    torch.cuda.synchronize()
    t = trace.cpu().view(16, 512)
    print(f"== config {cfg}")
    t0 = min(int(t[wv, 1]) for wv in range(16) if int(t[wv, 0]) > 0)
    for wv in range(16):
        cnt = int(t[wv, 0])
        if cnt == 0:
            continue
        ev = [int(x) - t0 for x in t[wv, 1:1 + cnt]]
        print(f"wave {wv:2d} n={cnt} first={ev[0]} last={ev[-1]}")
        d = [ev[i + 1] - ev[i] for i in range(len(ev) - 1)]
        print("   deltas:", " ".join(str(x) for x in d[:120]))
        print("   tail  :", " ".join(str(x) for x in d[-12:]))
lib.emage_set_tuning(4, -1)

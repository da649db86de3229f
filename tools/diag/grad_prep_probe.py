"""Where does emage_grad_prep differ from the separate launches?  (diagnostic; GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from pantomatrix_amd import ops
DEV = torch.device("cuda", 0)
g = torch.Generator().manual_seed(11)
for m, c, slope in ((3584, 768, None), (130, 256, 0.1), (70, 337, 0.0), (64, 64, None), (1, 8, 0.2), (200, 1536, 0.0)):
    base = torch.randn(m, c + 24, generator=g).to(DEV)
    dy = base[:, 8:8 + c]
    y = torch.randn(m, c, generator=g).to(DEV)
    y[0, 0] = 0.0
    n_store, m_store, scale = ops.round_up(c, 64), ops.round_up(m, 64), 1024.0
    dpre = dy if slope is None else ops.act_backward(dy.contiguous(), y, slope)
    want_h = ops.h2_cast(dpre, n_store, scale=scale)
    want_t = ops.h2_cast(dpre, m_store, scale=scale, transpose=True)
    got_h, got_t, got_b = ops.grad_prep(dy, None if slope is None else y, 0.0 if slope is None else slope, scale, n_store=n_store, m_store=m_store)
    torch.cuda.synchronize()
    for tag, a, b in (("h", got_h, want_h), ("t", got_t, want_t)):
        d = (a.view(torch.int32) != b.view(torch.int32))
        if bool(d.any()):
            idx = d.nonzero()
            print(m, c, slope, tag, "mismatches", int(d.sum()), "of", d.numel(), "first", idx[:4].tolist(), "last", idx[-2:].tolist(),
                  "rows", int(idx[:, 0].min()), int(idx[:, 0].max()), "cols", int(idx[:, 1].min()), int(idx[:, 1].max()))
            r, cc = idx[0].tolist()
            print("   got", a.view(torch.int32)[r, cc - cc % 8:cc - cc % 8 + 8].tolist(), "want", b.view(torch.int32)[r, cc - cc % 8:cc - cc % 8 + 8].tolist())
        else:
            print(m, c, slope, tag, "equal")

#!/usr/bin/env python
"""The EMAGE training step at BASELINE configs[2]'s per-GPU batch, alone (what bench.py's `train_step` object times) — for rocprofv3:
    rocprofv3 --kernel-trace --stats -d <dir> -o train --output-format csv -- python tools/bench_train_step.py [--cpu]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cpu = "--cpu" in sys.argv
quick = "--quick" in sys.argv                   # captured step only (A/B runs): no eager timing
acc = None if "--accumulate-dw" not in sys.argv else int(sys.argv[sys.argv.index("--accumulate-dw") + 1])
variant = 0 if "--h2-variant" not in sys.argv else int(sys.argv[sys.argv.index("--h2-variant") + 1])
ln_fused = None if "--ln-fused" not in sys.argv else int(sys.argv[sys.argv.index("--ln-fused") + 1])
defer = None if "--defer-finalize" not in sys.argv else int(sys.argv[sys.argv.index("--defer-finalize") + 1])
h2f = None if "--h2-forward" not in sys.argv else int(sys.argv[sys.argv.index("--h2-forward") + 1])
dcx = None if "--direct-conv-dx" not in sys.argv else int(sys.argv[sys.argv.index("--direct-conv-dx") + 1])
fga = None if "--fuse-grad-adds" not in sys.argv else int(sys.argv[sys.argv.index("--fuse-grad-adds") + 1])
sys.argv = ["bench.py"]
import bench  # noqa: E402

if variant:                                     # tools library: emage_set_tuning key 5 (EMAGE_H2 dispatch-heuristic variant)
    from pantomatrix_amd import _lib
    _lib.use_tools(True)
    _lib.load().emage_set_tuning(5, variant)
if ln_fused is not None:
    from pantomatrix_amd import ops
    ops.FUSED_LAYERNORM_BACKWARD = {0: False, 1: 16}.get(ln_fused, ln_fused)
line = bench.bench_train_step(torch.device("cuda", 0), cpu=cpu, eager=not quick, accumulate_dw=acc, defer_finalize=defer, exchange=not quick, h2_forward=h2f, direct_conv_dx=dcx, fuse_grad_adds=fga)
line["ab"] = {"accumulate_dw": acc, "h2_variant": variant, "ln_fused": ln_fused, "defer_finalize": defer, "h2_forward": h2f, "direct_conv_dx": dcx, "fuse_grad_adds": fga}
print(json.dumps(line))

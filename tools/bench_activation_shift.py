#!/usr/bin/env python
"""What the activation shift of the EMAGE_H2 path costs (round 6; include/emage_hip.h EMAGE_H2_SHIFT, `model.activation_shift`): for k = 0, 2, 4, 6, 8
the VQ-code agreement of the 64-clip BASELINE batch with the real reference's run (tests/golden/infer_128f_b64.npz — k = 0 must stay at 1.0) and the
captured step time (the shift is a launch argument: no kernel changes shape).  One JSON line.
    python tools/bench_activation_shift.py [--steps 20]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
steps = 20 if "--steps" not in sys.argv else int(sys.argv[sys.argv.index("--steps") + 1])
sys.argv = ["bench.py"]
import bench  # noqa: E402
from pantomatrix_amd import synthetic  # noqa: E402
from pantomatrix_amd.runtime import ClipRunner  # noqa: E402
from tools import workloads as common  # noqa: E402

dev = torch.device("cuda", 0)
model, vq = common.product_models(precision="f16x3", device=dev)
n = synthetic.samples_for_frames(128)
audio = synthetic.synthetic_audio(64, n).to(dev)
out = {"workload": "64 x 128-frame synthetic clips (BASELINE configs[1]), f16x3 / EMAGE_H2", "steps": steps, "shifts": {}}
for k in (0, 2, 4, 6, 8, 0):
    model.set_activation_shift(k)
    vq.set_activation_shift(k)
    agree = bench.code_agreement(model, vq, audio, 64)
    runner = ClipRunner(model, vq, 64, n, use_graph=True)
    runner.run_device(audio)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.run_device()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    key = str(k) if str(k) not in out["shifts"] else f"{k} (again)"
    out["shifts"][key] = {"activation_scale": 16.0 * 2.0 ** -k, "finite_below": 4094.0 * 2.0 ** k, "ms_per_step": ms,
                          "code_agreement": {p: agree[p] for p in ("upper", "hands", "lower", "face", "frames_with_all_body_codes_equal")}}
    del runner
    torch.cuda.empty_cache()
print(json.dumps(out))

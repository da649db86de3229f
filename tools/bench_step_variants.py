#!/usr/bin/env python
"""A/B of EMAGE_H2 dispatch variants on the 64-clip BASELINE step: `ClipRunner` graph replays, tools library, one arm per `emage_set_tuning` key-5 value
(comma-separated argument; 131072 = neutral, the shipped dispatch).  Arms are interleaved and repeated; results are compared bit for bit with the first arm.
    python tools/bench_step_variants.py 131072,4325376        # 4325376 = 131072 | 4194304: XCD-aware runs for every launch (round 4's tile order)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pantomatrix_amd import _lib, synthetic  # noqa: E402
from pantomatrix_amd.runtime import ClipRunner  # noqa: E402
from tools import workloads as common  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    lib = _lib.use_tools(True)
    arms = [int(x) for x in sys.argv[1].split(",")]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    b, n, steps = 64, synthetic.samples_for_frames(128), 30
    audio = synthetic.synthetic_audio(b, n, seed=1234).to(dev)
    first = None
    for rep in range(reps):
        for variant in arms:
            lib.emage_set_tuning(5, variant)
            model, vq = common.product_models(precision="f16x3", device=dev)
            runner = ClipRunner(model, vq, b, n)
            for _ in range(5):
                res = runner(audio)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                res = runner(audio)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / steps
            got = [np.array(r) for r in res]
            if first is None:
                first = got
            print(json.dumps({"h2_variant": variant, "rep": rep, "ms_per_step": round(ms, 3),
                              "same_bits": all(np.array_equal(a, r) for a, r in zip(first, got))}), flush=True)
            del runner, model, vq
            torch.cuda.empty_cache()
    lib.emage_set_tuning(5, 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""A/B of the EMAGE_H2 dispatch for grids of at most one tile per CU (config 188: a lone 64 x 64 block per CU with a ring of 8 K-tiles; round 5):
ONE clip (128 frames, 28 s), 8 clips and the 64-clip BASELINE batch through `runtime.ClipRunner` graph replays, tools library,
`emage_set_tuning` key 5 = 1048576 (config 189: the tile on 8 waves) / 524288 (config 188 for such grids) against 131072 (neutral: the shipped dispatch, every such launch on the three-blocks-per-CU tile).
Measured in round 5 (profiles/r05_small_grids_ring_of_8_ab.txt, where the tree still SHIPPED 188 and 262144 switched it off): slower — not shipped.
Prints one JSON line per arm; the results of the two arms are compared bit for bit (same tile, same MFMA order)."""
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
    cases = (("b1_128f", 1, synthetic.samples_for_frames(128), 30), ("b1_28s", 1, 448000, 8), ("b8_128f", 8, synthetic.samples_for_frames(128), 20),
             ("b64_128f", 64, synthetic.samples_for_frames(128), 20))
    outs = {}
    arms = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [189, 188, 0]      # EMAGE_H2 configuration ids for the lone-block grids; 0 = shipped (120)
    for rep in range(2):
        for small_cfg in arms:
            lib.emage_set_tuning(5, 131072)
            lib.emage_set_tuning(7, small_cfg)
            variant = small_cfg
            model, vq = common.product_models(precision="f16x3", device=dev)
            line = {"small_grid_config": variant or 120, "rep": rep}
            for key, b, n, steps in cases:
                runner = ClipRunner(model, vq, b, n)
                audio = synthetic.synthetic_audio(b, n, seed=1234).to(dev)
                for _ in range(3):
                    res = runner(audio)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    res = runner(audio)
                torch.cuda.synchronize()
                line[key + "_ms"] = round(1e3 * (time.perf_counter() - t0) / steps, 3)
                prev = outs.setdefault(key, [np.array(r) for r in res])
                line[key + "_same_bits"] = all(np.array_equal(a, r) for a, r in zip(prev, res))
                del runner
                torch.cuda.empty_cache()
            print(json.dumps(line), flush=True)
            del model, vq
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

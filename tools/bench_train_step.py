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
sys.argv = ["bench.py"]
import bench  # noqa: E402

print(json.dumps(bench.bench_train_step(torch.device("cuda", 0), cpu=cpu)))

#!/usr/bin/env python
"""Upper bound of what folding LayerNorm into the neighbouring GEMMs could buy: the bench step with every transformer
LayerNorm launch REMOVED (results are wrong by construction; timing only).  Usage: as bench.py (flags passed through)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pantomatrix_amd import modeling_emage_audio as M  # noqa: E402

M.EmageAudioModel._ln = lambda self, cx, key, s, add=None: s

import bench  # noqa: E402

if __name__ == "__main__":
    bench.main()

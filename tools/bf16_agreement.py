#!/usr/bin/env python
"""How close does the production bf16 path get to the fp32 reference results?  (run on the MI355X)
Prints, for the golden 128-frame clips: relative error of the face latent, VQ code agreement per part, and the
rotation error on frames whose codes all agree — the honest form of the parity statement for bf16."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
from pantomatrix_amd import synthetic  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "infer_128f_b2.npz"))
for precision in ("fp32", "bf16"):
    model, vq = common.product_models(precision=precision, device="cuda")
    audio = synthetic.synthetic_audio(2, synthetic.samples_for_frames(128))
    (poses, expr, trans), lat = common.product_infer_clip(model, vq, audio)
    sel = model._select_codes(lat)
    rel = float(np.linalg.norm(lat["rec_face"].cpu().numpy() - g["rec_face"]) / np.linalg.norm(g["rec_face"]))
    agree = {p: float((sel[f"{p}_index"].cpu().numpy() == g[f"index_{p}"]).mean()) for p in ("upper", "hands", "lower")}
    ok = np.ones_like(g["index_upper"], dtype=bool)
    for p in ("upper", "hands", "lower"):
        ok &= sel[f"{p}_index"].cpu().numpy() == g[f"index_{p}"]
    err_all = float(np.abs(poses - g["poses"]).max())
    print(f"{precision}: rec_face rel err {rel:.2e}; code agreement {agree}; frames with all body codes equal {ok.mean():.3f}; "
          f"max |pose err| over all frames {err_all:.3e}; expr max err {float(np.abs(expr - g['expressions']).max()):.3e}")

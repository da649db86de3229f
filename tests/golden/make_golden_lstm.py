"""Generate tests/golden/lstm_models.npz from the REAL reference DisCo / CaMN modules (build container only):
    python tests/golden/make_golden_lstm.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import reference_harness as rh  # noqa: E402
from test_lstm_models_oracle import CFG, inputs, weights  # noqa: E402


def main():
    assert rh.available(), "needs /root/reference"
    out = {}
    for kind in ("disco", "camn"):
        model = rh.build_reference_lstm_model(kind, CFG, weights(kind))
        for tag, wsm in (("plain", False), ("seeded", True)):
            audio, spk, motion = inputs(with_seed_motion=wsm)
            with torch.no_grad():
                ref = model(audio, spk, seed_frames=CFG["seed_frames"], seed_motion=motion)
            out[f"{kind}_{tag}_motion"] = ref["motion"].reshape(2, -1, 258).numpy()
            out[f"{kind}_{tag}_axis_angle"] = ref["motion_axis_angle"].numpy()
    path = os.path.join(HERE, "lstm_models.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})
    # a clip whose WavEncoder frame counts are even after blocks 0-2 (6638 / 1104 / 1104, like the 8.5 s and 28 s clips of the
    # BASELINE configs): the product runs its 32-channel blocks as convolutions over position pairs there
    even = {}
    for kind in ("disco", "camn"):
        model = rh.build_reference_lstm_model(kind, CFG, weights(kind))
        audio, spk, motion = inputs(with_seed_motion=True)
        audio = audio[:, :30000]
        with torch.no_grad():
            ref = model(audio, spk, seed_frames=CFG["seed_frames"], seed_motion=motion)
        even[f"{kind}_motion"] = ref["motion"].reshape(2, -1, 258).numpy()
        even[f"{kind}_axis_angle"] = ref["motion_axis_angle"].numpy()
    np.savez_compressed(os.path.join(HERE, "lstm_models_even.npz"), **even)
    print("wrote lstm_models_even.npz", {k: v.shape for k, v in even.items()})


if __name__ == "__main__":
    main()

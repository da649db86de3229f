"""Generate tests/golden/train_step_b{2,56}.npz from the REAL reference training step (train_emage_audio.py:132-180 run
through oracle/reference_harness.reference_train_step) — build container only (/root/reference).
    python tests/golden/make_golden_train.py [batch size, default 2; 56 = BASELINE configs[2]'s per-GPU batch]
The fixture holds what the oracle (and later the HIP training path) is compared with on machines without the reference:
the seven losses, per-parameter gradient norm and first entry, and the parameter sums after the Adam update."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import common  # noqa: E402
from oracle import reference_harness as rh  # noqa: E402
from test_train_oracle import train_batch  # noqa: E402

ITERATION, SEED = 0, 11


def main(bs=2):
    assert rh.available(), "needs /root/reference"
    acfg, vqc, gc = common.cfg_dicts()
    model, vq = rh.build_reference(acfg, vqc, gc, 0)
    losses, grads, sd_after = rh.reference_train_step(model, vq, acfg, train_batch(bs=bs), ITERATION, SEED)
    names = sorted(grads)
    gmax = max(float(g.abs().max()) for g in grads.values())
    out = {"iteration": ITERATION, "seed": SEED, "bs": bs, "grad_names": np.array(names),
           "grad_norms": np.array([float(grads[n].norm()) for n in names]),
           "grad_first": np.array([float(grads[n].reshape(-1)[0]) for n in names]),
           # conv biases in front of a train-mode BatchNorm: true gradient 0, Adam moves them by +-lr on fp32 noise
           "shadowed": np.array([float(grads[n].abs().max()) < 1e-5 * gmax for n in names]),
           "param_sum_after": np.array([float(sd_after[n].double().sum()) for n in names])}
    for k, v in losses.items():
        out["loss_" + k] = v
    path = os.path.join(HERE, f"train_step_b{bs}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: round(v, 5) for k, v in losses.items()}, "params with grad:", len(names),
          "shadowed:", int(out["shadowed"].sum()))


if __name__ == "__main__":
    # bs 2: the small fixture every training test uses; bs 56: BASELINE configs[2]'s per-GPU batch (56 x 64-frame clips; CPU minutes)
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2)

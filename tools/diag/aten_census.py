"""Which ATen launches does a training step issue OUTSIDE the emage kernels, and from where?  Runs `Trainer.step` on the CPU stand-ins
(tests/fake_ops.py) under a TorchDispatchMode and attributes every non-view ATen call to the innermost pantomatrix_amd frame (calls made
inside a stand-in are the kernel itself and are skipped).  No GPU needed: the host logic is the same as on the device.
usage: python tools/diag/aten_census.py [f16x3|fp32] [top] [cuda]"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from torch.utils._python_dispatch import TorchDispatchMode

VIEWS = {"view", "slice", "detach", "t", "permute", "expand", "select", "as_strided", "unsqueeze", "squeeze", "_unsafe_view", "alias", "empty",
         "empty_like", "empty_strided", "reshape", "transpose", "unbind", "split", "_reshape_alias", "narrow", "unfold", "lift_fresh", "item",
         "_local_scalar_dense", "is_same_size", "sym_size", "sym_stride", "stride", "size", "is_contiguous", "new_empty", "view_as", "chunk",
         "split_with_sizes", "resize_", "set_", "_to_copy_noop", "scalar_tensor"}


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()
        self.ops = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in VIEWS:
            return out
        if name == "_to_copy":
            a = args[0]
            if kwargs and kwargs.get("dtype", a.dtype) == a.dtype and kwargs.get("device", a.device) == a.device:
                pass
        site = None
        stack = traceback.extract_stack(limit=60)
        if any(fr.filename.endswith("fake_ops.py") for fr in stack):
            return out                          # inside a kernel stand-in
        for fr in reversed(stack):
            fn = fr.filename
            if "/pantomatrix_amd/" in fn or fn.endswith("workloads.py"):
                site = f"{os.path.basename(fn)}:{fr.lineno} {fr.name}"
                break
        if site is None:
            return out
        numel = max([a.numel() for a in args if torch.is_tensor(a)] + [0])
        self.sites[(site, name)] += 1
        self.ops[name] += 1
        return out


def main():
    precision = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    on_gpu = len(sys.argv) > 3 and sys.argv[3] == "cuda"          # the real operators on the device: also sees what the op wrappers allocate / fill
    import common
    import fake_ops
    import train_common as tc
    import numpy as np
    from pantomatrix_amd import training
    g = np.load(os.path.join(ROOT, "tests", "golden", "train_step_b2.npz"))
    batch, _, masks, random_mask, _ = tc.oracle_step(int(g["seed"]), int(g["iteration"]))
    import contextlib
    if on_gpu:
        dev = torch.device("cuda", 0)
        batch = {k: v.to(dev) for k, v in batch.items()}
        masks, random_mask = None, random_mask.to(dev)
    model, vq = common.product_models(precision=precision, **({"device": torch.device("cuda", 0)} if on_gpu else {}))
    trainer = training.Trainer(model, vq)
    with (contextlib.nullcontext() if on_gpu else fake_ops.installed()), torch.no_grad():
        trainer.step(batch, 0, masks, random_mask)          # first step: lazy initialisation, schedule learning
        c = Census()
        with c:
            trainer.step(batch, 1, masks, random_mask)
    total = sum(c.ops.values())
    print(f"{total} non-view ATen calls in one step ({precision})")
    for name, n in c.ops.most_common(25):
        print(f"  {n:6d}  {name}")
    print("by call site:")
    for (site, name), n in c.sites.most_common(top):
        print(f"  {n:6d}  {name:28s} {site}")


if __name__ == "__main__":
    main()

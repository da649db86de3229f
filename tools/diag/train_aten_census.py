"""Which ATen kernels does one EMAGE training step (training.Trainer.step, f16x3) issue besides the emage ops, and from where?  On the MI355X: the real
step (2 clips; the launch COUNT does not depend on the batch).  Without a GPU: the CPU stand-ins (tests/fake_ops.py) — the host code's own ATen calls
only (the `ops` wrappers are replaced there, and what the stand-ins do is left out).
    python tools/diag/train_aten_census.py [--second]      # --second: count the SECOND step (packing caches warm)"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import common, fake_ops, train_common as tc
from pantomatrix_amd import training
VIEWS = {"view", "slice", "detach", "t", "permute", "expand", "select", "as_strided", "unsqueeze", "squeeze", "_unsafe_view", "alias", "empty", "empty_like",
         "empty_strided", "reshape", "transpose", "unbind", "split", "_reshape_alias", "narrow", "unfold", "lift_fresh", "new_empty", "view_as", "chunk", "split_with_sizes",
         "resize_", "set_", "record_stream", "_to_copy_noop", "is_same_size", "sym_size", "stride", "size", "numel", "_local_scalar_dense", "item"}
import contextlib
gpu = torch.cuda.is_available()
batch, _ref, masks, random_mask, _ = tc.oracle_step(3, 0)
model, vq = common.product_models(precision="f16x3", **({"device": "cuda"} if gpu else {}))
if gpu:
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
    masks, random_mask = None, random_mask.cuda()
trainer = training.Trainer(model, vq)
sites = collections.Counter()
class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in VIEWS or "emage" in str(func):
            return out
        site = None
        for fr in reversed(traceback.extract_stack(limit=60)):
            if fr.filename.endswith("fake_ops.py"):
                return out                                  # issued by a stand-in, not by the host code
            if "/pantomatrix_amd/" in fr.filename:
                site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                break
        numel = max([a.numel() for a in list(args) + ([out] if torch.is_tensor(out) else []) if torch.is_tensor(a)] + [0])
        sites[(site, name)] += 1
        return out
with (contextlib.nullcontext() if gpu else fake_ops.installed()), torch.no_grad():
    if "--second" in sys.argv:
        trainer.step(batch, 0, masks, random_mask)
    with M():
        trainer.step(batch, 1, masks, random_mask)
tot = collections.Counter()
for (site, name), k in sites.items():
    tot[name] += k
print("per op:", dict(tot.most_common(25)))
for (site, name), k in sorted(sites.items(), key=lambda kv: -kv[1])[:70]:
    print(f"x{k:4d} {name:24s} {site}")

"""Which ATen kernels does one EMAGE inference step (runtime.ClipRunner._step, eager) launch besides the emage ops, and from where?  (run on the MI355X)"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from tools import workloads as common
from pantomatrix_amd import synthetic
from pantomatrix_amd.runtime import ClipRunner
VIEWS = {"view", "slice", "detach", "t", "permute", "expand", "select", "as_strided", "unsqueeze", "squeeze", "_unsafe_view", "alias", "empty", "empty_like",
         "empty_strided", "reshape", "transpose", "unbind", "split", "_reshape_alias", "narrow", "unfold", "lift_fresh", "new_empty", "view_as", "chunk", "split_with_sizes",
         "resize_", "set_", "record_stream"}
dev = torch.device("cuda:0")
model, vq = common.product_models(precision="f16x3", device=dev)
n = synthetic.samples_for_frames(128)
r = ClipRunner(model, vq, 64, n, use_graph=False)
sites = collections.Counter()
class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        ns, name = func.__module__ if hasattr(func, "__module__") else "", func.__name__.split(".")[0]
        if name in VIEWS or "emage" in str(func):
            return out
        site = None
        for fr in reversed(traceback.extract_stack(limit=40)):
            if "/pantomatrix_amd/" in fr.filename:
                site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                break
        numel = max([a.numel() for a in list(args) + [out] if torch.is_tensor(a)] + [0])
        sites[(site, name, numel)] += 1
        return out
with torch.no_grad(), M():
    r._step()
for (site, name, numel), k in sorted(sites.items(), key=lambda kv: -kv[1] * max(kv[0][2], 1))[:40]:
    print(f"x{k:3d} {name:22s} numel {numel:10d}  {site}")

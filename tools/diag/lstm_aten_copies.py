import sys, os, traceback, collections, torch
sys.path.insert(0, "/root/repo")
from tools.workloads import lstm_product
from pantomatrix_amd import synthetic
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "camn"
batch, seconds = (256, 28.0) if kind == "camn" else (128, 8.5)
model = lstm_product(kind, "f16x3", dev)
audio = synthetic.synthetic_audio(batch, int(seconds * 16000), seed=5).to(dev)
spk = torch.zeros(batch, 1, dtype=torch.long, device=dev)
with torch.no_grad():
    model(audio, spk)
sites = collections.Counter(); byts = collections.Counter()
class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in ("copy_", "clone", "contiguous", "_to_copy", "cat", "index", "index_select", "zeros", "zero_", "fill_", "add", "mul", "where", "stack"):
            st = traceback.extract_stack(limit=40)
            site = None
            for fr in reversed(st):
                if "/pantomatrix_amd/" in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"; break
            n = max([a.numel() * a.element_size() for a in list(args) + [out] if torch.is_tensor(a)] + [0])
            sites[(site, name)] += 1; byts[(site, name)] += n
        return out
with torch.no_grad(), M():
    model(audio, spk)
for k, v in sorted(byts.items(), key=lambda kv: -kv[1])[:25]:
    print(f"{v / 1e6:10.1f} MB  x{sites[k]:3d}  {k[1]:12s} {k[0]}")

"""Where do the small ATen kernels of one eager EMAGE training step come from?  torch.profiler with stacks on the MI355X: ATen ops that launched device kernels,
grouped by (op, nearest pantomatrix_amd frame), with launch counts and device time.
    python tools/diag/train_kernel_origins.py"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
from torch.profiler import profile, ProfilerActivity
import common, train_common as tc
from pantomatrix_amd import training
batch, _ref, _masks, random_mask, _ = tc.oracle_step(3, 0)
model, vq = common.product_models(precision="f16x3", device="cuda")
batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
random_mask = random_mask.cuda()
trainer = training.Trainer(model, vq)
with torch.no_grad():
    trainer.step(batch, 0, None, random_mask)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        trainer.step(batch, 1, None, random_mask)
        torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or not ev.kernels:
        continue
    site = next((s for s in ev.stack if "/pantomatrix_amd/" in s), "?")
    site = site.split("/pantomatrix_amd/")[-1]
    k = agg[(ev.name, site)]
    k[0] += len(ev.kernels)
    k[1] += sum(kk.duration for kk in ev.kernels)
tot = collections.Counter()
for (name, site), (n, us) in agg.items():
    tot[name] += n
print("device kernels per ATen op:", dict(tot.most_common(20)))
for (name, site), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"x{n:4d} {us:9.1f} us  {name:28s} {site}")

"""Peak memory and step time of the captured 56-clip training step against `TrainForward.conv_backward_rows` (output rows per piece of a long convolution's backward).
    python tools/diag/train_piece_rows.py 131072 65536 32768"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tools import workloads as common
from pantomatrix_amd import training
dev = torch.device("cuda", 0)
out = {}
for rows in [int(a) for a in sys.argv[1:]] or [1 << 17]:
    model, vq = common.product_models(precision="f16x3", device=dev)
    data = {k: v.to(dev) for k, v in common.train_batch(bs=56, t=64).items()}
    random_mask = (torch.rand(56, 64, 337, generator=torch.Generator().manual_seed(6)) < 0.5).float().to(dev)
    torch.cuda.reset_peak_memory_stats()
    trainer = training.Trainer(model, vq, seed=1)
    trainer.fwd.conv_backward_rows = rows
    trainer.capture(data, random_mask)
    trainer.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        losses = trainer.replay()
    torch.cuda.synchronize()
    out[rows] = {"ms_per_step": 1e3 * (time.perf_counter() - t0) / 3, "peak_memory_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "loss": losses["all"]}
    del trainer, model, vq, data
    torch.cuda.empty_cache()
print(json.dumps(out))

"""A/B of the in-launch split-K (runtime.ClipRunner split_k, ops.SplitKScratch) for ONE clip and small batches (run on the MI355X)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from tools import workloads as common
from pantomatrix_amd import synthetic
from pantomatrix_amd.runtime import ClipRunner
dev = torch.device("cuda:0")
model, vq = common.product_models(precision="f16x3", device=dev)
for b, nsamp, tag in ((1, synthetic.samples_for_frames(128), "1 x 128f"), (1, 448000, "1 x 28s"), (4, synthetic.samples_for_frames(128), "4 x 128f"), (8, synthetic.samples_for_frames(128), "8 x 128f")):
    audio = synthetic.synthetic_audio(b, nsamp, seed=1234).to(dev)
    for sk in (True, False, True, False):
        r = ClipRunner(model, vq, b, nsamp, split_k=sk)
        for _ in range(3): r(audio)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): r(audio)
        torch.cuda.synchronize()
        print(f"{tag:10s} split_k={sk!s:5s} {(time.perf_counter() - t0) * 100:.3f} ms", flush=True)
        del r

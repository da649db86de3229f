#!/usr/bin/env python
"""bench.py — EMAGE inference throughput on MI355X (BASELINE.json metric: motion-frames/s, 128-frame clips).

One "step" = one pass of the hot path over one batch of 64 synthetic 128-frame clips per GPU:
`EmageAudioModel.inference` (2 dependent 64-frame windows) + the final `EmageVQModel.decode(get_global_motion=
True)` + D2H of poses/expressions/trans, audio already resident in HBM.  120 frames are emitted per clip.
N > 1: one process per GPU (torchrun), clips sharded across ranks, no data-path collective ("replicas only",
weak scaling); timing is barrier + synchronize bracketed, max over ranks.

Prints ONE JSON line on rank 0 with the contract fields plus `roofline` (dominant kernel, HIP-event timed)
and `cpu_baseline` (the CPU oracle, a port of the reference, timed on this host on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

BF16_MFMA_PEAK_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md
F32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0


def usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def build_models(precision, device):
    import common
    return common.product_models(precision=precision, device=device)


def one_step(runner, audio):
    """The timed unit: audio (already in HBM) -> runner (hipGraph replay of inference + final decode) -> poses /
    expressions / trans on the host."""
    return runner(audio)


def profile_kernels(model, vq, audio, spk, zeros_trans):
    def eager_step():
        lat = model.inference(audio, spk, vq)
        pred = vq.decode(**model._select_codes(lat), get_global_motion=True, ref_trans=zeros_trans)
        return pred["motion_axis_angle"].cpu()

    """One extra, untimed step with a HIP-event pair around every kernel launch (same stream the kernels run
    on).  Returns {family: [count, total_ms, algorithmic flops, algorithmic bytes]}."""
    from pantomatrix_amd import ops
    from pantomatrix_amd._lib import BF16
    records = []
    saved = {}

    def wrap(name, fn, cost):
        def inner(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            records.append((name, e0, e1) + cost(a, k, r))
            return r
        return inner

    def gemm_cost(a, k, r):
        dtype, A, W = a[0], a[1], a[2]
        n, cp, taps = k["n"], k["cp"], k.get("taps", 1)
        m = k.get("m") or A.shape[0]
        es = 2 if dtype == BF16 else 4
        kk = k.get("k_real") or taps * cp                    # unpadded contraction length
        flops = 2.0 * m * n * kk
        byts = (m * cp * es) + n * taps * cp * es + m * n * es
        return ("gemm_bf16" if dtype == BF16 else "gemm_f32", flops, byts)

    def attn_cost(a, k, r):
        dtype, b, h, tq, tk, hd = a[0], a[6], a[7], a[8], a[9], a[10]
        es = 2 if dtype == BF16 else 4
        return ("attention", 4.0 * b * h * tq * tk * hd, (2 * b * tq + 2 * b * tk) * h * hd * es)

    def generic_cost(tag):
        def c(a, k, r):
            byts = 0
            for t in list(a) + list(k.values()) + (list(r) if isinstance(r, tuple) else [r]):
                if torch.is_tensor(t):
                    byts += t.numel() * t.element_size()
            return (tag, 0.0, float(byts))
        return c

    def layer_cost(a, k, r):
        x, b, t = a[1], a[5], a[6]
        d, ffn, cross = x.shape[1], k["ffn"], k.get("mem_k") is not None
        m = b * t
        flops = 2.0 * m * d * (3 * d + d + 2 * ffn + (2 * d if cross else 0)) + 4.0 * m * k.get("tk", t) * d * (2 if cross else 1)
        byts = 2.0 * d * (3 * d + d + 2 * ffn + (2 * d if cross else 0)) + 2.0 * m * d * 2
        return ("transformer_layer", flops, byts)

    table = {"gemm": gemm_cost, "attention": attn_cost, "transformer_layer": layer_cost}
    for nm in ("layernorm", "add", "pack_motion", "cast_pad", "gather_rows", "vq_argmin", "argmax_logsoftmax",
               "wav_conv_in", "merge_parts", "velocity_to_position"):
        table[nm] = generic_cost(nm)
    try:
        for nm, cost in table.items():
            saved[nm] = getattr(ops, nm)
            setattr(ops, nm, wrap(nm, saved[nm], cost))
        eager_step()
        torch.cuda.synchronize()
    finally:
        for nm, fn in saved.items():
            setattr(ops, nm, fn)
    fam = {}
    for name, e0, e1, tag, flops, byts in records:
        f = fam.setdefault(tag, [0, 0.0, 0.0, 0.0])
        f[0] += 1
        f[1] += e0.elapsed_time(e1)
        f[2] += flops
        f[3] += byts
    return fam


def cpu_baseline(frames, seconds_budget=15.0):
    """The CPU oracle (a port of the reference path, oracle/emage_oracle.py) on a bounded sample of the same
    workload: `bs` 128-frame clips per call, repeated until ~seconds_budget of CPU work."""
    import common
    from oracle import emage_oracle as orc
    from pantomatrix_amd import synthetic
    # use the cores this container may actually run on (affinity AND cgroup CPU quota): forcing os.cpu_count()
    # threads on a quota-limited box measures the scheduler, not the code
    torch.set_num_threads(usable_cores())
    omodel, ovq = common.oracle_models()
    t0 = time.time()
    orc.infer_clip(omodel, ovq, synthetic.synthetic_audio(1, synthetic.samples_for_frames(frames)))   # warm-up
    t_one = time.time() - t0
    log(f"cpu_baseline: warm-up clip took {t_one:.2f}s with {torch.get_num_threads()} threads")
    bs = 8 if t_one < 1.0 else (2 if t_one < 4.0 else 1)
    audio = synthetic.synthetic_audio(bs, synthetic.samples_for_frames(frames))
    times, out_frames, t_start = [], 0, time.time()
    while True:
        t0 = time.time()
        poses, _, _ = orc.infer_clip(omodel, ovq, audio)
        times.append(time.time() - t0)
        out_frames = poses.shape[0] * poses.shape[1]
        if time.time() - t_start > seconds_budget or len(times) >= 200:
            break
    med = float(np.median(times))
    return {"value": out_frames / med, "unit": "motion-frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{bs} clips x {frames} frames per call, median of {len(times)} calls ({sum(times):.1f} s CPU), fp32 torch CPU oracle"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU per step (BASELINE config 2: 64)")
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "bf16", "fp32"],
                    help="f16x3 (default): fp32 storage, split-f16 MFMA — the parity-green fast mode; bf16: bf16 operands "
                         "(not index-exact); fp32: exact-fp32 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--sub-batches", type=int, default=1, help="independent clip groups issued on parallel stream lanes")
    ap.add_argument("--no-hoist", action="store_true", help="A/B: compute waveform features inside every window")
    ap.add_argument("--no-concurrent", action="store_true", help="A/B: single stream, no fork/join lanes")
    ap.add_argument("--fused-layers", action="store_true", help="A/B: one launch per transformer layer (emage_transformer_layer)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying the captured hipGraph")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from pantomatrix_amd import dist as pdist
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    from pantomatrix_amd import synthetic
    model, vq = build_models(args.precision, dev)
    model.hoist_audio = not args.no_hoist
    model.concurrent = not args.no_concurrent
    model.fused_layers = args.fused_layers
    for part in (vq.vq_model_face, vq.vq_model_upper, vq.vq_model_hands, vq.vq_model_lower, vq.global_motion):
        part.concurrent = not args.no_concurrent
    n_samples = synthetic.samples_for_frames(args.frames)
    # clip i of the global batch lives on rank i % world; every rank gets `batch` clips
    audio = synthetic.synthetic_audio(args.batch, n_samples, seed=1234 + rank).to(dev)
    spk = torch.zeros(args.batch, 1, dtype=torch.long, device=dev)
    zeros_trans = torch.zeros(1, 3, device=dev)

    def barrier():
        pdist.barrier()
        torch.cuda.synchronize()

    from pantomatrix_amd.runtime import ClipRunner
    log(f"models built on {dev}; capturing the clip graph")
    runner = ClipRunner(model, vq, args.batch, n_samples, use_graph=not args.no_graph, sub_batches=args.sub_batches)
    # the process group (RCCL; barriers and the max-over-ranks of the timing only) is joined AFTER the graph capture, so
    # no communicator thread is alive while the stream capture is open; None without a launcher
    pdist.init("nccl", dev)
    log(f"warm-up x{args.warmup}")
    for _ in range(args.warmup):
        poses, _, _ = one_step(runner, audio)
    log("timed region")
    frames_per_step = poses.shape[0] * poses.shape[1] if args.warmup else None
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        poses, expr, trans = one_step(runner, audio)
    barrier()
    elapsed = time.perf_counter() - t0
    frames_per_step = poses.shape[0] * poses.shape[1]
    elapsed = pdist.max_over_ranks(elapsed, dev)
    assert np.isfinite(poses).all()
    log(f"timed: {1e3 * elapsed / args.steps:.2f} ms/step")

    result = {
        "metric": "motion-frames/sec (30fps SMPL-X) EMAGE infer, 128-frame clips",
        "value": frames_per_step * world * args.steps / elapsed,
        "unit": "motion-frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"EMAGE inference {args.precision}, batch={args.batch}x{args.frames}-frame synthetic clips per GPU "
                               f"(BASELINE configs[1]): 2 windows of 64 frames + final VQ decode with global motion, "
                               f"{frames_per_step // args.batch} frames out per clip; synthetic seeded weights",
                   "clips_per_gpu": args.batch, "frames_in": args.frames, "frames_out_per_clip": frames_per_step // args.batch,
                   "parallelism": f"replicas x{world} (clip-sharded, no collective)",
                   "launch": "eager" if args.no_graph else "hipGraph replay", "sub_batches": args.sub_batches, "batches_in_flight": 1},
    }
    if rank == 0 and not args.no_roofline:
        fam = profile_kernels(model, vq, audio, spk, zeros_trans)
        total_ms = sum(v[1] for v in fam.values())
        dom = max(fam.items(), key=lambda kv: kv[1][1])
        name, (cnt, ms, flops, byts) = dom
        if name.startswith("gemm") or name in ("attention", "transformer_layer"):
            peak = BF16_MFMA_PEAK_TFLOPS if args.precision == "bf16" else F32_MFMA_PEAK_TFLOPS
            ach = flops / (ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak}
        else:
            ach = byts / (ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):
            with open(tfile) as f:
                traffic = json.load(f).get(name)
        roof.update({"traffic": traffic, "kernel": name, "launches_per_step": cnt, "avg_launch_us": 1e3 * ms / cnt,
                     "algorithmic_gflop_per_launch": flops / cnt / 1e9,
                     "share_of_kernel_time": ms / total_ms if total_ms else None,
                     "kernel_time_ms_by_family": {k: round(v[1], 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])},
                     "launches_by_family": {k: v[0] for k, v in fam.items()}})
        result["roofline"] = roof
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.frames)
    if rank == 0:
        print(json.dumps(result))
    pdist.finalize()


if __name__ == "__main__":
    main()

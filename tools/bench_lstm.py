#!/usr/bin/env python
"""Throughput of the DisCo / CaMN inference path at the BASELINE configs[3] / [4] sizes (run on the MI355X):
  DisCo  batch 128 x 8.5 s clips (128 frames at 15 fps);  CaMN  batch 256 x 28 s clips (~420 frames).
One step = one hipGraph replay of the whole forward (runtime.LstmClipRunner) + D2H of the motion; prints one JSON line per
model with motion-frames/s, the share of the recurrence, and the CPU oracle timed on a 2-clip sample beside it."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--models", default="disco,camn")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--per-step", action="store_true", help="one launch per time step instead of the persistent recurrence (A/B; same bits)")
    ap.add_argument("--no-pair-convs", action="store_true", help="WavEncoder: zero-padded 64-channel route for the 32-channel blocks (A/B)")
    ap.add_argument("--h2-proj", action="store_true", help="A/B (measured: no gain, off in the product): the per-layer input projections on pre-split EMAGE_H2 operands instead of float32 activations split in the GEMM")
    ap.add_argument("--layer-only", action="store_true", help="also time one bare lstm_layer launch at the model's size")
    args = ap.parse_args()
    from pantomatrix_amd import synthetic
    from pantomatrix_amd.runtime import LstmClipRunner
    from test_lstm_host_logic import product
    from test_lstm_models_oracle import weights, run_oracle
    dev = "cuda"
    for kind, batch, seconds in (("disco", 128, 8.5), ("camn", 256, 28.0)):
        if kind not in args.models.split(","):
            continue
        n = int(seconds * 16000)
        model = product(kind, "f16x3", dev)
        model.persistent_lstm = not args.per_step
        model.pair_convs = not args.no_pair_convs
        model.h2_input_projection = bool(args.h2_proj)
        audio = synthetic.synthetic_audio(batch, n, seed=5).to(dev)
        t0 = time.time()
        runner = LstmClipRunner(model, batch, n)
        t_capture = time.time() - t0
        runner(audio)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            motion, aa = runner(audio)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.steps
        frames = motion.shape[0] * motion.shape[1]
        line = {"model": kind, "batch": batch, "frames_per_clip": int(motion.shape[1]), "ms_per_step": ms, "value": frames / (ms * 1e-3),
                "unit": "motion-frames/s (15 fps)", "dtype": "f16x3", "launch": "hipGraph replay", "graph_capture_s": t_capture,
                "recurrence": "one launch per step" if args.per_step else "persistent (one launch per layer)",
                "input_projections": "pre-split EMAGE_H2 operands" if args.h2_proj else "float32 activations, split in the GEMM",
                "wav_encoder_narrow_blocks": "zero-padded to 64 channels" if args.no_pair_convs else "position pairs",
                "lstm_layers": (1 if kind == "disco" else 2) * 4, "steps_per_layer": int(motion.shape[1])}
        if args.layer_only and not args.per_step:
            from pantomatrix_amd import ops
            from pantomatrix_amd._lib import F16X3
            hid, tt = 512, int(motion.shape[1])
            g = torch.Generator().manual_seed(1)
            wp, ws = [], []
            for _ in range(2):
                p_, s_ = ops.split_f16_weights(torch.randn(4 * hid, hid, generator=g) / hid ** 0.5)
                wp.append(p_.to(dev))
                ws.append(s_)
            gx = torch.randn(batch, tt, 8 * hid, generator=g).to(dev)
            hseq = torch.empty(batch, tt, 2 * hid, device=dev)
            sync = ops.lstm_layer_sync(batch, hid, dev)
            from pantomatrix_amd import _lib
            lib = _lib.use_tools(True)      # tools build of the library: every tile configuration + emage_set_tuning
            names = {0: "shipped (data-as-flag hand-over, two pipelined phases)", 32: "round 2's arrival-counter protocol", 2: "no MFMA phase", 4: "no h load / staging",
                     8: "no wait for the group (one read)", 14: "skeleton: cell + h store only", 64: "shipped + s_sleep 1 between re-reads",
                     128: "round 3: one hand-over phase", 256: "four pipelined hand-over phases", 512: "64 clips per block for every batch (round 4's geometry)",
                     1024: "two 32-clip blocks per CU for batches up to 256 clips (measured slower: tools only)"}
            line["lstm_layer_us_per_step"] = {}
            line["same_bits_as_shipped"] = {}
            base = None
            for dbg in (0, 1024, 512, 0, 1024, 512, 2, 4, 8, 14):
                lib.emage_set_tuning(3, dbg)
                ops.lstm_layer(F16X3, gx, wp, ws, hseq, sync)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    ops.lstm_layer(F16X3, gx, wp, ws, hseq, sync)
                e1.record()
                torch.cuda.synchronize()
                ops.lstm_layer_check(sync)
                if dbg == 0 and base is None:
                    base = hseq.clone()
                elif dbg in (128, 256, 512, 1024):
                    line["same_bits_as_shipped"][names[dbg]] = bool(torch.equal(hseq, base))
                key = names[dbg] if names[dbg] not in line["lstm_layer_us_per_step"] else names[dbg] + " (second run)"
                line["lstm_layer_us_per_step"][key] = round(1e3 * e0.elapsed_time(e1) / 3 / tt, 2)
            lib.emage_set_tuning(3, 0)
        if not args.no_cpu:
            torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
            sd = weights(kind)
            a2 = audio[:2].cpu()
            spk = torch.zeros(2, 1, dtype=torch.long)
            run_oracle(kind, sd, a2[:, :n // 4], spk, None)
            t0 = time.time()
            ref = run_oracle(kind, sd, a2, spk, None)
            dt = time.time() - t0
            line["cpu_baseline"] = {"value": 2 * ref["motion"].shape[1] / dt, "kind": "port", "cores": torch.get_num_threads(), "sample": "2 clips, one call"}
            line["max_err_vs_oracle_clip0_1"] = float((torch.from_numpy(motion[:2]) - ref["motion"].reshape(2, -1, 258)).abs().max())
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

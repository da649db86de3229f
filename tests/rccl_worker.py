"""Worker of tests/test_rccl_gpu.py (its own process: a wedged collective must not take the GPU suite with it).

Joins a WORLD_SIZE = 1 process group on the `nccl` backend (= RCCL on ROCm) on the one MI355X of the box, so that every collective of the
training exchange — the four bucket all-reduces, SyncBatchNorm's all-gathers and small all-reduces (train_emage_audio.py:214, 248-251) —
goes through RCCL on the device, and runs

  1. `Trainer(sync_bn=True, exchange=True).step` (eager) against the REAL reference step's golden (tests/golden/train_step_b2.npz),
     counting the collectives the backend was asked for;
  2. `Trainer.capture` + `replay` of the same step WITH its collectives inside the hipGraph, against the same golden.

Prints one JSON line: {"eager": {...}, "captured": {...}}; a stage that raised carries {"error": "<type>: <text>"} instead."""
import json
import os
import socket
import sys
import time
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def main():
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as tdist

    import common
    import train_common as tc
    import test_train_forward_gpu as ttf
    from pantomatrix_amd import dist as pd
    from pantomatrix_amd import training

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    res = {"torch": torch.__version__, "hip": torch.version.hip}
    assert pd.init("nccl", device=dev) is not None
    res["backend"], res["world"] = str(tdist.get_backend()), tdist.get_world_size()
    try:
        res["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as e:  # noqa: BLE001
        res["rccl_version"] = f"unknown ({e})"

    g = np.load(os.path.join(HERE, "golden", "train_step_b2.npz"))
    batch, _, masks, random_mask, _ = tc.oracle_step(int(g["seed"]), int(g["iteration"]))
    batch = {k: v.to(dev) for k, v in batch.items()}
    masks = [[m.to(dev).contiguous() for m in fm] for fm in masks]
    random_mask = random_mask.to(dev)

    # ---- 1. the eager step through RCCL ----
    try:
        model, vq = common.product_models(precision="f16x3", device=dev)
        before = {k: v.clone() for k, v in model._flat_params().items()}
        trainer = training.Trainer(model, vq, sync_bn=True, exchange=True)
        seen = {}
        with pd.CollectiveCounter() as cc:
            t0 = time.time()
            losses = trainer.step(batch, int(g["iteration"]), masks, random_mask, grad_hook=lambda gr: seen.update({k: float(v.norm()) for k, v in gr.items()}))
            torch.cuda.synchronize()
            dt = time.time() - t0
        calls = cc.counts
        worst = ttf._check_step_against(g, losses, model._flat_params(), before, grads=seen, what="eager step over RCCL")
        res["eager"] = dict(ok=True, collectives=dict(calls), worst=worst, loss_all=losses["all"], first_step_s=round(dt, 2),
                            exchange_log=[e[0] for e in trainer.exchange_log])
    except Exception as e:  # noqa: BLE001
        res["eager"] = dict(error=f"{type(e).__name__}: {e}", trace=traceback.format_exc()[-1500:])

    # ---- 2. the step captured WITH its collectives ----
    try:
        model, vq = common.product_models(precision="f16x3", device=dev)
        before = {k: v.clone() for k, v in model._flat_params().items()}
        trainer = training.Trainer(model, vq, sync_bn=True, exchange=True)
        trainer.capture(batch, random_mask, masks)
        for k, v in before.items():
            assert torch.equal(model._flat_params()[k], v), k
        losses = trainer.replay()
        worst = ttf._check_step_against(g, losses, model._flat_params(), before, what="captured step with RCCL collectives inside the graph")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            trainer._graph.replay()
        e1.record()
        torch.cuda.synchronize()
        res["captured"] = dict(ok=True, worst=worst, loss_all=losses["all"], steps_done=trainer.steps_done, replay_ms=round(e0.elapsed_time(e1) / 3, 2))
    except Exception as e:  # noqa: BLE001
        res["captured"] = dict(error=f"{type(e).__name__}: {e}", trace=traceback.format_exc()[-1500:])
    # ---- 3. hand-over: captured steps, a RAGGED last batch through the eager step, the graph again ----
    try:
        ragged = {k: v[:1].contiguous() for k, v in batch.items()}
        ragged_mask = random_mask[:1].contiguous()
        model, vq = common.product_models(precision="f16x3", device=dev)
        trainer = training.Trainer(model, vq, sync_bn=True, exchange=True, seed=3)
        trainer.capture(batch, random_mask)                      # dropout masks drawn on the device from (seed, step)
        la = [trainer.replay()["all"], trainer.step(ragged, random_mask=ragged_mask)["all"], trainer.replay()["all"]]
        counter = int(trainer._step_counter)
        model2, vq2 = common.product_models(precision="f16x3", device=dev)
        twin = training.Trainer(model2, vq2, sync_bn=True, exchange=True, seed=3)
        lb = [twin.step(batch, random_mask=random_mask)["all"], twin.step(ragged, random_mask=ragged_mask)["all"], twin.step(batch, random_mask=random_mask)["all"]]
        pa, pb = model._flat_params(), model2._flat_params()
        worst, equal = 0.0, True
        for k, v in pa.items():
            if not v.dtype.is_floating_point:
                continue
            d = float((v - pb[k]).abs().max())
            worst = max(worst, d / (float(pb[k].abs().max()) + 1e-12))
            equal = equal and bool(torch.equal(v, pb[k]))
        res["handover"] = dict(ok=True, losses_graph_eager_graph=la, losses_twin_eager=lb, steps_done=[trainer.steps_done, twin.steps_done],
                               device_step_counter=counter, worst_rel_param_diff=worst, bit_equal=equal)
    except Exception as e:  # noqa: BLE001
        res["handover"] = dict(error=f"{type(e).__name__}: {e}", trace=traceback.format_exc()[-1500:])
    print("RCCL_WORKER " + json.dumps(res, default=float), flush=True)
    try:
        tdist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass


if __name__ == "__main__":
    main()

"""The training exchange ON HARDWARE: a world_size = 1 `nccl` (= RCCL) process group on the one MI355X of
the GPU box carries every collective of a step — 4 bucket all-reduces, 12 SyncBatchNorm all-gathers, 12 + 1 small all-reduces
(train_emage_audio.py:214, 248-251) — eagerly and INSIDE the captured hipGraph, and the step still reproduces the REAL reference's
golden.  Runs in a worker process (tests/rccl_worker.py) under a timeout: a wedged collective fails this test, not the suite."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def rccl_result():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_worker.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("RCCL_WORKER ")]
    assert lines, f"the RCCL worker printed no result (exit code {p.returncode}):\n{p.stdout[-3000:]}"
    res = json.loads(lines[-1][len("RCCL_WORKER "):])
    print("RCCL worker:", json.dumps({k: v for k, v in res.items() if k not in ("eager", "captured")}))
    return res


@pytest.mark.gpu
def test_eager_step_exchanges_through_rccl_and_matches_the_reference(rccl_result):
    e = rccl_result["eager"]
    assert rccl_result["backend"] == "nccl" and rccl_result["world"] == 1
    assert "error" not in e, e
    # the collectives of ONE step with the shared encoder pass (the counts tests/test_dist_cpu.py pins on two gloo ranks)
    c = e["collectives"]
    assert (c["all_gather"], c["all_reduce_small"], c["all_reduce_bucket"]) == (12, 13, 4), c
    assert c["bucket_bytes"] > 500_000_000, c                    # the fp32 gradients of every parameter the forward uses (508 MB) went through RCCL
    assert e["exchange_log"].count("reduce") == 4 and e["exchange_log"][-1] == "wait"
    print("eager step over RCCL:", json.dumps(e))


@pytest.mark.gpu
def test_captured_step_holds_its_collectives_and_matches_the_reference(rccl_result):
    c = rccl_result["captured"]
    assert "error" not in c, c
    assert c["steps_done"] == 1
    print("captured step with RCCL collectives:", json.dumps(c))


@pytest.mark.gpu
def test_ragged_last_batch_hands_over_from_the_graph_to_the_eager_step_and_back(rccl_result):
    """A captured multi-rank step replays the batch size it captured (SyncBatchNorm's clip total is a launch-time constant); the ragged LAST
    batch of an epoch therefore takes `Trainer.step` — which exchanges the count itself — and the graph continues behind it: replay, eager
    step on ONE clip, replay equals three eager steps of a twin trainer (losses, parameters, step counts on host and device)."""
    h = rccl_result["handover"]
    assert "error" not in h, h
    assert h["steps_done"] == [3, 3] and h["device_step_counter"] == 3
    for a, b in zip(h["losses_graph_eager_graph"], h["losses_twin_eager"]):
        assert abs(a - b) <= 1e-6 * max(1.0, abs(b)), h
    assert h["worst_rel_param_diff"] <= 1e-6, h
    print("hand-over graph -> eager -> graph:", json.dumps(h))

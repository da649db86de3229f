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

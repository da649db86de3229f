"""bench.py's multi-GPU entry on CPU: `--gpus N` without a launcher re-executes itself under torch.distributed.run (one
process per GPU, 127.0.0.1 rendezvous), and the timed region + whole-job aggregation run here on two gloo ranks with a
stand-in step (the real step needs an MI355X; the launch / barrier / reduce plumbing does not)."""
import os
import socket
import subprocess
import sys
import time

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_spawn_command_is_the_drivers_recipe():
    import bench
    cmd = bench.spawn_command(["--gpus", "4", "--steps", "7"], 4, 29517)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    assert cmd[-5].endswith("bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "7"]


def test_gpus_flag_spawns_ranks_and_fails_loudly_without_gpus():
    """On this CPU box the spawned ranks must refuse to run (no CPU path) and the parent must report the failure:
    `python bench.py --gpus 2` never silently benches one device."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env={**os.environ, "OMP_NUM_THREADS": "1"})
    assert r.returncode != 0
    assert "re-executing as -m torch.distributed.run --nnodes=1 --nproc-per-node=2" in r.stderr
    assert "needs an MI355X" in r.stderr          # raised inside the spawned ranks
    assert "motion-frames" not in r.stdout        # no bench line from a run that did not happen


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    from pantomatrix_amd import dist as pd
    assert pd.init("gloo") is not None
    calls = []

    def step():                                   # rank 1 is the slow replica
        calls.append(1)
        time.sleep(0.02 * (1 + rank))
        return ("poses",)

    elapsed, out = bench.timed_steps(step, steps=5, warmup=2, barrier=pd.barrier, reduce_max=pd.max_over_ranks)
    line = bench.result_line("f16x3", elapsed, 5, 2, world, 64 * 120, 64, 128, "hipGraph replay")
    q.put((rank, len(calls), elapsed, line))
    pd.finalize()


def test_timed_region_on_two_gloo_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, n0, e0, l0), (_, n1, e1, l1) = res
    assert n0 == n1 == 7                                    # 2 warm-up + exactly 5 timed steps on every rank
    assert e0 == e1 and e0 >= 5 * 0.04                      # both ranks report the slowest rank's time
    assert l0["n_gpus"] == 2 and l0["scaling"] == "weak" and l0["steps"] == 5 and l0["warmup"] == 2
    assert abs(l0["value"] - 2 * 64 * 120 * 5 / e0) < 1e-6  # whole-job frames / max-over-ranks time
    assert l0["config"]["frames_out_per_clip"] == 120 and "replicas x2" in l0["config"]["parallelism"]


def _train_leg_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import bench
    import common
    import fake_ops
    from pantomatrix_amd import dist as pd
    assert pd.init("gloo") is not None
    torch.set_num_threads(2)
    models = common.product_models(precision="fp32")
    with fake_ops.installed(), torch.no_grad():
        out = bench.bench_train_step_ranks(torch.device("cpu"), world, rank, pd.barrier, pd.max_over_ranks, steps=1, warmup=1, batch=1, models=models)
    q.put((rank, out))
    pd.finalize()


def test_train_step_leg_on_two_gloo_ranks():
    """bench.py's training leg at N > 1 ranks (BASELINE configs[2]): both arms run on two gloo ranks with the CPU
    stand-ins of the kernels (eager — a captured graph needs the device and the nccl backend): the exchanging arm issues the step's
    collectives (4 bucket all-reduces, SyncBatchNorm's 12 all-gathers and 12 + 1 + 1 small all-reduces), the other arm none; both ranks
    report the same max-over-ranks time, the whole-job rate and the exposed exchange time."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_leg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=1500) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, o0), (_, o1) = res
    for o in (o0, o1):
        assert o["backend"] == "gloo" and o["world"] == 2 and o["launch"] == "eager"
        w, wo = o["with_exchange"], o["without_exchange"]
        assert w["collectives_per_step"] == {"all_gather": 12, "all_reduce_small": 14, "all_reduce_bucket": 4}, w
        assert 500e6 < w["bytes_all_reduced_per_step_and_rank"] < 600e6 and len(w["bucket_bytes"]) == 4
        assert abs(o["exposed_exchange_ms"] - (w["ms_per_step"] - wo["ms_per_step"])) < 1e-9
        assert abs(o["value"] - 1 * 2 / (w["ms_per_step"] * 1e-3)) < 1e-6 and o["unit"] == "clip-windows/s"
        assert o["roofline"]["peak"] == 2 * 2500.0
    assert o0["ms_per_step"] == o1["ms_per_step"]                      # max over ranks: every rank reports the slowest


def test_main_prints_one_json_line_and_reports_a_failing_extra(monkeypatch, capsys):
    """bench.main() end to end with the device work stubbed out (a stand-in runner, stand-in profiling): ONE JSON line with the
    contract's keys and every additional object; an exception inside an additional object (here: the training-step line) is reported
    in that object's place and does not cost the headline line."""
    import json
    import numpy as np
    import torch
    import bench
    from pantomatrix_amd import dist as pd
    from pantomatrix_amd import synthetic

    class _Audio:                                                      # what main() does with the synthetic audio batch
        def pin_memory(self):
            return self

        def to(self, dev):
            return self

        def numel(self):
            return 64 * 68267

    poses = np.zeros((64, 120, 165), np.float32)
    runner = lambda a=None: (poses, np.zeros((64, 120, 100), np.float32), np.zeros((64, 120, 3), np.float32))

    def failing_train_step(dev, cpu=True):
        raise RuntimeError("out of memory (stand-in)")

    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "2", "--warmup", "1"])
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    monkeypatch.setattr(synthetic, "synthetic_audio", lambda *a, **k: _Audio())
    monkeypatch.setattr(pd, "init", lambda backend, dev=None: None)
    monkeypatch.setattr(bench, "build", lambda precision, dev, args: (object(), object(), runner, 68267))
    monkeypatch.setattr(bench, "profile_kernels", lambda r, m, v: ([], 0.0))
    monkeypatch.setattr(bench, "serialized_graph_ms", lambda *a, **k: 15.0)
    monkeypatch.setattr(bench, "roofline_report", lambda records, precision, ms, serial_ms: {"bound": "mfma", "achieved": 1.0, "peak": 2500.0, "unit": "TFLOP/s",
                                                                                             "frac": 4e-4, "traffic": None, "serialized_kernel_ms": serial_ms})
    monkeypatch.setattr(bench, "vq_argmin_large", lambda dev: {"n": 1 << 20})
    monkeypatch.setattr(bench, "cpu_baseline", lambda frames: {"value": 2000.0, "unit": "motion-frames/s", "cores": 16, "kind": "port", "sample": "stand-in"})
    monkeypatch.setattr(bench, "bench_lstm_models", lambda dev, cpu=True: {"disco": {"ms_per_step": 9.0}, "camn": {"ms_per_step": 72.0}})
    monkeypatch.setattr(bench, "bench_train_step", failing_train_step)
    monkeypatch.setattr(bench, "bench_config1", lambda precision, dev, args, cpu=True: {"b1_128f": {"ms": 4.0}, "b1_28s": {"ms": 25.0}})
    monkeypatch.setattr(bench, "bench_batch_sweep", lambda precision, dev, args: {"by_batch": {"1": {}, "8": {}, "64": {}, "256": {}}})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    bench.main()
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "pcie_inclusive", "other_precisions", "lstm_models", "train_step", "config1", "batch_sweep", "host"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["dtype"] == "f16x3" and d["vs_baseline"] is None and "workload" in d["config"]
    assert d["roofline"]["vq_argmin"]["n_1m"] == {"n": 1 << 20} and set(d["other_precisions"]) == {"bf16"}
    assert d["lstm_models"]["camn"]["ms_per_step"] == 72.0 and d["config1"]["b1_28s"]["ms"] == 25.0 and set(d["batch_sweep"]["by_batch"]) == {"1", "8", "64", "256"}
    assert d["host"]["usable_cores"] >= 1
    assert d["train_step"] == {"error": "RuntimeError: out of memory (stand-in)"}

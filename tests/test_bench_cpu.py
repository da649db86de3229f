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

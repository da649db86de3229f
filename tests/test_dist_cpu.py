"""The N>1 launch path on CPU: two gloo processes (the same env contract as torchrun) shard clips, meet at the
barrier and reduce timings exactly as bench.py does on RCCL."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from pantomatrix_amd import dist as pd
    assert pd.init("gloo") is not None
    mine = pd.shard_clips(10, rank, world)
    pd.barrier()
    elapsed = 0.5 + 0.25 * rank                        # rank 1 is the slow one
    frames = 120.0 * len(mine)
    q.put((rank, mine, pd.max_over_ranks(elapsed), pd.sum_over_ranks(frames), pd.job_throughput(frames, elapsed)))
    pd.finalize()


def test_two_rank_gloo_sharding_and_timing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, mx0, sm0, th0), (r1, c1, mx1, sm1, th1) = res
    assert c0 == [0, 2, 4, 6, 8] and c1 == [1, 3, 5, 7, 9]           # disjoint, complete, round-robin
    assert mx0 == mx1 == 0.75 and sm0 == sm1 == 1200.0               # max over ranks / total units
    assert abs(th0 - 1600.0) < 1e-9 and th0 == th1                   # whole-job rate = all units / slowest rank


def test_single_process_is_identity():
    from pantomatrix_amd import dist as pd
    assert pd.shard_clips(5, 0, 1) == [0, 1, 2, 3, 4]
    assert pd.max_over_ranks(1.5) == 1.5 and pd.job_throughput(240.0, 2.0) == 120.0

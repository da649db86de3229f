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


def _train_exchange_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import common
    from pantomatrix_amd import dist as pd
    assert pd.init("gloo") is not None
    model, _ = common.product_models(precision="f16x3")
    plan, unused = pd.emage_bucket_plan(list(model.named_parameters()))
    buckets = pd.GradientBuckets(plan)
    g = torch.Generator().manual_seed(100 + rank)
    for name, view in buckets.grads.items():                      # what a backward would write: rank-specific gradients
        view.copy_(torch.randn(view.shape, generator=g))
    local = {k: v.clone() for k, v in list(buckets.grads.items())[::37]}
    for i in range(len(buckets.flat)):                            # backward order: heads first, encoders last
        buckets.reduce(i)
    buckets.wait()
    # the bf16 exchange (SURVEY 8f1 "bf16 + fp32 master", VERDICT round 3 next #6c): half the bytes, the sum to bf16 accuracy
    half = pd.GradientBuckets(plan, exchange_dtype=torch.bfloat16)
    assert sum(half.nbytes()) * 2 == sum(buckets.nbytes())
    for k, v in local.items():
        half.grads[k].copy_(v)
    for i in range(len(half.flat)):
        half.reduce(i)
    half.wait()
    for k in local:                                               # buckets.grads holds the fp32 average of the same local gradients
        want = buckets.grads[k]
        assert float((half.grads[k] - want).abs().max()) <= 2e-2 * float(want.abs().max()) + 1e-6, k
    # SyncBatchNorm statistics of two layers in one message
    x = [torch.randn(5 + rank, 64, generator=g), torch.randn(7, 128, generator=g) + rank]
    means, variances = pd.sync_batch_stats([t.sum(0) for t in x], [(t * t).sum(0) for t in x], [t.shape[0] for t in x])
    np_ = lambda t: t.detach().numpy().copy()                     # by value: the parent reads after this process is gone
    q.put((rank, [t for t, _ in plan], buckets.nbytes(), len(unused), {k: np_(v) for k, v in local.items()},
           {k: np_(buckets.grads[k]) for k in local}, [np_(t) for t in x], [np_(t) for t in means], [np_(t) for t in variances]))
    pd.finalize()


def test_training_exchange_on_two_gloo_ranks():
    """The training path's only collectives: bucketed gradient all-reduce in backward order and the
    SyncBatchNorm statistics exchange, on 2 gloo ranks with the real EMAGE parameter tree."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, tags, nbytes, n_unused, loc0, red0, x0, m0, v0), (_, _, _, _, loc1, red1, x1, m1, v1) = res
    assert tags == ["heads", "cross", "self_face", "encoders"] and n_unused > 0        # the unused template layers get no bucket
    assert 500e6 < sum(nbytes) < 600e6 and min(nbytes) > 30e6                           # ~555 MB fp32 in 4 large messages
    import numpy as np
    for k in loc0:                                                                     # every rank ends with the mean gradient
        want = (loc0[k] + loc1[k]) / 2
        assert np.allclose(red0[k], want, atol=1e-6) and np.array_equal(red0[k], red1[k]), k
    for i in range(2):                                                                 # global-batch statistics == one big batch
        allx = np.concatenate([x0[i], x1[i]])
        assert np.allclose(m0[i], allx.mean(0), atol=1e-5) and np.allclose(v0[i], allx.var(0), atol=1e-4)
        assert np.array_equal(m0[i], m1[i])


def _trainer_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import common
    import fake_ops
    import train_common as tc
    from pantomatrix_amd import dist as pd
    from pantomatrix_amd import training
    assert pd.init("gloo") is not None
    torch.set_num_threads(2)
    batch, _, masks, random_mask, _ = tc.oracle_step(seed=20 + rank, iteration=0)          # each rank: its own draws (its own shard of the data)
    pick = lambda grads: {k: v.clone().numpy() for k, v in grads.items() if k.startswith(("face_out_proj", "audio_encoder_body.feat_extractor.5.conv2", "mask_embedding"))}
    seen = {}
    # this rank's own gradients: the same step with the exchange switched off
    model_l, vq = common.product_models(precision="fp32")
    with fake_ops.installed(), torch.no_grad():
        training.Trainer(model_l, vq, exchange=False).step(batch, 0, masks, random_mask, grad_hook=lambda g: seen.update(local=pick(g)))
    # the data-parallel step: buckets all-reduced by the trainer itself (SUM; Adam applies the 1 / world)
    model, vq = common.product_models(precision="fp32")
    trainer = training.Trainer(model, vq)
    with fake_ops.installed(), torch.no_grad():
        trainer.step(batch, 0, masks, random_mask, grad_hook=lambda g: seen.update(summed=pick(g)))
        log1 = list(trainer.exchange_log)
        trainer.step(batch, 0, masks, random_mask)            # second step: the learned schedule overlaps the exchange with the third backward
        log2 = list(trainer.exchange_log)
    after = {k: model._flat_params()[k].clone().numpy() for k in ("face_out_proj.weight", "mask_embedding")}
    q.put((rank, seen["local"], seen["summed"], after, [b.numel() for b in trainer.buckets.flat], log1, log2, dict(trainer.schedule)))
    pd.finalize()


def test_two_rank_training_step_averages_gradients():
    """training.Trainer.step on two gloo ranks (CPU stand-ins of the kernels): the four bucket messages are summed over ranks by the
    trainer, Adam applies the average and both replicas stay in step; from the SECOND step on every bucket's all-reduce is issued
    behind its last gradient of the step — bucket 0 (heads / refinement layers) and bucket 1 (the cross-attention stack, which takes
    no part in the third forward) before the encoders' backward has even started (Missing #4)."""
    import numpy as np
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=1200) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, loc0, sum0, after0, sizes0, log1, log2, sched), (_, loc1, sum1, after1, sizes1, _l1, _l2, sched1) = res
    assert sizes0 == sizes1 and len(sizes0) == 4 and sched == sched1
    for k in loc0:
        assert not np.array_equal(loc0[k], loc1[k])                                   # different data on the two ranks
        np.testing.assert_allclose(sum0[k], loc0[k] + loc1[k], rtol=1e-6, atol=1e-9)
        np.testing.assert_array_equal(sum0[k], sum1[k])
    for k in after0:
        np.testing.assert_array_equal(after0[k], after1[k])                            # replicas stay in step
    # step 1 learns the schedule: every reduce behind the third backward
    i_done = log1.index(("backward_done", 2))
    assert all(log1.index(e) > i_done for e in log1 if e[0] == "reduce") and sum(e[0] == "reduce" for e in log1) == 4
    # step 2: bucket 1 at the very start of the third backward, bucket 0 behind the heads / refinement layers, the encoders' bucket last
    red = {e[1]: e[2] for e in log2 if e[0] == "reduce"}
    i_done = log2.index(("backward_done", 2))
    assert sorted(red) == [0, 1, 2, 3] and all(log2.index(e) < i_done for e in log2 if e[0] == "reduce")
    assert red[1] == -1 and -1 < red[0] < red[2] < red[3] and sched == red
    assert log2.index(("backward_done", 1)) < log2.index(("reduce", 1, -1)) and log2[-1] == ("wait",)


def _sync_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import common
    import fake_ops
    import train_common as tc
    from pantomatrix_amd import dist as pd
    from pantomatrix_amd import training
    assert pd.init("gloo") is not None
    torch.set_num_threads(2)
    bs, per = 4, 2
    batch, _, masks, random_mask, _ = tc.oracle_step(seed=31, iteration=0, bs=bs)              # the same global draws on every rank
    lo, hi = rank * per, (rank + 1) * per
    model, vq = common.product_models(precision="fp32")
    trainer = training.Trainer(model, vq, sync_bn=True)
    got = {}

    def spy(grads):                                        # the trainer has summed the buckets over ranks: the data-parallel average is / world
        got.update({k: (v / world).clone().numpy() for k, v in grads.items()})

    import torch.distributed as tdist
    calls = {"all_gather": 0, "all_reduce_small": 0, "all_reduce_bucket": 0}
    real_gather, real_reduce = tdist.all_gather_into_tensor, tdist.all_reduce

    def count_gather(*a, **k):
        calls["all_gather"] += 1
        return real_gather(*a, **k)

    def count_reduce(t, *a, **k):
        calls["all_reduce_bucket" if t.numel() > 1_000_000 else "all_reduce_small"] += 1
        return real_reduce(t, *a, **k)

    tdist.all_gather_into_tensor, tdist.all_reduce = count_gather, count_reduce
    try:
        with fake_ops.installed(), torch.no_grad():
            trainer.step({k: v[lo:hi] for k, v in batch.items()}, 0, [tc.shard_masks(m, lo, hi, bs) for m in masks], random_mask[lo:hi], grad_hook=spy)
    finally:
        tdist.all_gather_into_tensor, tdist.all_reduce = real_gather, real_reduce
    # VERDICT round 3, next #6b/d: the SyncBatchNorm exchanges of a step.  Round 3: one all-gather per BatchNorm and forward (32 x 3) + one
    # all-reduce per BatchNorm and backward (32 x 3) = 192, each followed by a host read of the row count.  Now: the step's three forwards share
    # ONE WavEncoder pass, whose blocks run the two encoders in lock step — per block one exchange for the bn1 pair and one for the bn2 +
    # shortcut BatchNorms: 12 all-gathers forward, 12 all-reduces backward (the BatchNorms of consecutive stages depend on each other: no
    # further merging keeps nn.SyncBatchNorm's arithmetic), plus ONE integer all-reduce (the clips of all ranks) per step
    # round 6: + ONE integer all-reduce (MAX) of the non-finite-LOSS flag, so `on_nonfinite="raise"` raises on every rank together
    assert calls == {"all_gather": 12, "all_reduce_small": 12 + 1 + 1, "all_reduce_bucket": 4}, calls
    keep = ("audio_encoder_face.feat_extractor.0.bn1.weight", "audio_encoder_body.feat_extractor.4.conv2.weight", "face_out_proj.weight",
            "audio_motion_cross_attn.layers.3.linear1.bias", "mask_embedding", "motion_encoder.main.0.weight", "speaker_embedding_body.weight")
    rv = model._flat_params()["audio_encoder_body.feat_extractor.2.bn2.running_var"].clone().numpy()
    q.put((rank, {k: got[k] for k in keep}, rv))
    pd.finalize()


def test_two_ranks_with_sync_batchnorm_equal_one_rank_on_the_whole_batch():
    """Data-parallel training is the single-process step: 2 gloo ranks x 2 clips with SyncBatchNorm statistics / gradient sums and
    the bucketed gradient average give the gradients (and BatchNorm buffers) of ONE process on the 4 clips (CPU stand-ins)."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import common
    import fake_ops
    import train_common as tc
    from pantomatrix_amd import training
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    batch, _, masks, random_mask, _ = tc.oracle_step(seed=31, iteration=0, bs=4)
    model, vq = common.product_models(precision="fp32")
    ref = {}
    with fake_ops.installed(), torch.no_grad():
        training.Trainer(model, vq).step(batch, 0, masks, random_mask, grad_hook=lambda g: ref.update({k: v.clone().numpy() for k, v in g.items()}))
    ref_rv = model._flat_params()["audio_encoder_body.feat_extractor.2.bn2.running_var"].numpy()
    res = sorted((q.get(timeout=900) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for _, grads, rv in res:
        np.testing.assert_allclose(rv, ref_rv, rtol=2e-5, atol=1e-7)
        for k, g in grads.items():
            scale = float(np.abs(ref[k]).max())
            if k.startswith("audio_encoder"):                    # LeakyReLU kink flips between differently rounded runs: L2 criterion
                assert float(np.linalg.norm(g - ref[k])) <= 5e-2 * float(np.linalg.norm(ref[k])), k
            else:
                # the two runs round the BatchNorm statistics differently (fp32 sums over ranks): activations within rounding of a
                # ReLU / LeakyReLU kink take the other slope, which moves single gradient entries by up to ~1e-3 of the scale
                assert float(np.abs(g - ref[k]).max()) <= 2e-3 * scale + 1e-7, (k, float(np.abs(g - ref[k]).max()), scale)
                assert float(np.linalg.norm(g - ref[k])) <= 2e-3 * float(np.linalg.norm(ref[k])) + 1e-7, k

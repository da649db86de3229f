"""Round-5 host-logic tests (CPU): the ADVICE round-4 items — the SyncBatchNorm clip total, the skip decision of a step, the lock-step
recording guard, the staleness stamp of the packed operands — and the gradient exchange at world size 1."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest
import torch

import common
import fake_ops
import train_common as tc
from pantomatrix_amd import dist as pdist
from pantomatrix_amd import ops, training


def _golden_step(golden_dir):
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    batch, _, masks, random_mask, _ = tc.oracle_step(int(g["seed"]), int(g["iteration"]))
    return g, batch, masks, random_mask


def test_sync_bn_clip_total_is_exchanged_by_every_step(golden_dir, monkeypatch):
    """the global clip count of SyncBatchNorm is exchanged ONCE PER STEP by every rank (not cached per local batch
    size across steps): a neighbour's changed batch is seen at once — the BatchNorm row counts and the n / (n - 1) running-variance
    correction follow it — and every rank issues the same collectives at the same point."""
    g, batch, masks, random_mask = _golden_step(golden_dir)
    model, vq = common.product_models(precision="fp32")
    trainer = training.Trainer(model, vq, sync_bn=True)
    asked, totals = [], iter([2, 6])                       # step 1: this rank alone; step 2: a neighbour shows up with 4 clips

    def fake_total(value, device="cpu", group=None):
        asked.append(int(value))
        return next(totals)

    monkeypatch.setattr(pdist, "total_over_group", fake_total)
    name = "audio_encoder_body.feat_extractor.2.bn2"
    with fake_ops.installed(), torch.no_grad():
        trainer.step(batch, 0, masks, random_mask)
        n1 = trainer.fwd._bn_count[name]
        trainer.step(batch, 0, masks, random_mask)
        n2 = trainer.fwd._bn_count[name]
    assert asked == [2, 2], asked                          # one exchange per step (the step's three forwards share one encoder pass)
    assert n2 == 3 * n1, (n1, n2)                          # rows / b x clips of all ranks: the new total was used, not a cached one


def test_nonfinite_loss_with_finite_gradients_leaves_the_bookkeeping_alone(golden_dir, monkeypatch):
    """the skip decision is the DEVICE's health word only.  A non-finite loss whose gradients were all finite
    means the update was applied: the host must not decrement `steps_done`, halve `grad_scale` or ask for a re-capture (under DDP the
    losses are rank-local — a rank acting on its own loss would run a warm-up step, with collectives, that the others do not)."""
    g, batch, masks, random_mask = _golden_step(golden_dir)
    real = training.losses

    def poisoned(cfg, pred, index, latent, ws=None):
        rec, cls = real(cfg, pred, index, latent, ws)
        return rec * float("inf"), cls                     # the loss VALUE is inf; the backward never reads it

    monkeypatch.setattr(training, "losses", poisoned)
    model, vq = common.product_models(precision="fp32")
    before = {k: v.clone() for k, v in model._flat_params().items()}
    trainer = training.Trainer(model, vq)
    with fake_ops.installed(), torch.no_grad():
        with pytest.raises(FloatingPointError, match="update WAS applied"):
            trainer.step(batch, 0, masks, random_mask)
        assert int(trainer.health) == 0 and trainer.steps_done == 1 and trainer.skipped_steps == 0 and trainer.nonfinite_loss_steps == 1
        assert trainer.fwd.grad_scale == 1024.0 and not trainer._recapture_pending
        assert all(st["step"] == 1 for st in trainer.state.values())
        assert not torch.equal(model._flat_params()["face_out_proj.weight"], before["face_out_proj.weight"])      # the device did update
        trainer.on_nonfinite = "skip"
        losses = trainer.step(batch, 0, masks, random_mask)
        assert trainer.steps_done == 2 and trainer.skipped_steps == 0 and trainer.nonfinite_loss_steps == 2 and trainer.fwd.grad_scale == 1024.0
        assert losses["all"] != losses["all"] or abs(losses["all"]) == float("inf")


def test_lockstep_check_mode_flags_a_torch_op_on_a_recorded_output(monkeypatch):
    """inside a lock-step chain emage ops are deferred while torch ops run at once.  `ops.LOCKSTEP_CHECK` turns
    that silent hazard into an error: a torch operator touching the storage of a recorded, not yet launched emage output raises;
    views, fresh allocations and torch ops on other tensors pass."""
    monkeypatch.setattr(ops, "LOCKSTEP_CHECK", True)
    a, b = torch.randn(8, 64), torch.randn(8, 64)
    monkeypatch.setattr(ops.Lockstep, "run", lambda self: None)          # no device here: the recorded launches are never issued
    with ops.lockstep() as ls:
        with ls.chain():
            out = torch.empty(8, 64)
            ops._add(ops.F32, a, b, None, None, out, 0, 0, 3)                              # recorded, not launched
            assert len(ls.cur) == 1 and ls.pending
            v = out.view(4, 128)[:, :64]                                 # views are fine
            other = torch.zeros(8, 64) + a                               # torch arithmetic on other tensors is fine
            assert v.shape == (4, 64) and other.shape == (8, 64)
    for touch in (lambda t: t.zero_(), lambda t: t + 1.0, lambda t: t[:, :8].copy_(torch.ones(8, 8)), lambda t: float(t.sum())):
        with pytest.raises(RuntimeError, match="recorded in a chain and not launched yet"):
            with ops.lockstep() as ls:
                with ls.chain():
                    out = torch.empty(8, 64)
                    ops._add(ops.F32, a, b, None, None, out, 0, 0, 3)
                    touch(out)
    assert ops._RECORDER[0] is None
    monkeypatch.setattr(ops, "LOCKSTEP_CHECK", False)
    with ops.lockstep() as ls:                                           # the default: no guard, no bookkeeping
        with ls.chain():
            out = torch.empty(8, 64)
            ops._add(ops.F32, a, b, None, None, out, 0, 0, 3)
            out.zero_()
            assert not ls.pending


def test_version_stamp_sees_replaced_tensors_and_raw_pointer_updates():
    """the staleness stamp of the packed operands carries the IDENTITY of every parameter / buffer (a tensor object
    replaced after packing is seen) and `bump_versions` marks updates made through raw device pointers (`emage_adam_multi`)."""
    model, _vq = common.product_models(precision="fp32")
    s0 = model._version_stamp()
    assert s0 == model._version_stamp()
    mod = model.get_submodule("face_out_proj")
    old = mod.weight
    mod.weight = torch.nn.Parameter(old.detach().clone())               # same values, same version, NEW object
    s1 = model._version_stamp()
    assert s1 != s0
    model.bump_versions([mod.weight])
    s2 = model._version_stamp()
    assert s2 != s1 and mod.weight._version == 1
    bn = model.get_submodule("audio_encoder_body.feat_extractor.2.bn2")
    bn._buffers["running_var"] = bn._buffers["running_var"].clone()      # a re-registered buffer
    assert model._version_stamp() != s2
    with torch.no_grad():
        mod.weight.mul_(1.0)                                             # an ordinary in-place update is still seen
    assert mod.weight._version == 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _single_rank_worker(port, q):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    assert pdist.init("gloo") is not None
    torch.set_num_threads(4)
    batch, _, masks, random_mask, _ = tc.oracle_step(seed=31, iteration=0, bs=2)
    model, vq = common.product_models(precision="fp32")
    trainer = training.Trainer(model, vq, sync_bn=True)
    assert trainer._exchanging() and trainer._world() == 1
    got = {}
    with fake_ops.installed(), torch.no_grad():
        l1 = trainer.step(batch, 0, masks, random_mask, grad_hook=lambda gr: got.update({k: v.clone().numpy() for k, v in gr.items()}))
        log1 = list(trainer.exchange_log)
        trainer.step(batch, 0, masks, random_mask)
        log2 = list(trainer.exchange_log)
    try:
        trainer.capture(batch, random_mask, masks)
        refused = ""
    except RuntimeError as e:
        refused = str(e)
    keep = ("face_out_proj.weight", "audio_encoder_body.feat_extractor.4.conv2.weight", "mask_embedding")
    q.put((l1["all"], {k: got[k] for k in keep}, log1, log2, refused))
    pdist.finalize()


def test_single_rank_process_group_runs_the_exchange():
    """A process group of ONE rank runs the whole exchange (like DistributedDataParallel does): four bucket all-reduces — the identity —
    started from the learned schedule during the third backward, and the step equals the step without a process group.  This is the
    code path tests/test_rccl_gpu.py drives through RCCL on the device.  gloo cannot be captured into a graph: `capture` says so."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_rank_worker, args=(_free_port(), q))
    p.start()
    batch, _, masks, random_mask, _ = tc.oracle_step(seed=31, iteration=0, bs=2)
    model, vq = common.product_models(precision="fp32")
    ref = {}
    with fake_ops.installed(), torch.no_grad():
        l_ref = training.Trainer(model, vq).step(batch, 0, masks, random_mask, grad_hook=lambda gr: ref.update({k: v.clone().numpy() for k, v in gr.items()}))
    loss, grads, log1, log2, refused = q.get(timeout=600)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert abs(loss - l_ref["all"]) <= 1e-6 * abs(l_ref["all"])
    for k, v in grads.items():
        assert float(np.abs(v - ref[k]).max()) <= 2e-3 * float(np.abs(ref[k]).max()) + 1e-7, k      # SyncBatchNorm's float64 merge vs plain fp32 statistics
    assert [e[0] for e in log1].count("reduce") == 4 and log1[-1] == ("wait",)
    red = [e for e in log2 if e[0] == "reduce"]
    assert len(red) == 4 and log2.index(("backward_done", 2)) > log2.index(red[1])                  # overlapped with the third backward from step 2 on
    assert "nccl" in refused and "gloo" in refused

"""Multi-GPU plumbing for the inference path: clips are independent units, so ranks are replicas —
one process per GPU, clip i handled by rank i % world, NO collective on the data path.  torch.distributed
(backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests) is used only to line ranks up and to reduce timings.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str, device=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun's env)."""
    rank, _local, world = env_world()
    if world == 1 and "RANK" not in os.environ:      # plain `python bench.py`: no process group at all
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist.group.WORLD


def is_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def shard_clips(n_clips: int, rank: int, world: int):
    """Indices of the clips rank `rank` owns: round-robin, i -> rank i % world."""
    return list(range(rank, n_clips, world))


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device="cpu") -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def job_throughput(units_local: float, elapsed_local: float, device="cpu") -> float:
    """Whole-job rate: units processed by ALL ranks / the slowest rank's time."""
    return sum_over_ranks(units_local, device) / max_over_ranks(elapsed_local, device)


def finalize():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------------------------------
# Training exchange step (train_emage_audio.py:214,248-251): the ONLY collectives of the whole path.
# One process per GPU; gradients are averaged with a few large bucketed all-reduces over RCCL (xGMI is point-to-point:
# a ring all-reduce is bound per link, so few large messages beat many small ones).  `training.Trainer` accumulates the
# backward kernels' results straight into the bucket buffers below and starts bucket i's all-reduce as soon as its last
# gradient of the step is written (during the third backward: the overlap DDP gives the reference, T:251).
# SyncBatchNorm's per-channel statistics travel as one small all-reduce per BatchNorm.  Backend "nccl" == RCCL on ROCm,
# "gloo" in tests.  `gradient_allreduce_hook` is the non-overlapped form (a `grad_hook` that reduces after the backward).
# ----------------------------------------------------------------------------------------------------------------------
EMAGE_BUCKET_PREFIXES = (
    # backward order of EmageAudioModel (M:315-330 run last in forward, so their gradients are ready first)
    # (the face branch runs BEFORE the body branch in forward, M:288-294: its heads belong to the later bucket)
    ("heads", ("body_motion_decoder_", "motion2latent_", "motion_out_proj_", "motion_cls_")),
    ("cross", ("audio_motion_cross_attn.", "audio_body_motion_proj")),
    ("self_face", ("motion_self_encoder.", "face_motion_decoder.", "face_out_proj", "face_cls", "moton_proj", "audio_face_motion_proj",
                   "bodyhints_", "speaker_embedding_")),
    ("encoders", ("audio_encoder_face.", "audio_encoder_body.", "motion_encoder.", "mask_embedding")),
)


def emage_bucket_plan(named_parameters):
    """Partition (name, tensor) pairs into the four backward-ordered buckets; parameters the forward never uses
    (`transformer_en_layer.*`, `audio_motion_cross_attn_layer.*` templates, M:238-242) get no gradient and no bucket."""
    plan = [(tag, []) for tag, _ in EMAGE_BUCKET_PREFIXES]
    unused = []
    for name, p in named_parameters:
        for i, (_tag, prefixes) in enumerate(EMAGE_BUCKET_PREFIXES):
            if name.startswith(prefixes):
                plan[i][1].append((name, p))
                break
        else:
            unused.append(name)
    return plan, unused


class GradientBuckets:
    """Flat fp32 gradient buffers, one per bucket; `grads[name]` are views into them, so a backward kernel that writes a
    parameter's gradient writes straight into the message.  `reduce(i)` starts bucket i's all-reduce (async), `wait()`
    finishes all of them and divides by the world size (the DDP average, train_emage_audio.py:251)."""

    def __init__(self, plan, device="cpu", group=None, exchange_dtype=torch.float32):
        """exchange_dtype: what travels.  float32 (default): the buckets themselves are the messages (555 MB per step, no copy: what DDP does
        for the reference, T:251).  bfloat16: each bucket is cast into a bf16 message, summed over the ranks in bf16 and added back into
        the fp32 master gradients (277.8 MB per step, SURVEY 8f1 "bf16 + fp32 master"): half the bytes on the xGMI ring for 3 significant
        digits per summand — the update differs from the reference's at the 1e-3 level of a gradient entry, so it is opt-in."""
        if exchange_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("exchange_dtype must be torch.float32 or torch.bfloat16")
        self.group, self.exchange_dtype = group, exchange_dtype
        self.tags = [tag for tag, _ in plan]
        self.flat, self.grads, self._work, self._msg = [], {}, [], {}
        pad = lambda k: (k + 7) // 8 * 8                  # every view starts on a 32-byte boundary: the backward contractions accumulate into
        for _tag, params in plan:                         # them with 16-byte accesses (the few padding words stay zero and travel along)
            n = sum(pad(p.numel()) for _, p in params)
            buf = torch.zeros(n, dtype=torch.float32, device=device)
            off = 0
            for name, p in params:
                self.grads[name] = buf[off:off + p.numel()].view(p.shape)
                off += pad(p.numel())
            self.flat.append(buf)

    def nbytes(self):
        """Bytes each bucket puts on the wire."""
        es = 2 if self.exchange_dtype == torch.bfloat16 else 4
        return [b.numel() * es for b in self.flat]

    def reduce(self, i):
        if dist.is_available() and dist.is_initialized():
            msg = self.flat[i]
            if self.exchange_dtype == torch.bfloat16:
                msg = self._msg[i] = self.flat[i].to(torch.bfloat16)
            self._work.append(dist.all_reduce(msg, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self, average=True):
        """Finish the all-reduces started by `reduce`; average=True divides by the world size here (False: the caller folds the
        1 / world into its optimiser step, training.Trainer)."""
        for w in self._work:
            w.wait()
        self._work = []
        for i, msg in self._msg.items():                  # bf16 messages: the summed gradient replaces the fp32 master bucket's content
            self.flat[i].copy_(msg)
        self._msg = {}
        if average and dist.is_available() and dist.is_initialized():
            world = dist.get_world_size(self.group)
            for b in self.flat:
                b.div_(world)


class CollectiveCounter:
    """Context manager: counts what the training exchange asks the backend for while it is active — all-gathers (SyncBatchNorm forward),
    small all-reduces (SyncBatchNorm backward, the clip count) and bucket all-reduces with their bytes — by wrapping
    `torch.distributed.all_reduce` / `all_gather_into_tensor`.  Used by bench.py's `train_step.exchange` and by the RCCL / gloo tests."""

    BUCKET_ELEMENTS = 1_000_000

    def __init__(self):
        self.counts = {"all_gather": 0, "all_reduce_small": 0, "all_reduce_bucket": 0, "bucket_bytes": 0}

    def __enter__(self):
        self._gather, self._reduce = dist.all_gather_into_tensor, dist.all_reduce

        def gather(*a, **k):
            self.counts["all_gather"] += 1
            return self._gather(*a, **k)

        def reduce(t, *a, **k):
            if t.numel() > self.BUCKET_ELEMENTS:
                self.counts["all_reduce_bucket"] += 1
                self.counts["bucket_bytes"] += t.numel() * t.element_size()
            else:
                self.counts["all_reduce_small"] += 1
            return self._reduce(t, *a, **k)

        dist.all_gather_into_tensor, dist.all_reduce = gather, reduce
        return self

    def __exit__(self, *exc):
        dist.all_gather_into_tensor, dist.all_reduce = self._gather, self._reduce
        return False


def gradient_allreduce_hook(model, device=None, group=None):
    """The `grad_hook` of `training.Trainer.step` for a multi-GPU run: the step's gradients go into the four backward-ordered
    bucket messages, are summed over ranks (RCCL / gloo) and averaged (the DDP average, train_emage_audio.py:251), and come
    back in place.  Every rank runs the same forwards, so every rank holds the same gradient names."""
    plan, _unused = emage_bucket_plan([(k, v) for k, v in model.named_parameters() if v.requires_grad])     # parameters only: no BatchNorm buffers
    buckets = GradientBuckets(plan, device=device if device is not None else model.device, group=group)

    def hook(param_grads):
        for b in buckets.flat:
            b.zero_()
        for name, g in param_grads.items():
            buckets.grads[name].copy_(g)
        for i in range(len(buckets.flat)):
            buckets.reduce(i)
        buckets.wait()
        for name, g in param_grads.items():
            g.copy_(buckets.grads[name])

    hook.buckets = buckets
    return hook


def sync_batch_stats(sums, sq_sums, count, group=None, return_counts=False):
    """SyncBatchNorm's exchange (train_emage_audio.py:248): per-channel sum and sum of squares of several BatchNorm layers
    plus their element counts, summed over ranks in ONE all-reduce.  sums / sq_sums: lists of (C_i,) tensors, count: list
    of numbers.  Returns (means, biased variances) lists over the GLOBAL batch (and the global counts when asked)."""
    flat = torch.cat([torch.cat([s.reshape(-1), q.reshape(-1), torch.as_tensor([float(c)], dtype=s.dtype, device=s.device)])
                      for s, q, c in zip(sums, sq_sums, count)])
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    means, variances, counts, off = [], [], [], 0
    for s in sums:
        c = s.numel()
        tot_s, tot_q, n = flat[off:off + c], flat[off + c:off + 2 * c], flat[off + 2 * c]
        mean = tot_s / n
        means.append(mean)
        variances.append(tot_q / n - mean * mean)
        counts.append(n)
        off += 2 * c + 1
    return (means, variances, counts) if return_counts else (means, variances)


def merge_batch_stats(mean_l, var_l, rows, group=None):
    """SyncBatchNorm's statistics exchange (train_emage_audio.py:248) as a count-weighted merge of the ranks' (mean, biased variance,
    row count) in float64 — Chan et al.'s parallel form, what torch.nn.SyncBatchNorm does with mean / invstd: no E[x^2] - mean^2
    cancellation for channels with |mean| >> std, counts exact beyond 2^24 rows.  ONE all-gather of 2 C + 1 float64 values per rank.
    Returns (mean, biased variance) as float32 (C,) tensors over the GLOBAL batch and the global row count (Python int)."""
    c = mean_l.numel()
    mine = torch.cat([mean_l.reshape(-1).double(), var_l.reshape(-1).double(), torch.tensor([float(rows)], dtype=torch.float64, device=mean_l.device)])
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size(group)
        flat = torch.empty(world * (2 * c + 1), dtype=torch.float64, device=mean_l.device)
        dist.all_gather_into_tensor(flat, mine.contiguous(), group=group)
        allv = flat.view(world, 2 * c + 1)
    else:
        allv = mine.view(1, -1)
    means, variances, n = allv[:, :c], allv[:, c:2 * c], allv[:, 2 * c:2 * c + 1]
    total = n.sum()
    mean = (means * n).sum(0) / total
    m2 = (variances * n).sum(0) + (n * (means - mean) ** 2).sum(0)
    return mean.float().contiguous(), (m2 / total).float().contiguous(), int(round(float(total)))


def merge_batch_stats_many(items, group=None):
    """`merge_batch_stats` for several INDEPENDENT BatchNorms in ONE all-gather and without a host read-back: items = [(mean_l (C_i,),
    var_l (C_i,), local rows)] -> [(mean, biased variance)] float32 over the global batch.  The message of a rank is the concatenation of
    [mean | variance | rows] per item in float64; the global row count stays on the device (the caller knows it from host arithmetic:
    every rank runs the same sequence lengths, `training.TrainForward._global_rows`).  This is what lets the two WavEncoders' BatchNorms of
    a block stage share a collective: 32 exchanges per forward become 12 (train_emage_audio.py:248)."""
    dev = items[0][0].device
    mine = torch.cat([torch.cat([m.reshape(-1).double(), v.reshape(-1).double(), torch.full((1,), float(rows), dtype=torch.float64, device=dev)])
                      for m, v, rows in items])
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size(group)
        flat = torch.empty(world * mine.numel(), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(flat, mine.contiguous(), group=group)
        allv = flat.view(world, mine.numel())
    else:
        allv = mine.view(1, -1)
    out, off = [], 0
    for m, _v, _rows in items:
        c = m.numel()
        means, variances, n = allv[:, off:off + c], allv[:, off + c:off + 2 * c], allv[:, off + 2 * c:off + 2 * c + 1]
        total = n.sum()
        mean = (means * n).sum(0) / total
        m2 = (variances * n).sum(0) + (n * (means - mean) ** 2).sum(0)
        out.append((mean.float().contiguous(), (m2 / total).float().contiguous()))
        off += 2 * c + 1
    return out


def total_over_group(value: int, device="cpu", group=None) -> int:
    """Sum of an integer over the ranks (the clips of all ranks: exchanged once per step, the only host read-back SyncBatchNorm needs)."""
    if not (dist.is_available() and dist.is_initialized()):
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def sum_over_group(tensors, group=None):
    """Element-wise sum over ranks of a list of small tensors in ONE all-reduce (SyncBatchNorm's backward sums); in place."""
    if not (dist.is_available() and dist.is_initialized()):
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view(t.shape))
        off += t.numel()
    return tensors

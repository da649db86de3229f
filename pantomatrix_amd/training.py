"""EmageAudioModel.forward in TRAIN mode on the MI355X kernels — the forward half of the EMAGE training step
(/root/reference/train_emage_audio.py:130-204 drives three such forwards per step).

What differs from the inference forward (`modeling_emage_audio.EmageAudioModel.forward`), all of it behaviour of torch
modules inside /root/reference/models/emage_audio/modeling_emage_audio.py (M:) and processing_emage_audio.py (P:):

  * the two WavEncoders (P:262-314) run nn.BatchNorm1d with BATCH statistics: the convolutions are launched with their raw
    weights (no BatchNorm fold), `emage_bn_stats` reduces each conv output over all clips and positions (float64 sums) and
    updates the running statistics the way torch does, `emage_bn_apply` normalises and applies what follows inside
    BasicBlock.forward (P:283-294: LeakyReLU; the shortcut, itself batch-normalised in the downsample blocks);
  * dropout (p = 0.1) at every site torch has one — PeriodicPositionalEncoding (P:341-343), and inside
    nn.TransformerEncoderLayer / nn.TransformerDecoderLayer (M:238-262): on the attention probabilities
    (`emage_attention_dropout`), on each sub-layer output before the residual add, inside the FFN (`emage_mul_add`).

Randomness is an INPUT: `dropout_masks` is the list of mask tensors (values bernoulli / (1 - p)) in the order the reference
draws them and in the logical shapes its modules see — (B, T, d) for the positional encodings, (B, H, Tq, Tk) for attention
probabilities, (T, B, C) for the layers' dropouts (its layers are batch_first=False).  The parity tests take the masks from
the CPU oracle's replay of the reference's generator draws (oracle/emage_train_oracle.py), so outputs, BatchNorm buffers
and losses can be compared number for number; a training loop would fill them from a device generator.

No CPU fallback: every arithmetic step is a launch into libemage_hip.so.  Backward, Adam and the gradient exchange
(pantomatrix_amd/dist.py has the bucket plan) are the next pieces of this row.
"""
from __future__ import annotations

import torch

from . import ops, spec
from ._lib import F32, H2
from .modeling_emage_audio import OUT_KEYS, _Ctx, _WAV_TAPS, _rup

BN_MOMENTUM = 0.1          # nn.BatchNorm1d default
DROPOUT_P = 0.1            # PeriodicPositionalEncoding (P:329) and nn.Transformer*Layer defaults


def _capturing(device=None):
    """True while the current stream of `device` is recording a hipGraph (no host read-back, no blocking collective result)."""
    return bool(torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())


def dropout_mask_count():
    """Masks one train-mode forward consumes: 3 positional encodings, 15 decoder layers x 6, 1 encoder layer x 4."""
    return 3 + 6 * (spec.N_FACE_LAYERS + spec.N_CROSS_LAYERS + 3) + 4


_ZERO_LONG = torch.zeros((), dtype=torch.long)           # stands for a missing `num_batches_tracked` buffer


class _Masks:
    """The dropout masks of one forward: a given list (the parity tests: the reference's recorded draws), or — masks None — drawn on
    the device as they are needed (`ops.dropout_mask`: Philox keyed by rng = (seed, step, first mask id); step may be a device scalar)."""

    def __init__(self, masks, dev, rng=None, lazy=True):
        self.masks, self.i, self.dev, self.rng, self.lazy = (None if masks is None else list(masks)), 0, dev, rng, lazy
        if masks is None and rng is None:
            raise RuntimeError("train-mode forward: pass dropout_masks or an rng (seed, step, first mask id)")

    def consumed_all(self):
        return self.masks is None or self.i == len(self.masks)

    def take(self, shape, lazy=False):
        """The next mask of the forward, of logical shape `shape`.  lazy (device-drawn masks only): the caller consumes it through
        `ops.mul_add` alone, which draws the mask inside its kernel — return the key (`ops.PhiloxMask`) instead of a tensor."""
        if self.masks is None:
            seed, step, base = self.rng
            if lazy and self.lazy and shape[-1] % 4 == 0:
                self.i += 1
                return ops.PhiloxMask(shape, DROPOUT_P, seed, base + self.i - 1, step, self.dev)
            m = torch.empty(tuple(shape), dtype=torch.float32, device=self.dev)
            ops.dropout_mask(m, DROPOUT_P, seed, base + self.i, step)
            self.i += 1
            return m
        if self.i >= len(self.masks):
            raise RuntimeError(f"dropout_masks: {len(self.masks)} masks given, the forward needs {dropout_mask_count()}")
        m = self.masks[self.i]
        self.i += 1
        if tuple(m.shape) != tuple(shape):
            raise RuntimeError(f"dropout mask {self.i - 1}: shape {tuple(m.shape)}, the reference draws {tuple(shape)} here")
        return m.to(device=self.dev, dtype=torch.float32).contiguous()


class _Token:
    """Stands for a logical (rows, cols) tensor that exists only in pieces (a projection whose V columns are stored transposed)."""

    def __init__(self, shape):
        self.shape = shape


def _accumulate(dst, g, out=None):
    """out (default dst) = dst + g on fp32 (rows, cols) views: emage_add where its 16-byte granularity allows, torch otherwise
    (the 337-column motion rows)."""
    out = dst if out is None else out
    if g.shape[1] % 4 == 0 and all(t.stride(0) % 4 == 0 and t.data_ptr() % 8 == 0 for t in (dst, g, out)):
        ops.add(F32, dst, g, out=out)
    else:
        torch.add(dst, g, out=out)


class _Tape:
    """Reverse-mode bookkeeping of one forward: backward closures in launch order, gradient buffers keyed by the forward tensor
    (or token) they belong to, column-slice views routed into their parent's buffer.  Gradients are fp32 (rows, cols).
    Memory: a gradient buffer lives from its first contribution until the node that PRODUCED its tensor has run (every consumer of a
    tensor runs before its producer in reverse order, and a tensor has one producer); a closure — and with it the activations it saved —
    is dropped as soon as it has run.  So the live set shrinks as the backward walks instead of doubling until its end."""

    def __init__(self, dev):
        self.dev, self.nodes, self.g, self.views, self.keep, self.kids, self.pos = dev, [], {}, {}, {}, {}, -1
        self._fetched = None            # ids whose gradient the running node has taken (dropped behind the node)
        self.pinned = set()

    def node(self, fn):
        self.nodes.append(fn)

    def view(self, parent, view, c0, written=False):
        """`view` = parent[:, c0:c0 + width] takes part in the forward under its own identity.  written: the view is the OUTPUT of a node
        (the parent is assembled in pieces, so it has several producers): its gradient buffer then stays until the tape is dropped."""
        self.views[id(view)] = (parent, c0, view.shape[1])
        self.keep[id(view)] = view                       # ids are keys: the objects must stay alive while their entries exist
        self.kids.setdefault(id(parent), []).append(id(view))
        if written:
            self.pinned.add(id(parent))

    def buffer(self, t, zero=True):
        """The zero-initialised, tape-owned gradient buffer of `t` (created on first use).  zero=False: the caller (and its siblings)
        OVERWRITE every column block before anyone reads the buffer — the attention backward's dQ / dK / dV blocks of a projection."""
        e = self.g.get(id(t))
        if e is None:
            alloc = torch.zeros if zero else torch.empty
            e = [alloc(tuple(t.shape), dtype=torch.float32, device=self.dev), True]
            self.g[id(t)] = e
            self.keep[id(t)] = t
        elif not e[1]:
            e[0], e[1] = e[0].clone(), True
        return e[0]

    def add(self, t, g, cols=None):
        """grad(t)[:, :cols] += g.  The first gradient of a tensor is kept by reference (never modified in place afterwards)."""
        if id(t) in self.views:
            parent, c0, n = self.views[id(t)]
            dst = self.buffer(parent)[:, c0:c0 + (n if cols is None else cols)]
            _accumulate(dst, g)
            return
        e = self.g.get(id(t))
        if cols is not None and cols != t.shape[1]:
            dst = self.buffer(t)[:, :cols]
            _accumulate(dst, g)
        elif e is None:
            self.g[id(t)] = [g, False]
            self.keep[id(t)] = t
        elif e[1]:
            _accumulate(e[0], g)
        else:
            s = torch.empty(tuple(e[0].shape), dtype=torch.float32, device=self.dev)
            _accumulate(e[0], g, out=s)
            e[0], e[1] = s, True

    def add_through(self, t, make, cols=None):
        """grad(t)[:, :cols] += c for a contribution that a contraction produces: `make(prev)` launches it and returns `c` (prev None) or
        `prev + c` as a NEW tensor (prev: the gradient so far, handed to the launch as its float32 residual — the add rides in the
        epilogue instead of being a launch of its own; the same bits: fl(prev + fl(c)))."""
        e = None if id(t) in self.views or (cols is not None and cols != t.shape[1]) else self.g.get(id(t))
        if e is None:
            self.add(t, make(None), cols)
        else:
            e[0], e[1] = make(e[0]), True

    def get(self, t):
        if id(t) in self.views:                          # a column block of its parent's gradient
            parent, c0, n = self.views[id(t)]
            e = self.g.get(id(parent))
            return None if e is None else e[0][:, c0:c0 + n]
        e = self.g.get(id(t))
        if e is None:
            return None
        if self._fetched is not None:
            self._fetched.append(id(t))
        return e[0]

    def _drop(self, i):
        if i in self.pinned:
            return
        self.g.pop(i, None)
        self.keep.pop(i, None)
        for v in self.kids.pop(i, ()):
            self.views.pop(v, None)
            self.keep.pop(v, None)

    def run(self, progress=None, base=0):
        """Run the recorded closures in reverse; `pos` counts the executed nodes (what `TrainForward._param_grad` stamps a parameter's
        last contribution with); progress(pos) is called behind each node (the overlapped gradient exchange hangs on it).  base: position
        of the first node (a tape run as the continuation of another one: the step-shared WavEncoder pass behind the third backward)."""
        self.pos = base - 1
        if progress is not None and base == 0:
            progress(-1)
        k = 0
        while self.nodes:
            fn = self.nodes.pop()
            self.pos = base + k
            self._fetched = []
            try:
                fn()
            finally:
                fetched, self._fetched = self._fetched, None
            del fn
            for i in fetched:
                self._drop(i)
            if progress is not None:
                progress(base + k)
            k += 1


class StepShare:
    """What the forwards of ONE optimisation step share.  The reference runs both WavEncoders in each of the three forwards of a step
    (T:156-172) on the SAME audio with the SAME weights: the same features and the same batch statistics three times, and three backward
    passes through them whose parameter gradients are summed.  Here the first forward's encoder pass is kept — features, BatchNorm
    statistics (later forwards replay the running-buffer updates from them), and its own tape — and the encoders are differentiated ONCE,
    behind the third backward, from the SUM of the three feature gradients (the backward is linear in them): 1/3 of the WavEncoder work of a
    step, and 1/3 of its SyncBatchNorm exchanges."""

    def __init__(self):
        self.feats, self.tape, self.bn_log = None, None, []


class TrainForward:
    """Callable train-mode forward of an `EmageAudioModel` (f16x3 or fp32 precision), with `backward()` for the part of the
    network behind the convolutional front ends."""

    def __init__(self, model, sync_bn=False, group=None):
        """sync_bn: BatchNorm statistics (forward) and their gradient sums (backward) are taken over all ranks of `group`, the
        behaviour of nn.SyncBatchNorm the reference converts its model to (train_emage_audio.py:248); torch.distributed must be
        initialised."""
        if model.precision == "bf16":
            raise ValueError("the training forward runs in the fp32-storage precisions (f16x3 / fp32)")
        self.model = model
        self.sync_bn, self.group, self._bn_count = sync_bn, group, {}
        self._clips = {}                # local clip count -> clips of all ranks (sync_bn: exchanged at the first BatchNorm of EVERY eager forward, see `_clip_total`)
        self._bn_log = None             # list collecting (name, mean, var, rows) of a WavEncoder pass that later forwards of the step replay
        # f16x3 precision: the backward contractions run as split-fp16 MFMA on pre-split (EMAGE_H2) operands — gradients pre-scaled by
        # a power of two so that their fp16 planes stay normal (the loss-scaling of mixed-precision training, undone exactly in the
        # GEMM epilogue); fp32 precision keeps the exact-fp32 MFMA contractions
        self.h2_backward = model.precision == "f16x3"
        # h2_forward (round 6, A/B switch, OFF: measured 89.7-90.1 vs 87.5-89.3 ms per step): the forward's Linear contractions on EMAGE_H2 operands too —
        # weights packed as H2 images (`_engine(lin_h2=True)`), a LayerNorm writes the image of its result beside the float32 tensor, every other
        # operand is split by one `h2_cast` launch in front of its contraction; the float32 tensors the backward reads are unchanged.  The H2
        # contractions save 2.5 ms per step against the in-kernel split of EMAGE_F16X3, the 237 cast launches cost 3.5 (profiles/r06_train_h2_forward_ab.txt)
        self.h2_forward = False
        self._h2img = {}                # id(float32 operand) -> (operand, its EMAGE_H2 image), filled by the producers that write one
        self.grad_scale = 1024.0
        self.conv_backward_rows = 1 << 17                 # output rows per piece of a long convolution's backward (_conv_backward_h2)
        self._w_scale, self._wt_cache = {}, {}
        self.fuse_grad_adds = True      # a Linear's input gradient is added to what its input has collected so far by the contraction's own epilogue (A/B switch; same bits)
        self.direct_conv_dx = True      # stride-1 convolutions: the input gradient as ONE implicit-GEMM convolution of dY (A/B switch; False: dcol = dY W + col2im)
        self.accumulate_dw = True       # Linear weight gradients are added into the gradient rows by their contraction (A/B switch; False: a temporary + a queued add)
        self.defer_finalize = True      # bias / affine-gradient reductions end in batched finalize launches (`ops.FinalizeQueue`; A/B switch: False = one each)
        self._fin = ops.FinalizeQueue()
        self.splitk_workspace_bytes = 32 << 20            # scratch of the two-pass split-K weight-gradient contractions (emage_gemm_ws: K-slices as planes, added
        self._splitk_buf = None                           # in slice order — bit-reproducible gradients); 0: the fp32-atomic form of emage_gemm
        self.lazy_masks = True          # device-drawn (T, B, d) dropout masks live as Philox keys: drawn inside `mul_add`, forward and backward (-0.9 GB per forward)
        self._wt_keep = False           # True inside `Trainer._device_step`: the transposed weight images (`_weight_t_h2`) serve all three forwards
        self.range_flag = None          # int32 device counter: transposed-weight operands (backward dX) whose cached scale no longer fits
        self._range_pending, self._range_seen = [], set()
        self.tape = None
        self._pgrads = {}               # name -> gradient accumulator (`param_grads` is the flushed view of it)
        self._pg_dst, self._pg_src, self._pg_spans, self._pg_bytes = [], [], {}, 0       # queued parameter-gradient contributions (_param_grad)
        self.grad_views = None          # name -> preallocated fp32 gradient tensor (views of the exchange buckets, training.Trainer)
        self.touch = None               # dict filled with name -> tape position of the parameter's last contribution of a backward
        self._shared, self.last_run_nodes = None, 0
        self._pcache = None

    def _splitk_ws(self, dev=None):
        """The workspace of the weight-gradient contractions (one buffer per trainer: the launches of a step run in stream order)."""
        if not self.splitk_workspace_bytes:
            return None
        dev = self.model.device if dev is None else dev
        if self._splitk_buf is None or self._splitk_buf.device != dev or self._splitk_buf.numel() * 4 != self.splitk_workspace_bytes:
            self._splitk_buf = torch.empty(self.splitk_workspace_bytes // 4, dtype=torch.float32, device=dev)
        return self._splitk_buf

    # ---- packing of what the inference pack does not hold: un-folded WavEncoder convolutions ----------------------------
    def _train_pack(self, pk):
        if "train.wav_in" in pk.w:
            return
        m = self.model
        w0, b0 = [], []
        for enc in ("audio_encoder_face", "audio_encoder_body"):
            for i, (cin, cout, stride, pad, ds) in enumerate(m._wav_blocks()):
                base = f"{enc}.feat_extractor.{i}"
                if i == 0:
                    for conv in (base + ".conv1", base + ".downsample.0"):
                        w, b = pk.folded(conv, None)
                        w0.append(w.reshape(cout, _WAV_TAPS))
                        b0.append(b)
                else:
                    pk.conv(base + ".conv1.raw", base + ".conv1", extra=(base + ".downsample.0", None) if ds else None)
                pk.conv(base + ".conv2.raw", base + ".conv2")
                for bn in ("bn1", "bn2") + (("downsample.1",) if ds else ()):
                    pk.w[f"{base}.{bn}.affine"] = (pk.f32(f"{base}.{bn}.weight"), pk.f32(f"{base}.{bn}.bias"))
        pk.w["train.wav_in"] = dict(w=torch.cat(w0, 0).float().contiguous(), b=torch.cat(b0).float().contiguous())
        pk.w["train.ones"] = torch.ones(pk.w["train.wav_in"]["w"].shape[0], dtype=torch.float32, device=pk.device)
        pk.finish_range_check()

    # ---- BatchNorm bookkeeping ----------------------------------------------------------------------------------------------
    def _global_rows(self, rows, b):
        """Rows of a BatchNorm input over the GLOBAL batch (nn.SyncBatchNorm's count): every rank runs the same sequence lengths, so it is
        rows / b x (clips of all ranks) — host arithmetic on the clip total exchanged ONCE per forward / step (`_clip_total`), no read-back
        of a device count per BatchNorm."""
        return rows // b * self._clip_total(b)

    def begin_step(self):
        """Start of an optimisation step (`Trainer._device_step`): outside a graph capture the global clip count of SyncBatchNorm is
        forgotten, so the step's first BatchNorm exchanges it afresh (`_clip_total`)."""
        if self.sync_bn and not _capturing(self.model.device):
            self._clips.clear()

    def _clip_total(self, b):
        """Clips of all ranks for a local batch of b clips (sync_bn).  Exchanged — one small all-reduce with a host read-back — at the first
        BatchNorm of EVERY eager step (`begin_step`; of every eager forward outside a step: `__call__`) — with the shared encoder pass that is
        once per step —, on every rank at the same point of the launch sequence, so the collectives always pair and a rank whose neighbour changed
        its batch size never divides by a stale count.  Inside a graph capture the host cannot read a device count: the
        value of the warm-up step is used, and `Trainer.capture` documents the contract that goes with it — every rank replays the batch
        sizes it captured (a changed batch needs a re-capture on ALL ranks, as the collectives inside the graphs must pair anyway)."""
        if not self.sync_bn:
            return b
        hit = self._clips.get(b)
        if hit is None:
            if _capturing(self.model.device):
                raise RuntimeError("TrainForward: the global clip count of a SyncBatchNorm step must be known before a graph capture "
                                   "(run one eager step with the same batch size first: Trainer.capture does)")
            from . import dist as pdist
            hit = self._clips[b] = pdist.total_over_group(b, device=self.model.device, group=self.group)
        return hit

    def _bn_many(self, cx, items, b, new_stats):
        """Batch statistics of several INDEPENDENT BatchNorms: items = [(name, conv output x (M, C))] -> [(mean, biased var)]; the running
        buffers advance in `new_stats`.  sync_bn (nn.SyncBatchNorm, T:248): the local (mean, variance, rows) of ALL items travel in ONE
        all-gather and are merged count-weighted in float64 (`dist.merge_batch_stats_many`) — the two encoders' bn1 of a block, or their
        bn2 + shortcut BatchNorms, are one collective instead of two / four."""
        params = self._params()
        out = []
        if self.sync_bn:
            from . import dist as pdist
            local = [ops.bn_stats(x, None, None, BN_MOMENTUM) for _name, x in items]
            merged = pdist.merge_batch_stats_many([(mean, var, x.shape[0]) for (mean, var), (_name, x) in zip(local, items)], group=self.group)
        # fresh copies of the running buffers (updated in place below) and the advanced batch counters of ALL items: three multi-tensor launches
        src = lambda nm, sfx: new_stats.get(nm + sfx, params[nm + sfx]).detach()
        rms = torch._foreach_mul([src(name, ".running_mean").to(cx.dev, torch.float32) for name, _x in items], 1.0)
        rvs = torch._foreach_mul([src(name, ".running_var").to(cx.dev, torch.float32) for name, _x in items], 1.0)
        nbts = torch._foreach_add([new_stats.get(name + ".num_batches_tracked", params.get(name + ".num_batches_tracked", _ZERO_LONG)).detach().to(cx.dev)
                                   for name, _x in items], 1)
        for k, (name, x) in enumerate(items):
            rm, rv = rms[k], rvs[k]
            n = x.shape[0]
            if self.sync_bn:
                stats = merged[k]
                n = self._global_rows(x.shape[0], b)
                rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * stats[0])
                rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * stats[1] * (n / max(n - 1, 1)))
                self._bn_count[name] = n
            else:
                stats = ops.bn_stats(x, rm, rv, BN_MOMENTUM)
            new_stats[name + ".running_mean"], new_stats[name + ".running_var"] = rm, rv
            new_stats[name + ".num_batches_tracked"] = nbts[k]
            if self._bn_log is not None:
                self._bn_log.append((name, stats[0], stats[1], n))
            out.append(stats)
        return out

    def _bn(self, cx, name, x, new_stats, b=None):
        return self._bn_many(cx, [(name, x)], x.shape[0] if b is None else b, new_stats)[0]

    def _replay_bn(self, log, new_stats):
        """The running-buffer updates of a forward whose WavEncoder pass is SHARED with an earlier forward of the step (same audio, same
        weights: same batch statistics): momentum update and batch counter of every BatchNorm once more, from the logged statistics."""
        params = self._params()
        dev = self.model.device
        names = [e[0] for e in log]
        rms = [new_stats.get(n + ".running_mean", params[n + ".running_mean"]).detach().to(dev, torch.float32) for n in names]
        rvs = [new_stats.get(n + ".running_var", params[n + ".running_var"]).detach().to(dev, torch.float32) for n in names]
        rms = torch._foreach_mul(rms, 1 - BN_MOMENTUM)
        torch._foreach_add_(rms, [e[1] for e in log], alpha=BN_MOMENTUM)
        rvs = torch._foreach_mul(rvs, 1 - BN_MOMENTUM)
        torch._foreach_add_(rvs, torch._foreach_mul([e[2] for e in log], [BN_MOMENTUM * e[3] / max(e[3] - 1, 1) for e in log]))
        nbts = torch._foreach_add([new_stats.get(n + ".num_batches_tracked", params.get(n + ".num_batches_tracked", _ZERO_LONG)).detach().to(dev) for n in names], 1)
        for n, rm, rv, nbt in zip(names, rms, rvs, nbts):
            new_stats[n + ".running_mean"], new_stats[n + ".running_var"], new_stats[n + ".num_batches_tracked"] = rm, rv, nbt

    def _wav_encoders(self, cx, encs, audio, b, new_stats):
        """WavEncoder.forward (P:296-314) with train-mode BatchNorm for several encoders IN LOCK STEP (encs = [(name, index)]): block by
        block, so that the BatchNorm statistics of a stage — every encoder's bn1, then every encoder's bn2 and shortcut BatchNorm — form one
        exchange under sync_bn.  -> ([features (B*T', audio_f) fp32 per encoder], T')."""
        m = self.model
        blocks = m._wav_blocks()
        lens = m._wav_lengths(audio.shape[1])
        k, q = _WAV_TAPS, blocks[0][1]
        w_in = cx.pk.w["train.wav_in"]
        xs, lin = [None] * len(encs), None
        for i, (cin, cout, stride, pad, ds) in enumerate(blocks):
            bases = [f"{enc}.feat_extractor.{i}" for enc, _e in encs]
            lout = lens[i]
            rows = b * lout
            ys = []
            for (enc, e), base, x in zip(encs, bases, xs):
                if i == 0:
                    r = slice(e * 2 * q, (e + 1) * 2 * q)
                    y = torch.empty(rows, 2 * q, dtype=torch.float32, device=cx.dev)
                    ops.wav_conv_in(F32, audio, w_in["w"][r], w_in["b"][r], cx.pk.w["train.ones"][:2 * q], y, lout, stride, pad)
                else:
                    ent = cx.pk.w[base + ".conv1.raw"]
                    y, _ = cx.gemm(x, base + ".conv1.raw", conv=(stride, pad, lin, lout), m=rows, n_store=_rup(ent["n"]))
                ys.append(y)
            c1s = [y[:, :cout] for y in ys]
            st1s = self._bn_many(cx, [(base + ".bn1", c1) for base, c1 in zip(bases, c1s)], b, new_stats)
            c2s, y1s = [], []
            for base, c1, st1 in zip(bases, c1s, st1s):
                g1, b1 = cx.pk.w[base + ".bn1.affine"]
                y1 = torch.empty(rows, _rup(cout), dtype=torch.float32, device=cx.dev)
                if y1.shape[1] != cout:
                    y1.zero_()                             # padded channels feed conv2's contraction: they must be zero
                ops.bn_apply(c1, st1, g1, b1, y1[:, :cout], slope=0.01)
                c2, _ = cx.gemm(y1, base + ".conv2.raw", conv=(1, k // 2, lout, lout), m=rows, n_store=_rup(cout))
                c2s.append(c2[:, :cout])
                y1s.append(y1)
            items = []
            for base, c2, y in zip(bases, c2s, ys):        # P:287-290: bn2, then the shortcut's BatchNorm — independent of each other
                items.append((base + ".bn2", c2))
                if ds:
                    items.append((base + ".downsample.1", y[:, cout:2 * cout]))
            sts = self._bn_many(cx, items, b, new_stats)
            saved_all, outs = [], []
            for j, (base, c2, y, y1, c1, st1, x) in enumerate(zip(bases, c2s, ys, y1s, c1s, st1s, xs)):
                g2, b2 = cx.pk.w[base + ".bn2.affine"]
                out = torch.empty(rows, cout, dtype=torch.float32, device=cx.dev)
                cds, std = None, None
                if ds:
                    st2, std = sts[2 * j], sts[2 * j + 1]
                    cds = y[:, cout:2 * cout]
                    gd, bd = cx.pk.w[base + ".downsample.1.affine"]
                    ops.bn_apply(c2, st2, g2, b2, out, slope=0.01, sc=cds, sc_bn=(*std, gd, bd))
                else:
                    st2 = sts[j]
                    ops.bn_apply(c2, st2, g2, b2, out, slope=0.01, sc=x[:, :cout])
                outs.append(out)
                saved_all.append(dict(i=i, base=base, geom=(cin, cout, stride, pad, ds), x_prev=x, lin=lin, lout=lout, b=b, audio=audio, c1=c1, st1=st1,
                                      y1=y1[:, :cout], c2=c2, st2=st2, cds=cds, std=std, out=out))
            if self.tape is not None:
                self.tape.node(lambda svs=saved_all: self._wav_blocks_backward(cx, svs))
            xs, lin = outs, lout
        return xs, lens[-1]

    def _wav_encoder(self, cx, enc, e, audio, b, new_stats):
        """One encoder alone (tests; the step runs both in lock step through `_wav_encoders`)."""
        xs, ta = self._wav_encoders(cx, [(enc, e)], audio, b, new_stats)
        return xs[0], ta

    def _conv_backward(self, cx, x, cin, origins, dy, taps, stride, pad, lin, lout, nseq, need_dx):
        """Conv1d backward on channels-last rows through emage_gemm (exact fp32): x (nseq*lin, >= cin) the layer input, dy (nseq*lout, N)
        the gradient of its pre-activation output; origins = [(weight name, bias name)] of the convolutions stacked along N.
        Accumulates the parameter gradients; returns dx (nseq*lin, cin) when asked."""
        ws = [self._param(wn).float() for wn, _ in origins]                      # (Cout_i, Cin, k)
        n = sum(w.shape[0] for w in ws)
        m = dy.shape[0]
        mp = _rup(m)
        if self.h2_backward:
            return self._conv_backward_h2(cx, x, cin, origins, ws, dy, n, m, mp, taps, stride, pad, lin, lout, nseq, need_dx)
        dy_t = torch.zeros(n, mp, dtype=torch.float32, device=cx.dev)
        ops.transpose(dy, dy_t)
        col_t = ops.im2col_t(x, cin, taps, stride, pad, lin, lout, nseq, mp)     # (taps*cin, mp)
        kc = taps * cin
        dw = torch.empty(n, _rup(kc, 4), dtype=torch.float32, device=cx.dev)[:, :kc]          # 16-byte row pitch for emage_gemm's stores
        ops.gemm(F32, dy_t, col_t, None, None, None, dw, None, None, n=kc, cp=mp)
        db = ops.col_sum(dy)
        r0 = 0
        for (wn, bn), w in zip(origins, ws):
            rows = w.shape[0]
            self._param_grad(wn, slice(None), dw[r0:r0 + rows].reshape(rows, taps, cin).permute(0, 2, 1))
            self._param_grad(bn, slice(None), db[r0:r0 + rows])
            r0 += rows
        if not need_dx:
            return None
        if n % 64:
            raise RuntimeError(f"convolution backward: {n} output channels (not a multiple of 64)")
        wflat = torch.cat([w.permute(0, 2, 1).reshape(w.shape[0], taps * cin) for w in ws], 0).contiguous()        # (N, taps*cin), taps major
        dcol = torch.empty(m, _rup(kc, 4), dtype=torch.float32, device=cx.dev)[:, :kc]
        ops.gemm(F32, dy, ops.transpose(wflat), None, None, None, dcol, None, None, n=kc, cp=n)
        return ops.col2im(dcol, cin, taps, stride, pad, lin, lout, nseq)

    def _conv_backward_h2(self, cx, x, cin, origins, ws, dy, n, m, mp, taps, stride, pad, lin, lout, nseq, need_dx):
        """`_conv_backward` with both contractions as split-fp16 MFMA on EMAGE_H2 operands: dW = (gs dY)^T im2col(X) with the im2col matrix
        written directly as an H2 image, dcol = (gs dY) W with the flattened weights converted once per forward.  Long inputs (the first
        WavEncoder blocks: 56 clips x 6 827 positions) are walked in pieces of whole sequences of about `conv_backward_rows` output rows, so
        the im2col image, the transposed dY and dcol exist for one piece at a time (0.5 GB instead of 1.5 GB each); dW is summed over the pieces."""
        gs = self.grad_scale
        kc = taps * cin
        per = max(1, self.conv_backward_rows // max(lout, 1))
        pieces = [(s0, min(s0 + per, nseq)) for s0 in range(0, nseq, per)]
        w_t = wsc = None
        np_ = _rup(n)
        # stride-1 'same' convolutions (every conv2 of the WavEncoder blocks, the motion encoder): dX is itself a convolution of dY with the
        # flipped, transposed filters — ONE implicit-GEMM launch per piece writing dX, instead of dcol = dY W (M x taps*Cin float32: 1.5 GB for
        # the first block at 56 clips) + col2im
        direct = need_dx and self.direct_conv_dx and stride == 1 and lin == lout and 2 * pad == taps - 1 and cin % 8 == 0
        if direct:
            key = ("conv^T",) + tuple(wn for wn, _ in origins)
            hit = self._wt_cache.get(key)
            if hit is None:
                wcat = torch.cat(ws, 0) if len(ws) > 1 else ws[0]                                                      # (N, Cin, k)
                wrev = torch.nn.functional.pad(wcat.flip(2).permute(1, 2, 0), (0, np_ - n)).reshape(cin, taps * np_).contiguous()   # [ci][k-1-tap][co], co padded
                wsc = self._backward_scale(key, wcat)
                hit = self._wt_cache[key] = (ops.h2_cast(wrev, taps * np_, scale=wsc / 16.0), wsc)
            w_t, wsc = hit
        elif need_dx:
            key = ("conv",) + tuple(wn for wn, _ in origins)
            hit = self._wt_cache.get(key)
            if hit is None:
                wflat = torch.cat([w.permute(0, 2, 1).reshape(w.shape[0], kc) for w in ws], 0).contiguous()          # (N, taps*cin), taps major
                wsc = self._backward_scale(key, wflat)
                hit = self._wt_cache[key] = (ops.h2_cast(wflat, np_, scale=wsc / 16.0, transpose=True), wsc)           # (taps*cin, rup64(N))
            w_t, wsc = hit
        dw = dx = None
        for s0, s1 in pieces:
            whole = len(pieces) == 1
            xs = x if whole else x[s0 * lin:s1 * lin]
            dys = dy if whole else dy[s0 * lout:s1 * lout]
            ms = (s1 - s0) * lout
            mps = _rup(ms)
            dy_t = ops.h2_cast(dys, mps, scale=gs, transpose=True)                          # (N, mps)
            col_t = ops.im2col_t_h2(xs, cin, taps, stride, pad, lin, lout, s1 - s0, mps)    # (taps*cin, mps)
            dwp = torch.empty(n, _rup(kc, 4), dtype=torch.float32, device=cx.dev)[:, :kc]
            ops.gemm(H2, dy_t, col_t, None, None, None, None, dwp, None, n=kc, cp=mps, w_scale=16.0, a_scale=16.0 * gs, workspace=self._splitk_ws(cx.dev))
            del col_t, dy_t
            dw = dwp if dw is None else dw.add_(dwp)
            if direct:
                dy_h = ops.h2_cast(dys, np_, scale=gs)
                if dx is None:
                    dx = torch.empty(nseq * lin, cin, dtype=torch.float32, device=cx.dev)
                ops.gemm(H2, dy_h, w_t, None, None, None, None, dx if whole else dx[s0 * lin:s1 * lin], None, n=cin, cp=np_, taps=taps, stride=1, pad=taps - 1 - pad,
                         lin=lout, lout=lin, m=(s1 - s0) * lin, w_scale=wsc, a_scale=16.0 * gs)
                del dy_h
            elif need_dx:
                dy_h = ops.h2_cast(dys, np_, scale=gs)
                dcol = torch.empty(ms, _rup(kc, 4), dtype=torch.float32, device=cx.dev)[:, :kc]
                ops.gemm(H2, dy_h, w_t, None, None, None, None, dcol, None, n=kc, cp=np_, w_scale=wsc, a_scale=16.0 * gs, workspace=self._splitk_ws(cx.dev))
                del dy_h
                part = ops.col2im(dcol, cin, taps, stride, pad, lin, lout, s1 - s0)
                del dcol
                if whole:
                    dx = part
                else:
                    if dx is None:
                        dx = torch.empty(nseq * lin, cin, dtype=torch.float32, device=cx.dev)
                    dx[s0 * lin:s1 * lin].copy_(part)
                del part
        db = ops.col_sum(dy)
        r0 = 0
        for (wn, bn), w in zip(origins, ws):
            rows = w.shape[0]
            self._param_grad(wn, slice(None), dw[r0:r0 + rows].reshape(rows, taps, cin).permute(0, 2, 1))
            self._param_grad(bn, slice(None), db[r0:r0 + rows])
            r0 += rows
        return dx

    def _bn_backward_many(self, items):
        """Training-mode BatchNorm backward of several INDEPENDENT BatchNorms: items = [(name, x, stats, dy)] -> [dx]; parameter gradients
        are accumulated.  sync_bn: the per-channel sums of ALL items are summed over the ranks in ONE all-reduce (local sums -> parameter
        gradients, averaged later with all the others; their sum over ranks -> the input gradients)."""
        out = []
        if self.sync_bn:
            from . import dist as pdist
            sums = [ops.bn_backward_sums(x, stats, dy) for _name, x, stats, dy in items]
            tot = pdist.sum_over_group([t.clone() for pair in sums for t in pair], group=self.group)
        for k, (name, x, stats, dy) in enumerate(items):
            gamma = self._param(name + ".weight").float()
            if self.sync_bn:
                dg, db = sums[k]
                dx = ops.bn_backward_apply(x, stats, gamma, dy, (tot[2 * k], tot[2 * k + 1]), self._bn_count[name])
            else:
                dx, dg, db = ops.bn_backward(x, stats, gamma, dy)
            self._param_grad(name + ".weight", slice(None), dg)
            self._param_grad(name + ".bias", slice(None), db)
            out.append(dx)
        return out

    def _bn_backward(self, name, x, stats, dy):
        return self._bn_backward_many([(name, x, stats, dy)])[0]

    def _wav_blocks_backward(self, cx, svs):
        """BasicBlock.forward (P:283-294) backwards for the same block of several encoders in lock step: LeakyReLU, [bn2 and the
        (batch-normalised) shortcut of every encoder: one exchange], conv2, LeakyReLU, [bn1 of every encoder: one exchange], conv1
        (+ shortcut conv)."""
        tape = self.tape
        svs = [sv for sv in svs if tape.get(sv["out"]) is not None]
        if not svs:
            return
        k = _WAV_TAPS
        dvs = [ops.act_backward(tape.get(sv["out"]), sv["out"], 0.01) for sv in svs]
        items = []
        for sv, dv in zip(svs, dvs):
            items.append((sv["base"] + ".bn2", sv["c2"], sv["st2"], dv))
            if sv["geom"][4]:
                items.append((sv["base"] + ".downsample.1", sv["cds"], sv["std"], dv))
        dxs = iter(self._bn_backward_many(items))
        dc2s, dcdss = [], []
        for sv, dv in zip(svs, dvs):
            cin, cout, stride, pad, ds = sv["geom"]
            dc2s.append(next(dxs))
            dcdss.append(next(dxs) if ds else None)
            if not ds:
                tape.add(sv["x_prev"], dv, cols=cout)
        pre1 = []
        for sv, dc2 in zip(svs, dc2s):
            cin, cout, stride, pad, ds = sv["geom"]
            base, b, lout = sv["base"], sv["b"], sv["lout"]
            dy1 = self._conv_backward(cx, sv["y1"], cout, [(base + ".conv2.weight", base + ".conv2.bias")], dc2, k, 1, k // 2, lout, lout, b, True)
            pre1.append(ops.act_backward(dy1, sv["y1"], 0.01))
        dc1s = self._bn_backward_many([(sv["base"] + ".bn1", sv["c1"], sv["st1"], g) for sv, g in zip(svs, pre1)])
        for sv, dc1, dcds in zip(svs, dc1s, dcdss):
            self._wav_block_conv1_backward(cx, sv, dc1, dcds)

    def _wav_block_conv1_backward(self, cx, sv, dc1, dcds):
        tape = self.tape
        cin, cout, stride, pad, ds = sv["geom"]
        base, k, b, lin, lout = sv["base"], _WAV_TAPS, sv["b"], sv["lin"], sv["lout"]
        origins = [(base + ".conv1.weight", base + ".conv1.bias")]
        dys = dc1
        if ds:
            origins.append((base + ".downsample.0.weight", base + ".downsample.0.bias"))
            dys = torch.cat([dc1, dcds], 1)
        if sv["i"] == 0:
            if self.h2_backward:       # the first layer's weight gradient (Cin = 1) as ONE split-fp16 contraction over all positions (split-K)
                audio = sv["audio"]
                m = dys.shape[0]
                mp = _rup(m)
                dys_t = ops.h2_cast(dys, mp, scale=self.grad_scale, transpose=True)                         # (2*cout, mp)
                col = ops.im2col_t_h2(audio.reshape(-1, 1), 1, k, stride, pad, audio.shape[1], lout, b, mp)  # (k, mp)
                dw = torch.empty(dys.shape[1], _rup(k, 4), dtype=torch.float32, device=cx.dev)[:, :k]
                ops.gemm(H2, dys_t, col, None, None, None, None, dw, None, n=k, cp=mp, w_scale=16.0, a_scale=16.0 * self.grad_scale, workspace=self._splitk_ws(cx.dev))
            else:
                dw = ops.wav_conv_in_backward(dys, sv["audio"], lout, k, stride, pad)      # (2*cout, k): conv1 rows, then the shortcut conv's
            db = ops.col_sum(dys)
            for j, (wn, bn) in enumerate(origins):
                self._param_grad(wn, slice(None), dw[j * cout:(j + 1) * cout].view(cout, 1, k))
                self._param_grad(bn, slice(None), db[j * cout:(j + 1) * cout])
        else:
            tape.add(sv["x_prev"], self._conv_backward(cx, sv["x_prev"], cin, origins, dys, k, stride, pad, lin, lout, b, True), cols=cin)

    # ---- motion pre-encoder (VQEncoderV6, P:213-235), tape-aware ----------------------------------------------------------------------
    def _conv3(self, cx, x, key, t, b, cin, slope=None, res=None):
        y, _ = cx.conv3(x, key, t, slope=slope, res=res, n_store=_rup(cx.pk.w[key]["n"]))
        if self.tape is not None:
            def bw():
                g = self.tape.get(y)
                if g is None:
                    return
                if res is not None:
                    self.tape.add(res, g)
                dpre = g if slope is None else ops.act_backward(g, y, slope)
                dx = self._conv_backward(cx, x, cin, [(key + ".weight", key + ".bias")], dpre, 3, 1, 1, t, t, b, True)
                self.tape.add(x, dx, cols=cin)
            self.tape.node(bw)
        return y

    def _motion_encoder(self, cx, x0, t, b, cin0):
        h, cin = x0, cin0
        for i in range(spec.MOTION_ENC_LAYERS):
            p = f"motion_encoder.main.{3 * i}"
            h = self._conv3(cx, h, p, t, b, cin, slope=0.2)
            cin = cx.pk.w[p]["n"]
            r = self._conv3(cx, h, f"motion_encoder.main.{3 * i + 2}.model.0", t, b, cin, slope=0.2)
            h = self._conv3(cx, r, f"motion_encoder.main.{3 * i + 2}.model.2", t, b, cin, res=h)
        return h

    # ---- differentiable pieces: each wrapper launches the forward op and, when a tape is attached, records its backward -----------
    def _image(self, x, e):
        """The EMAGE_H2 image of the float32 operand x for the packed Linear entry e: the one its producer wrote (`_layernorm`), else one cast launch."""
        hit = self._h2img.pop(id(x), None)
        if hit is not None and hit[0] is x and hit[1].shape[1] >= e["cp"]:
            return hit[1]
        k = e["k_real"]
        return ops.h2_cast(x if x.shape[1] == k else x[:, :k], e["cp"])

    def _gemm_fwd(self, cx, x, key, slope=None, out=None, **kw):
        """One forward Linear: float32 in, float32 out; on pre-split operands when the entry is an EMAGE_H2 packing (`h2_forward`)."""
        e = cx.pk.w[key]
        if e.get("dt") != H2:
            return cx.gemm(x, key, slope=slope, out=out, **kw)[0]
        if out is None and kw.get("out_t") is None:
            out = cx.f32(x.shape[0], e["n"])
        cx.gemm(self._image(x, e), key, slope=slope, out_f32=out, want=None, **kw)
        return out

    def _lin(self, cx, x, key, slope=None, out=None, need_dx=True):
        """y = act(x W^T + b) through emage_gemm (one fp32 output used both as the next operand and as the result)."""
        y = self._gemm_fwd(cx, x, key, slope=slope, out=out)
        if self.tape is not None:
            self.tape.node(lambda: self._lin_backward(cx, x, key, slope, y, y, need_dx))
        return y

    def _lin_kv(self, cx, x, key, n_first, t_rows, b):
        """A projection whose last columns are emitted transposed (V^T for emage_attention): returns (first n_first columns as
        rows, the V^T buffer, a token standing for the logical (M, N) output in the tape)."""
        n = cx.pk.w[key]["n"]
        first = cx.lo(x.shape[0], n_first)
        vt = cx.vt_buffer(b, n - n_first, t_rows)
        self._gemm_fwd(cx, x, key, out=first, out_t=vt, t_col0=n_first, t_rows=t_rows)
        token = _Token((x.shape[0], n))
        if self.tape is not None:
            self.tape.node(lambda: self._lin_backward(cx, x, key, None, token, None, True))
        return first, vt, token

    def _lin_backward(self, cx, x, key, slope, y_id, y, need_dx):
        tape = self.tape
        dy = tape.get(y_id)
        if dy is None:
            return
        ent = cx.pk.w[key]
        n, k = ent["n"], ent["k_real"]
        m = dy.shape[0]
        if dy.shape[1] != n or dy.stride(1) != 1:
            raise RuntimeError(f"{key}: gradient of shape {tuple(dy.shape)} for an output of {n} columns")
        if self.h2_backward:
            self._lin_backward_h2(cx, x, key, dy, y if slope is not None else None, slope, n, k, m, need_dx)
            return
        dpre = dy if slope is None else ops.act_backward(dy, y, slope)
        w32 = torch.cat([self._param(wn)[rs] for wn, _bn, rs in cx.pk.origin[key]], 0).float().contiguous()            # (N, K)
        mp = _rup(m)
        dpre_t = torch.zeros(n, mp, dtype=torch.float32, device=cx.dev)
        ops.transpose(dpre, dpre_t)
        x_t = torch.zeros(k, mp, dtype=torch.float32, device=cx.dev)
        ops.transpose(x[:, :k], x_t)
        dw = torch.empty(n, k, dtype=torch.float32, device=cx.dev)
        ops.gemm(F32, dpre_t, x_t, None, None, None, dw, None, None, n=k, cp=mp)                                          # dW = dpre^T x
        self._lin_param_grads(cx, key, dw, dpre)
        if need_dx:
            np_ = _rup(n)
            if np_ != n:
                raise RuntimeError(f"{key}: {n} output columns (not a multiple of 64) — pad before the backward contraction")
            w_t = ops.transpose(w32)                                                                                    # (K, N)
            dx = torch.empty(m, k, dtype=torch.float32, device=cx.dev)
            ops.gemm(F32, dpre, w_t, None, None, None, dx, None, None, n=k, cp=n)                                        # dX = dpre W
            tape.add(x, dx, cols=k)

    def _lin_param_grads(self, cx, key, dw, dpre):
        """Hand dW (N, K) and the bias gradient (column sums of dpre) of the packed Linear `key` to the parameters it stacks."""
        origin = cx.pk.origin[key]
        if len(origin) == 1:                              # the usual case: the reduction's finalize adds into the gradient rows itself
            wn, bn, rs = origin[0]
            self._param_grad(wn, rs, dw)
            self._param_grad_colsum(bn, rs, dpre)
            return
        db = ops.col_sum(dpre)
        r0 = 0
        for wn, bn, rs in origin:
            rows = rs.stop - rs.start
            self._param_grad(wn, rs, dw[r0:r0 + rows])
            self._param_grad(bn, rs, db[r0:r0 + rows])
            r0 += rows

    def _backward_scale(self, key, w):
        """Power-of-two scale of a backward weight operand: chosen at the first use of `key` (one read-back: max |w| into [2^11, 2^12)) and
        kept, so that later steps — and a captured step — convert without a host round trip.  Every later use checks ON THE DEVICE that
        max |w| * scale is still inside [2^10, 2^14) (`ops.f16_scale_out_of_range`; the fp16 hi plane overflows at 2^16) and counts a
        miss in `range_flag`, which `Trainer` reads with the losses and answers by re-deriving the scales (`reset_scales`)."""
        ws = self._w_scale.get(key)
        if ws is None:
            import math
            mx = float(w.abs().max())
            ws = self._w_scale[key] = 2.0 ** (11 - math.floor(math.log2(mx))) if mx > 0 and math.isfinite(mx) else 1.0
        elif w.numel() and key not in self._range_seen:     # once per key and step, all keys in a few launches (`flush_range_checks`)
            self._range_seen.add(key)
            self._range_pending.append((w, ws))
        return ws

    def flush_range_checks(self, new_step=False):
        """Fold the backward operands queued by `_backward_scale` into `range_flag`; new_step: forget which keys this step has seen."""
        if self._range_pending:
            ws, scales = zip(*self._range_pending)
            self._range_pending = []
            bad = ops.f16_scales_out_of_range(list(ws), list(scales))
            self.range_flag = bad if self.range_flag is None else self.range_flag + bad
        if new_step:
            self._range_seen = set()

    def reset_scales(self):
        """Forget the cached operand scales (backward weight operands here, the forward's packed operands in the model): the next
        forward chooses them afresh from the current weights."""
        self._w_scale, self._wt_cache, self.range_flag = {}, {}, None
        self._range_pending, self._range_seen = [], set()
        self.model.invalidate_packed(reset_scales=True)
        self._pcache = None

    def _weight_t_h2(self, cx, key, n, k):
        """The transposed weight of a Linear as an EMAGE_H2 operand, (K, rup64(N)) holding w * w_scale: (image, w_scale); built once per
        forward and key, scale from `_backward_scale`."""
        hit = self._wt_cache.get(key)
        if hit is not None:
            return hit
        rows = [self._param(wn)[rs] for wn, _bn, rs in cx.pk.origin[key]]
        w32 = (torch.cat(rows, 0) if len(rows) > 1 else rows[0]).float().contiguous()                                    # (N, K); one source: no copy
        ws = self._backward_scale(key, w32)
        img = ops.h2_cast(w32, _rup(n), scale=ws / 16.0, transpose=True)              # h2 images carry a fixed x16: the rest of the scale goes in front
        self._wt_cache[key] = (img, ws)
        return img, ws

    def _lin_backward_h2(self, cx, x, key, dy, y, slope, n, k, m, need_dx):
        """dW = dpre^T x, db = colsum(dpre), dX = dpre W as split-fp16 MFMA contractions (emage_gemm, EMAGE_H2).  dpre = dy through the
        activation's backward (y: the layer's saved output, None = no activation).  ONE pass over dy (`ops.grad_prep`) yields both
        gradient operands — pre-scaled by `grad_scale`, undone by the GEMM's output scale — and the bias gradient, which the reduction's
        finalize launch adds straight into the parameter's gradient rows when the packed key holds one Linear."""
        gs = self.grad_scale
        mp = _rup(m)
        origin = cx.pk.origin[key]
        single = len(origin) == 1
        bias_dst = self._grad_rows(origin[0][1], origin[0][2])[0] if single else None
        dpre_h, dpre_t, db = ops.grad_prep(dy, y, 0.0 if slope is None else slope, gs, n_store=_rup(n) if need_dx else None, m_store=mp,
                                           bias_grad=bias_dst, accumulate=single,                # (M, rup64(N)), (N, mp)
                                           defer=self._fin if (single and self.defer_finalize) else None)
        x_t = ops.h2_cast(x[:, :k], mp, scale=1.0, transpose=True)                    # (K, mp)
        wdst = self._grad_rows(origin[0][0], origin[0][2])[0] if single else None
        if self.accumulate_dw and wdst is not None and wdst.dim() == 2 and wdst.stride(1) == 1 and wdst.stride(0) % 4 == 0 and wdst.data_ptr() % 16 == 0:
            # dW added by the contraction itself (res == out_f32: the epilogue's residual add in place, or split-K atomics onto the contents)
            ops.gemm(H2, dpre_t, x_t, None, None, wdst, None, wdst, None, n=k, cp=mp, w_scale=16.0, a_scale=16.0 * gs, workspace=self._splitk_ws())
        else:
            dw = torch.empty(n, k, dtype=torch.float32, device=cx.dev)
            ops.gemm(H2, dpre_t, x_t, None, None, None, None, dw, None, n=k, cp=mp, w_scale=16.0, a_scale=16.0 * gs, workspace=self._splitk_ws(cx.dev))
            r0 = 0
            for wn, bn, rs in origin:
                rows = rs.stop - rs.start
                self._param_grad(wn, rs, dw[r0:r0 + rows])
                r0 += rows
        if not single:
            r0 = 0
            for wn, bn, rs in origin:
                rows = rs.stop - rs.start
                self._param_grad(bn, rs, db[r0:r0 + rows])
                r0 += rows
        if need_dx:
            w_t, ws = self._weight_t_h2(cx, key, n, k)

            def make(prev):                            # prev: the gradient x has collected so far (its residual path) — added by this launch's epilogue
                dx = torch.empty(m, k, dtype=torch.float32, device=cx.dev)
                if prev is not None and not (self.fuse_grad_adds and prev.shape == dx.shape and prev.stride(1) == 1 and prev.stride(0) % 4 == 0 and prev.data_ptr() % 16 == 0):
                    ops.gemm(H2, dpre_h, w_t, None, None, None, None, dx, None, n=k, cp=_rup(n), w_scale=ws, a_scale=16.0 * gs, workspace=self._splitk_ws(cx.dev))
                    out = torch.empty_like(dx)
                    _accumulate(prev, dx, out=out)
                    return out
                ops.gemm(H2, dpre_h, w_t, None, None, prev, None, dx, None, n=k, cp=_rup(n), w_scale=ws, a_scale=16.0 * gs, workspace=self._splitk_ws(cx.dev))
                return dx
            self.tape.add_through(x, make, cols=k)

    def _params(self):
        """name -> detached parameter / buffer view, built once per forward (walking the module tree per lookup costs more than
        the launches it feeds)."""
        if self._pcache is None:
            dev = self.model.device
            self._pcache = {k: v.detach().to(dev) for k, v in self.model._flat_params().items()}
        return self._pcache

    def _param(self, name):
        return self._params()[name]

    @property
    def param_grads(self):
        """name -> fp32 gradient accumulated so far (queued contributions applied first: reading is always consistent)."""
        self.flush_param_grads()
        return self._pgrads

    @param_grads.setter
    def param_grads(self, value):
        self.flush_param_grads()          # contributions queued for the previous accumulators belong to them
        self._pgrads = value

    def _grad_rows(self, name, rows):
        """The rows `rows` of parameter `name`'s gradient accumulator (the exchange bucket's view under a `Trainer`) as the destination of
        one more contribution: a queued contribution to overlapping rows is applied first; the tape position is stamped."""
        full = self._pgrads.get(name)
        if full is None:
            full = self.grad_views[name] if self.grad_views is not None else torch.zeros_like(self._param(name), dtype=torch.float32)
            self._pgrads[name] = full
        n0 = full.shape[0] if full.dim() else 1
        lo, hi = (0 if rows.start is None else rows.start), (n0 if rows.stop is None else rows.stop)
        spans = self._pg_spans.get(name)
        if spans is not None and any(lo < b and a < hi for a, b in spans):
            self.flush_param_grads()
        if self.touch is not None:
            self.touch[name] = self.tape.pos
        return full[rows], (lo, hi)

    def _param_grad(self, name, rows, g):
        """grad(name)[rows] += g — parameter-sized accumulation across the forwards of a step.  The contributions are QUEUED and applied
        in batches (`flush_param_grads`: one multi-tensor add for up to 256 of them instead of one small launch each, ~1 200 per step);
        every element still receives its contributions one by one in issue order, so the sums are the same bits."""
        dst, span = self._grad_rows(name, rows)
        g = g.reshape(dst.shape)
        self._pg_spans.setdefault(name, []).append(span)
        self._pg_dst.append(dst)
        self._pg_src.append(g)
        self._pg_bytes += g.numel() * 4
        if len(self._pg_dst) >= 256 or self._pg_bytes >= (96 << 20):
            self.flush_param_grads()

    def _param_grad_colsum(self, name, rows, x, y=None):
        """grad(name)[rows] += column sums of x (* y): the bias / affine gradients, accumulated by the reduction's own finalize launch."""
        dst, _span = self._grad_rows(name, rows)
        ops.col_sum(x, y, out=dst, accumulate=True, defer=self._fin if self.defer_finalize else None)

    def flush_param_grads(self):
        """Apply the queued parameter-gradient contributions (call before anything reads or exchanges the accumulators)."""
        self._fin.flush()                 # the queued finalize steps of the column reductions (bias / affine gradients)
        if self._pg_dst:
            # ONE pair that is not dense with equal strides (the convolutions' weight gradients arrive as permuted (Cout, taps, Cin) views) sends the
            # WHOLE multi-tensor add down torch's one-launch-per-tensor path (~440 launches per step): those pairs go in a call of their own.  The
            # queued destinations never overlap (`_grad_rows` flushes first), so the order between the two calls does not matter
            plain = [d.stride() == g.stride() and d.is_contiguous() for d, g in zip(self._pg_dst, self._pg_src)]
            for want in (True, False):
                dst = [d for d, ok in zip(self._pg_dst, plain) if ok == want]
                if dst:
                    torch._foreach_add_(dst, [g for g, ok in zip(self._pg_src, plain) if ok == want])
        self._pg_dst, self._pg_src, self._pg_spans, self._pg_bytes = [], [], {}, 0

    def _add(self, cx, a, bb, mod_b=0, grad_b=True):
        out = cx.lo(*a.shape)
        ops.add(cx.dt, a, bb, out=out, mod_b=mod_b)
        if self.tape is not None:
            def bw():
                g = self.tape.get(out)
                if g is None:
                    return
                self.tape.add(a, g)
                if grad_b and not mod_b:
                    self.tape.add(bb, g)
            self.tape.node(bw)
        return out

    def _mul_add(self, a, mask, res=None, t_rows=0):
        out = ops.mul_add(a, mask, res, mask_t_rows=t_rows)
        if self.tape is not None:
            def bw():
                g = self.tape.get(out)
                if g is None:
                    return
                self.tape.add(a, ops.mul_add(g, mask, None, mask_t_rows=t_rows))
                if res is not None:
                    self.tape.add(res, g)
            self.tape.node(bw)
        return out

    def _layernorm(self, cx, key, s_in):
        n = cx.pk.w[key]
        y = cx.lo(*s_in.shape)
        if self.h2_forward and s_in.shape[1] % 64 == 0:      # the result feeds a Linear: its EMAGE_H2 image leaves the same launch
            img = cx.f32(*s_in.shape)
            ops.layernorm(H2, s_in, n["g"], n["b"], 1e-5, None, y, img)
            self._h2img[id(y)] = (y, img)
        else:
            ops.layernorm(cx.dt, s_in, n["g"], n["b"], 1e-5, None, None, y)
        if self.tape is not None:
            def bw():
                g = self.tape.get(y)
                if g is None:
                    return
                dg, _ = self._grad_rows(key + ".weight", slice(None))
                db, _ = self._grad_rows(key + ".bias", slice(None))
                dx, _, _ = ops.layernorm_backward(s_in, n["g"], g, dgamma=dg, dbeta=db,      # d gamma / d beta += inside
                                                  defer=self._fin if self.defer_finalize else None)
                self.tape.add(s_in, dx)
            self.tape.node(bw)
        return y

    def _drop_add(self, cx, o, masks, t, res=None):
        """res + dropout(o) with the mask the reference draws on the (T, B, C) tensor."""
        m, c = o.shape
        mk = masks.take((t, m // t, c), lazy=True).view(m, c)
        return self._mul_add(o, mk, res, t_rows=t)

    def _mha(self, cx, masks, q_src, k_src, v_src, k_rows, vt, vt_rows, b, tq, tk):
        """Attention with probability dropout.  *_src = (tensor-or-token the operand is a column block of, first column); q / k
        operand views are built here.  k_rows: the row buffer holding the keys."""
        d, h = self.model.config.hidden_size, spec.N_HEAD
        q = q_src[2]
        k = k_rows
        pm = masks.take((b, h, tq, tk))
        att = cx.lo(b * tq, d)
        ops.attention_dropout(cx.gdt, q, k, vt, vt_rows, att, b, h, tq, tk, d // h, pm)
        if self.tape is not None:
            def bw():
                g = self.tape.get(att)
                if g is None:
                    return
                # tokens (projections whose column blocks are all attention operands) are covered block by block by the attention
                # backwards of their layers: no zero fill; a real tensor (the cross-attention query) may have other consumers
                gq = self.tape.buffer(q_src[0], zero=not isinstance(q_src[0], _Token))[:, q_src[1]:q_src[1] + d]
                gk = self.tape.buffer(k_src[0], zero=not isinstance(k_src[0], _Token))[:, k_src[1]:k_src[1] + d]
                gv = self.tape.buffer(v_src[0], zero=not isinstance(v_src[0], _Token))[:, v_src[1]:v_src[1] + d]
                ops.attention_backward(q, k, vt, vt_rows, pm, g, gq, gk, gv, b, h, tq, tk, d // h)
            self.tape.node(bw)
        return att

    def _self_attn(self, cx, masks, name, x, b, t):
        d = self.model.config.hidden_size
        qk, vt, tok = self._lin_kv(cx, x, name + ".sa.qkv", 2 * d, t, b)
        att = self._mha(cx, masks, (tok, 0, qk[:, :d]), (tok, d), (tok, 2 * d), qk[:, d:], vt, d, b, t, t)
        o = self._lin(cx, att, name + ".sa.out")
        return self._drop_add(cx, o, masks, t, res=x)

    def _ffn(self, cx, masks, name, x, t):
        f = self._lin(cx, x, name + ".ff1", slope=0.0)
        f = self._drop_add(cx, f, masks, t)
        o = self._lin(cx, f, name + ".ff2")
        return self._drop_add(cx, o, masks, t, res=x)

    def _encoder_layer(self, cx, masks, name, x, b, t):
        x = self._layernorm(cx, name + ".norm1", self._self_attn(cx, masks, name, x, b, t))
        return self._layernorm(cx, name + ".norm2", self._ffn(cx, masks, name, x, t))

    def _decoder_layer(self, cx, masks, name, x, b, t, mem, vt_rows, tk):
        """mem = (keys as rows, V^T buffer view, token of the K/V projection, first K column, first V column)."""
        d = self.model.config.hidden_size
        mem_k, mem_vt, tok, kc, vc = mem
        x = self._layernorm(cx, name + ".norm1", self._self_attn(cx, masks, name, x, b, t))
        q = self._lin(cx, x, name + ".ca.q")
        att = self._mha(cx, masks, (q, 0, q), (tok, kc), (tok, vc), mem_k, mem_vt, vt_rows, b, t, tk)
        o = self._lin(cx, att, name + ".ca.out")
        x = self._layernorm(cx, name + ".norm2", self._drop_add(cx, o, masks, t, res=x))
        return self._layernorm(cx, name + ".norm3", self._ffn(cx, masks, name, x, t))

    def _memory(self, cx, key, mem_rows, b, tk, n_layers):
        """K / V projection of a cross-attention memory for n_layers layers: per layer the tuple `_decoder_layer` takes."""
        d = self.model.config.hidden_size
        k, vt, tok = self._lin_kv(cx, mem_rows, key, n_layers * d, tk, b)
        return [(k[:, i * d:(i + 1) * d], vt[:, i * d:], tok, i * d, n_layers * d + i * d) for i in range(n_layers)]

    def _ppe(self, cx, masks, x, b, t):
        """PeriodicPositionalEncoding.forward (P:341-343): dropout(x + pe[:, :T]) on (B*T, d) rows."""
        m, d = x.shape
        y = self._add(cx, x, cx.pk.w["pe"][:t], mod_b=t)
        return self._mul_add(y, masks.take((b, t, d), lazy=True).view(m, d))

    def _speaker(self, cx, key, name, sid):
        rows = ops.gather_rows(cx.pk.w[key], sid, F32)
        if self.tape is not None:
            def bw():
                g = self.tape.get(rows)
                if g is None:
                    return
                if cx.pk.w[key].shape[0] != 1:
                    raise NotImplementedError("speaker-embedding gradient for more than one speaker row")
                self._param_grad(name, slice(None), ops.col_sum(g))
            self.tape.node(bw)
        return rows

    # ---- the forward --------------------------------------------------------------------------------------------------------
    def __call__(self, audio, speaker_id, masked_motion, mask, dropout_masks=None, use_audio=True, new_stats=None, tape=False, rng=None, share=None):
        """-> (dict of the 8 (B, T, 256) fp32 outputs, new_stats).  `new_stats` carries the BatchNorm running buffers from one
        forward of a step to the next (as oracle.emage_train_oracle.forward_train does); it is not written into the model.
        tape=True keeps what `backward()` needs (see there).  dropout_masks None: the masks are drawn on the device from
        rng = (seed, step, first mask id) — see `_Masks`.  share: the `StepShare` of the optimisation step this forward belongs to
        (same audio and weights as the step's other forwards): the WavEncoder pass is computed by the first of them only."""
        model = self.model
        c = model.config
        # the training forward keeps float32 activations (split inside the GEMMs in f16x3); train_only: the operand set leaves out what
        # only the eval-mode forward reads (the WavEncoder convolutions with their BatchNorms folded in)
        cx = _Ctx(model._engine(h2=False, train_only=True, lin_h2=self.h2_forward))
        pk, dev = cx.pk, cx.dev
        self._h2img = {}
        self._train_pack(pk)
        self.tape = _Tape(dev) if tape else None
        self._cx = cx
        self._pcache = None
        if not self._wt_keep:                    # inside a `Trainer` step the weights are those of the step's first forward: keep the images
            self._wt_cache = {}
        new_stats = {} if new_stats is None else new_stats
        if self.sync_bn and share is None and not _capturing(dev):      # a forward outside a `Trainer` step (the class API): the global clip
            self._clips.clear()                                         # count is exchanged afresh (see `_clip_total`; a step: `begin_step`)
        masks = _Masks(dropout_masks, dev, rng, lazy=self.lazy_masks)
        b, t, cm = masked_motion.shape
        m = b * t
        d, mf, af = c.hidden_size, c.motion_f, c.audio_f
        nf, nc = spec.N_FACE_LAYERS, spec.N_CROSS_LAYERS
        audio = audio.to(device=dev, dtype=torch.float32).contiguous()
        motion3 = masked_motion.to(device=dev, dtype=torch.float32).contiguous()
        mask3 = mask.to(device=dev, dtype=torch.float32).contiguous()

        # masked motion -> spatial hints (M:267-273)
        x0 = ops.pack_motion(cx.dt, motion3, mask3, pk.w["mask_emb"], _rup(cm))
        if self.tape is not None:
            def bw_mask_embedding():                                                 # x0 = mask ? mask_embedding : motion (M:267-268)
                g = self.tape.get(x0)
                if g is not None:
                    self._param_grad("mask_embedding", slice(None), ops.col_sum(g[:, :cm], mask3.view(m, cm)))
            self.tape.node(bw_mask_embedding)
        hint = self._motion_encoder(cx, x0, t, b, cm)
        hh = self._lin(cx, hint, "bodyhints.fc1", slope=0.1)
        memcat = cx.lo(m, af + mf)                                                   # [audio2face | body_hint_face] (M:288)
        hh_face, hh_body = hh[:, :d], hh[:, d:]
        hint_face = memcat[:, af:]
        if self.tape is not None:
            self.tape.view(hh, hh_face, 0)
            self.tape.view(hh, hh_body, d)
            self.tape.view(memcat, hint_face, af, written=True)
        self._lin(cx, hh_face, "bodyhints_face.fc2", out=hint_face)
        hint_body = self._lin(cx, hh_body, "bodyhints_body.fc2")

        # the two WavEncoders, batch statistics (M:275-281), block by block in lock step; shared by the forwards of a step
        self._shared = share
        if share is not None and share.feats is not None:
            a_face, a_body, ta = share.feats
            self._replay_bn(share.bn_log, new_stats)
        else:
            main = self.tape
            if share is not None:
                self._bn_log = share.bn_log
                if main is not None:
                    self.tape = share.tape = _Tape(dev)      # the encoders' backward waits for the gradients of all the step's forwards
            try:
                (a_face, a_body), ta = self._wav_encoders(cx, [("audio_encoder_face", 0), ("audio_encoder_body", 1)], audio, b, new_stats)
            finally:
                self.tape, self._bn_log = main, None
            if share is not None:
                share.feats = (a_face, a_body, ta)
        if ta < t:
            raise RuntimeError(f"Sizes of tensors must match: audio features {ta} frames vs motion {t} frames")
        memcat[:, :af] = a_face.view(b, ta, af)[:, :t].reshape(m, af)                # M:278-281: the FACE features are trimmed to T
        if self.tape is not None:
            def bw_face_features():
                g = self.tape.get(memcat)
                if g is None:
                    return
                ga = torch.zeros(b, ta, af, dtype=torch.float32, device=dev)
                ga[:, :t] = g[:, :af].reshape(b, t, af)
                self.tape.add(a_face, ga.view(b * ta, af))
            self.tape.node(bw_face_features)

        sid = speaker_id.to(dev).reshape(b, 1).expand(b, t)
        spk_body = self._speaker(cx, "spk_body", "speaker_embedding_body.weight", sid)
        spk_face = self._speaker(cx, "spk_face", "speaker_embedding_face.weight", sid)
        out = {}

        # face branch (M:288-294)
        mem_face = self._lin(cx, memcat, "audio_face_motion_proj")
        face = self._ppe(cx, masks, spk_face, b, t)
        fmem = self._memory(cx, "face.kv_all", mem_face, b, t, nf)
        for i in range(nf):
            face = self._decoder_layer(cx, masks, f"face_motion_decoder.layers.{i}", face, b, t, fmem[i], nf * d, t)
        out["rec_face"] = self._lin(cx, face, "face_out_proj")
        out["cls_face"] = self._lin(cx, self._lin(cx, out["rec_face"], "face_cls.fc1", slope=0.1), "face_cls.fc2")

        # body branch (M:297-312)
        x = self._ppe(cx, masks, self._lin(cx, hint_body, "moton_proj"), b, t)
        x = self._encoder_layer(cx, masks, "motion_self_encoder.layers.0", self._add(cx, x, spk_body), b, t)
        mem_body = self._lin(cx, a_body, "audio_body_motion_proj")                   # M:303
        base = self._ppe(cx, masks, self._add(cx, x, spk_body), b, t)               # M:304-305
        x = base
        bmem = self._memory(cx, "cross.kv_all", mem_body, b, ta, nc)
        for i in range(nc):
            x = self._decoder_layer(cx, masks, f"audio_motion_cross_attn.layers.{i}", x, b, t, bmem[i], nc * d, ta)
        fea = self._add(cx, base, x) if use_audio else base                          # motion_fea + cross; cross * 0 without audio (M:310-312)

        # part latents, refinement layers, heads (M:315-330)
        parts = ("upper", "hands", "lower")
        others = {"upper": ("hands", "lower"), "hands": ("upper", "lower"), "lower": ("upper", "hands")}
        hl = self._lin(cx, fea, "motion2latent.fc1", slope=0.1)
        lat = {}
        for i, p in enumerate(parts):
            hv = hl[:, i * d:(i + 1) * d]
            if self.tape is not None:
                self.tape.view(hl, hv, i * d)
            lat[p] = self._lin(cx, hv, f"motion2latent_{p}.fc2")
        refine = {}
        for p in parts:
            tgt = self._add(cx, lat[p], spk_body)
            mem_lo = self._add(cx, lat[others[p][0]], lat[others[p][1]])
            name = f"body_motion_decoder_{p}.layers.0"
            refine[p] = self._decoder_layer(cx, masks, name, tgt, b, t, self._memory(cx, name + ".ca.kv", mem_lo, b, t, 1)[0], d, t)
        for p in parts:
            out[f"rec_{p}"] = self._lin(cx, self._add(cx, lat[p], refine[p]), f"motion_out_proj_{p}")
        for p in parts:
            out[f"cls_{p}"] = self._lin(cx, self._lin(cx, out[f"rec_{p}"], f"motion_cls_{p}.fc1", slope=0.1), f"motion_cls_{p}.fc2")
        if not masks.consumed_all():
            raise RuntimeError(f"dropout_masks: {len(masks.masks)} masks given, the forward consumed {masks.i}")
        self._out2d = out
        return {key: out[key].view(b, t, -1) for key in OUT_KEYS}, new_stats

    # ---- backward through everything behind the motion encoder and the WavEncoders ------------------------------------------------
    def backward(self, index_gt, latent_gt, progress=None):
        """Gradients of `rec_loss + cls_loss` (T:106-130) of the LAST forward (called with tape=True) w.r.t. every trainable parameter
        that takes part in it; they ACCUMULATE in `self.param_grads` (name -> fp32 tensor of the parameter's shape) across the
        forwards of a step, like `loss_all.backward()` in the reference (T:174).  Every contraction (Linear and Conv1d, dX and dW) is
        an emage_gemm launch in exact-fp32 mode on transposed / im2col operands."""
        if self.tape is None:
            raise RuntimeError("backward() needs the forward to be run with tape=True")
        cfg, tape, out = self.model.config, self.tape, self._out2d
        for q in ("upper", "lower", "hands", "face"):
            pred = out[f"rec_{q}"]
            tape.add(pred, ops.mse_loss_grad(pred, latent_gt[q].reshape(pred.shape).to(pred.device), getattr(cfg, "l" + q[0])))
            logits = out[f"cls_{q}"]
            # cf = 0 in the reference's config: the face classifier's loss is evaluated and multiplied by 0 (T:113-128), its
            # parameters receive exact-zero gradients and an Adam state — kept that way
            tape.add(logits, ops.nll_loss_grad(logits, index_gt[q].reshape(-1).contiguous().to(logits.device), getattr(cfg, "c" + q[0])))
        tape.run(progress)
        self.flush_param_grads()
        self.last_run_nodes = tape.pos + 1
        share = self._shared
        if share is not None and share.tape is not None:           # the feature gradients of this forward join the shared encoder pass
            for feat in share.feats[:2]:
                g = tape.get(feat)
                if g is not None:
                    share.tape.add(feat, g)
        self.tape = None
        return self.param_grads

    def finish_shared(self, share, progress=None, base=0):
        """Backward of the step-shared WavEncoder pass from the summed feature gradients (call behind the step's last `backward`)."""
        if share is None or share.tape is None:
            return
        self.tape = share.tape
        try:
            share.tape.run(progress, base=base)
            self.flush_param_grads()
        finally:
            self.tape = None
        share.tape = None

    def backward_from(self, tape, out2d, grad_outputs):
        """Backward of ONE recorded forward from the gradients of its outputs (the autograd bridge of the model classes' train-mode
        forward: torch computes d loss / d output, this runs the rest): returns {parameter name: gradient} of that forward alone."""
        self.tape, saved = tape, self.param_grads
        self.param_grads = {}
        try:
            for key, g in grad_outputs.items():
                if g is not None:
                    t = out2d[key]
                    tape.add(t, g.reshape(t.shape).to(torch.float32).contiguous())
            tape.run()
            return self.param_grads
        finally:
            self._pg_dst, self._pg_src, self._pg_spans, self._pg_bytes = [], [], {}, 0       # (an exception mid-walk: drop what was queued)
            self._pgrads, self.tape = saved, None


# ======================================================================================
# the autograd bridge: `model.train(); out = model(audio, speaker_id, masked_motion, mask); loss.backward()` (T:156-181)
# ======================================================================================
class _TrainFn(torch.autograd.Function):
    """One train-mode forward of an EmageAudioModel as ONE autograd node: forward = `TrainForward.__call__` with a tape, backward =
    the tape run from the gradients torch hands in for the eight outputs.  The trainable parameters are inputs of the node, so
    `.grad` accumulation, optimisers and hooks behave as for the reference module."""

    @staticmethod
    def forward(ctx, fwd, names, audio, speaker_id, masked_motion, mask, use_audio, dropout_masks, rng, *params):
        with torch.no_grad():
            pred, stats = fwd(audio, speaker_id, masked_motion, mask, dropout_masks, use_audio=use_audio, tape=True, rng=rng)
            flat = fwd.model._flat_params()
            for name, v in stats.items():                 # train-mode BatchNorm: the running buffers advance in place, as in torch
                if name in flat:
                    flat[name].copy_(v.to(flat[name].dtype))
        ctx.fwd, ctx.names, ctx.tape, ctx.out2d = fwd, names, fwd.tape, fwd._out2d
        fwd.tape = None
        return tuple(pred[k] for k in OUT_KEYS)

    @staticmethod
    def backward(ctx, *grad_outputs):
        grads = ctx.fwd.backward_from(ctx.tape, ctx.out2d, dict(zip(OUT_KEYS, grad_outputs)))
        ctx.tape = None
        return (None,) * 9 + tuple(grads.get(n) for n in ctx.names)


def train_forward(model, audio, speaker_id, masked_motion, mask, use_audio=True):
    """EmageAudioModel.forward in TRAINING mode (what `model(...)` runs after `model.train()`): differentiable w.r.t. the model's
    parameters through `_TrainFn`.  Dropout masks come from the device generator (seeded by `torch.initial_seed()`, advancing with
    every call) unless `model.dropout_masks_override` holds recorded mask lists (the parity tests), consumed one per call."""
    fwd = model.__dict__.get("_train_fwd")
    if fwd is None:
        fwd = model.__dict__["_train_fwd"] = TrainForward(model)
        model.__dict__["_train_calls"] = 0
    # an optimiser step (or any in-place edit of a parameter / BatchNorm buffer) since the last forward is seen by `model._engine()`
    # itself (`_version_stamp`), in train and in eval mode: the packed operands are rebuilt from the current parameters
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    override = getattr(model, "dropout_masks_override", None)
    masks = override.pop(0) if override else None
    model.__dict__["_train_calls"] += 1
    rng = (int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF, model.__dict__["_train_calls"], 0)
    outs = _TrainFn.apply(fwd, [n for n, _p in named], audio, speaker_id, masked_motion, mask, bool(use_audio), masks, rng, *[p for _n, p in named])
    return dict(zip(OUT_KEYS, outs))


# ======================================================================================
# the three forwards of a step and their six losses (train_emage_audio.py:132-172)
# ======================================================================================
def targets(vq, motion_aa, expressions, trans, foot_contact):
    """Top of train_val_fn (T:146-152): axis-angle -> rot-6D, the frozen VQ-VAEs' code indices and quantised latents, and the
    337-channel motion the model is conditioned on — all on the device (`vq` is the product EmageVQModel)."""
    bs, t, jc = motion_aa.shape
    dev = vq.vq_model_face.device
    aa = motion_aa.to(device=dev, dtype=torch.float32).reshape(bs * t * (jc // 3), 3).contiguous()
    rot6d = ops.axis_angle_to_rot6d(aa).reshape(bs, t, jc // 3 * 6)
    expressions, trans, foot_contact = (x.to(device=dev, dtype=torch.float32) for x in (expressions, trans, foot_contact))
    index = vq.map2index(rot6d, expressions, tar_contact=foot_contact, tar_trans=trans)
    latent = vq.map2latent(rot6d, expressions, tar_contact=foot_contact, tar_trans=trans)
    return index, latent, torch.cat([rot6d, trans, foot_contact], dim=-1)


def losses(cfg, pred, index_gt, latent_gt, workspace=None):
    """(rec, cls) of T:106-130 as float64 device scalars: sum_q l_q * mse(rec_q, latent_q), sum_q c_q * NLL(log_softmax(cls_q), index_q)."""
    dev = pred["rec_face"].device
    ws = ops.loss_workspace(dev) if workspace is None else workspace
    rec, cls = torch.zeros(1, dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.float64, device=dev)
    for q in ("upper", "lower", "hands", "face"):
        b, t, k = pred[f"rec_{q}"].shape
        ops.mse_loss(pred[f"rec_{q}"].reshape(b * t, k), latent_gt[q].reshape(b * t, k), getattr(cfg, "l" + q[0]), rec, ws)
        ops.nll_loss(pred[f"cls_{q}"].reshape(b * t, -1), index_gt[q].reshape(-1).contiguous(), getattr(cfg, "c" + q[0]), cls, ws)
    return rec, cls


def step_losses(fwd: TrainForward, vq, batch, iteration, dropout_masks, random_mask):
    """The seed / audio / mask forwards of one step (T:132-172) -> (dict of the six losses + "all" as Python floats, BatchNorm
    buffers after the three forwards).  dropout_masks: three mask lists (one per forward); random_mask: the (B, T, 337)
    {0, 1} motion mask of forwards 2 and 3, `torch.rand(...) < mask_ratio` in the reference (T:163-165)."""
    cfg = fwd.model.config
    index, latent, masked_motion = targets(vq, batch["motion"], batch["expressions"], batch["trans"], batch["foot_contact"])
    bs, t = masked_motion.shape[:2]
    speaker_id = torch.zeros(bs, 1, dtype=torch.long, device=masked_motion.device)
    seed_mask = torch.ones_like(masked_motion)
    seed_mask[:, :cfg.seed_frames] = 0
    stats, out = {}, {}
    ws = ops.loss_workspace(masked_motion.device)
    for tag, mask, use_audio, masks in (("seed", seed_mask, True, dropout_masks[0]), ("audio", random_mask, True, dropout_masks[1]),
                                        ("mask", random_mask, False, dropout_masks[2])):
        pred, stats = fwd(batch["audio"], speaker_id, masked_motion, mask, masks, use_audio=use_audio, new_stats=stats)
        out["rec_" + tag], out["cls_" + tag] = losses(cfg, pred, index, latent, ws)
    res = {k: float(v) for k, v in out.items()}
    ops.loss_check(ws)
    res["all"] = sum(res.values())
    return res, stats


class Trainer:
    """One optimisation step of train_emage_audio.py:132-180 on the device: the three train-mode forwards with their backward
    passes (gradients accumulate like `loss_all.backward()`), torch.optim.Adam with the reference's settings (lr 1.5e-4 constant,
    betas .9 / .999, eps 1e-8, no weight decay: configs/emage_audio.yaml:63-78), the BatchNorm running buffers.  Parameters that
    take no part in the forward (the deep-copied template layers, M:246-262) get no gradient and stay untouched, as in torch.

    Gradients live in four flat, backward-ordered buckets (`dist.GradientBuckets`: the backward kernels' results are accumulated
    straight into the all-reduce messages, no copies).  In a multi-process run (`group` / an initialised default group) bucket i is
    all-reduced AS SOON AS its last gradient of the step has been written — during the third backward, while the remaining backward
    kernels are still being issued (the overlap DDP gives the reference, T:251): the tape position of every parameter's last
    contribution is learned during the first step (the launch sequence of a step is static) and drives the schedule from the second
    step on.  The 1 / world_size of the average is folded into Adam, which is ONE multi-tensor launch that also clears the gradients.
    Dropout masks: `dropout_masks` (three lists, the parity tests) or None = drawn on the device (`ops.dropout_mask`, seeded by `seed`).
    `grad_hook(param_grads)` still runs between backward and the update (tests spy on it)."""

    def __init__(self, model, vq, lr=1.5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, sync_bn=False, group=None, seed=0, exchange=True,
                 on_nonfinite="raise", share_encoders=True, exchange_dtype=torch.float32):
        """exchange=False: no built-in gradient all-reduce even in a multi-process run (a `grad_hook` may do it: `dist.gradient_allreduce_hook`).

        Health of a step (the f16x3 backward runs on fp16 planes of `grad_scale` x dY and of scaled weights: an overflow turns into
        inf / NaN): `emage_count_nonfinite` over the four gradient buckets runs INSIDE the step (eager and captured), in front of Adam, and
        Adam takes the count as its skip word — a poisoned step never reaches the parameters, the moments or the BatchNorm buffers.  The
        count travels to the host with the losses; on_nonfinite = "raise": FloatingPointError (the state is that of the previous step);
        "skip": the step is dropped (`skipped_steps`), `grad_scale` is halved and training goes on (a captured step is re-captured with
        the new scale: it is a launch argument) — the step-skip of loss-scaled mixed-precision training.  Weight operand scales are
        power-of-two constants chosen once; every re-packing checks on the device that max |w| x scale is still in [2^10, 2^14) and the
        trainer re-derives the scales behind the step that reports a miss (two doublings before an fp16 plane could overflow)."""
        from . import dist as pdist
        if on_nonfinite not in ("raise", "skip"):
            raise ValueError("on_nonfinite must be 'raise' or 'skip'")
        self.on_nonfinite, self.skipped_steps, self.rescaled = on_nonfinite, 0, 0
        self.nonfinite_loss_steps = 0                     # steps whose loss was inf / NaN although every gradient was finite (update applied; see `_finish`)
        # share_encoders: the WavEncoder pass of a step is computed (and differentiated) once for its three forwards — see `StepShare`;
        # False = the reference's schedule, three passes (same losses; gradients equal up to fp32 summation order)
        self.share_encoders = bool(share_encoders)
        self.exchange = bool(exchange)
        self.fwd, self.vq = TrainForward(model, sync_bn=sync_bn, group=group), vq
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.group, self.seed = group, int(seed)
        self.state = {}                                   # name -> dict(step, exp_avg, exp_avg_sq)
        named = [(k, v) for k, v in model.named_parameters() if v.requires_grad]
        plan, self.unused = pdist.emage_bucket_plan(named)
        self.buckets = pdist.GradientBuckets(plan, device=model.device, group=group, exchange_dtype=exchange_dtype)       # bfloat16: half the bytes on the wire
        self.fwd.grad_views = self.buckets.grads
        self.bucket_of = {name: i for i, (_tag, params) in enumerate(plan) for name, _p in params}
        self.schedule = None                              # bucket -> tape position (third backward) behind which it is complete
        self.exchange_log = []                            # (event, bucket or tape position) of the last step: what the overlap tests read
        self.steps_done = 0
        self._adam = None
        self._graph, self._recapture_pending = None, False
        self.health = torch.zeros(1, dtype=torch.int32, device=model.device)       # non-finite gradient words of the last step (Adam's skip word)

    def _world(self):
        import torch.distributed as tdist
        if not self.exchange:
            return 1
        return tdist.get_world_size(self.group) if (tdist.is_available() and tdist.is_initialized()) else 1

    def _exchanging(self):
        """True when the step runs its gradient exchange: a process group exists (and exchange=True) — ALSO at world size 1, like
        DistributedDataParallel: the four bucket all-reduces then go through the backend (RCCL on one MI355X: the only way to run the
        exchange on hardware without a multi-GPU node, tests/test_train_forward_gpu.py) and are the identity."""
        import torch.distributed as tdist
        return self.exchange and tdist.is_available() and tdist.is_initialized()

    def _rng(self, forward_index, step):
        return (self.seed, step, 128 * forward_index)

    def _device_step(self, batch, dropout_masks, random_mask, grad_hook=None, step_counter=None):
        """Everything of a step that runs on the device; returns (dict of float64 device loss scalars, loss workspace).  With
        `step_counter` (one int32 on the device) Adam and the mask generator read the step count from it — what a captured graph needs."""
        fwd, model = self.fwd, self.fwd.model
        cfg = model.config
        index, latent, masked_motion = targets(self.vq, batch["motion"], batch["expressions"], batch["trans"], batch["foot_contact"])
        bs = masked_motion.shape[0]
        speaker_id = torch.zeros(bs, 1, dtype=torch.long, device=masked_motion.device)
        seed_mask = torch.ones_like(masked_motion)
        seed_mask[:, :cfg.seed_frames] = 0
        fwd.param_grads = {}
        fwd.begin_step()
        fwd._wt_cache, fwd._wt_keep = {}, True            # the parameters move at the END of the step: one set of transposed weight images
        try:
            return self._device_step_body(batch, dropout_masks, random_mask, grad_hook, step_counter, index, latent, masked_motion, speaker_id, seed_mask)
        except BaseException:
            fwd._pg_dst, fwd._pg_src, fwd._pg_spans, fwd._pg_bytes = [], [], {}, 0       # a step that died mid-backward: its queued contributions die with it
            fwd._fin.entries = []                         # ... and its queued finalize launches (their float64 partials would land in the NEXT step's buckets)
            for flat in self.buckets.flat:                # Adam's zero_grad never ran for the dead step
                flat.zero_()
            raise
        finally:
            fwd._wt_cache, fwd._wt_keep = {}, False

    def _device_step_body(self, batch, dropout_masks, random_mask, grad_hook, step_counter, index, latent, masked_motion, speaker_id, seed_mask):
        fwd, model = self.fwd, self.fwd.model
        cfg = model.config
        stats, out = {}, {}
        ws = ops.loss_workspace(masked_motion.device)
        world, exchanging = self._world(), self._exchanging()
        buckets, log = self.buckets, []
        self.exchange_log = log
        learning = self.schedule is None
        step_for_rng = step_counter if step_counter is not None else self.steps_done + 1
        dm = dropout_masks if dropout_masks is not None else (None, None, None)
        share = StepShare() if self.share_encoders else None
        for f, (tag, mask, use_audio, masks) in enumerate((("seed", seed_mask, True, dm[0]), ("audio", random_mask, True, dm[1]),
                                                            ("mask", random_mask, False, dm[2]))):
            pred, stats = fwd(batch["audio"], speaker_id, masked_motion, mask, masks, use_audio=use_audio, new_stats=stats, tape=True,
                              rng=self._rng(f, step_for_rng), share=share)
            out["rec_" + tag], out["cls_" + tag] = losses(cfg, pred, index, latent, ws)
            progress = None
            if f == 2:
                if learning:
                    fwd.touch = {}
                elif exchanging:
                    ready = {}
                    for i, pos in self.schedule.items():
                        ready.setdefault(pos, []).append(i)

                    def progress(pos, ready=ready):
                        for i in ready.get(pos, ()):         # every gradient of bucket i is final: its all-reduce overlaps the rest of the backward
                            fwd.flush_param_grads()          # ... once its queued contributions have been applied
                            log.append(("reduce", i, pos))
                            buckets.reduce(i)
            fwd.backward(index, latent, progress)
            if f == 2:                                    # the shared WavEncoder pass: the tail of the third backward
                fwd.finish_shared(share, progress, base=fwd.last_run_nodes)
            log.append(("backward_done", f))
        # non-finite LOSS words of this rank — in an exchanging run the MAXIMUM over ranks (one more small all-reduce; a graph node in a captured
        # step): a loss is rank-local, so without it `on_nonfinite="raise"` would raise on ONE rank and leave the others in their next collective
        self.loss_health = (~torch.isfinite(torch.stack([v.reshape(()) for v in out.values()]))).sum().to(torch.int32).reshape(1)
        if exchanging and world > 1:
            import torch.distributed as tdist
            tdist.all_reduce(self.loss_health, op=tdist.ReduceOp.MAX, group=self.group)
        if learning:
            last = {}
            for name, pos in fwd.touch.items():
                i = self.bucket_of[name]
                last[i] = max(last.get(i, -1), pos)
            self.schedule = {i: last.get(i, -1) for i in range(len(buckets.flat))}      # -1: complete before the third backward starts
            fwd.touch = None
            if exchanging:
                for i in range(len(buckets.flat)):
                    log.append(("reduce", i, None))
                    buckets.reduce(i)
        grads = fwd.param_grads
        if exchanging:
            buckets.wait(average=False)                   # the 1 / world of the average is Adam's grad_scale
            log.append(("wait",))
        if grad_hook is not None:
            grad_hook(grads)
        fwd.flush_range_checks(new_step=True)
        # health: inf / NaN among the (exchanged) gradients -> Adam's skip word.  After the exchange, so every rank sees the same count
        self.health.zero_()
        for flat in buckets.flat:
            ops.count_nonfinite(flat, self.health)
        params = model._flat_params()                     # detached views of the nn.Parameters: updated in place
        if self._adam is None:
            quads = []
            for name, g in buckets.grads.items():
                p = params[name]
                st = self.state[name] = dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
                quads.append((p, g, st["exp_avg"], st["exp_avg_sq"]))
            self._adam = ops.AdamTable(quads, model.device)
        self.steps_done += 1
        for st in self.state.values():
            st["step"] = self.steps_done
        ops.adam_multi(self._adam, self.steps_done if step_counter is None else step_counter, self.lr, self.betas[0], self.betas[1], self.eps,
                       self.weight_decay, grad_scale=1.0 / world, zero_grad=True, skip=self.health)
        skipped = self.health[0] > 0                      # device scalar: the buffers below keep their values in a skipped step
        by_dtype = {}
        for name, v in stats.items():                     # BatchNorm running statistics after the three forwards
            if name in params:
                by_dtype.setdefault(params[name].dtype, []).append((params[name], v.to(device=params[name].device, dtype=params[name].dtype)))
        for pairs in by_dtype.values():                   # per dtype: gather, ONE select on the skip word, ONE multi-tensor copy back (96 buffers)
            olds, news = [p for p, _v in pairs], [v for _p, v in pairs]
            sel = torch.where(skipped, torch.cat([p.reshape(-1) for p in olds]), torch.cat([v.reshape(-1) for v in news]))
            torch._foreach_copy_(olds, [c.view(p.shape) for c, p in zip(sel.split([p.numel() for p in olds]), olds)])
        if step_counter is not None:                      # a skipped step does not count (Adam's bias correction, the mask generator's step)
            step_counter.sub_(skipped.to(step_counter.dtype))
        return out, ws

    def _deferred_range_check(self):
        """Context: `model._engine()` leaves the operand-scale flags of a re-packing unread (no host synchronisation in front of a step);
        `_finish` reads them with the losses."""
        import contextlib
        model = self.fwd.model

        @contextlib.contextmanager
        def cm():
            model.__dict__["_defer_range_check"] = True
            try:
                yield
            finally:
                model.__dict__["_defer_range_check"] = False
        return cm()

    def _range_flags(self):
        """The device counters of weight operands whose cached power-of-two scale no longer fits (forward packing, backward operands)."""
        pk = self.fwd.model._packed
        return [f for f in (self.fwd.range_flag, pk.range_flag if pk is not None else None) if f is not None]

    def _finish(self, out, ws):
        """Host side of a step: losses, the health word, the range flags (one synchronisation) -> loss dict; raises / skips / re-derives
        scales as `__init__` describes.  Returns (losses, recapture needed)."""
        res = {k: float(v) for k, v in out.items()}
        bad = int(self.health[0])
        stale = any(int(f) != 0 for f in self._range_flags())
        ops.loss_check(ws)
        res["all"] = sum(res.values())
        recapture = False
        if stale:                                         # results of this step are fine (margin of 4x); the NEXT packing takes fresh scales
            self.rescaled += 1
            self.fwd.reset_scales()
            recapture = self._recapture_pending = True
        finite = all(v == v and abs(v) != float("inf") for v in res.values())
        lh = getattr(self, "loss_health", None)
        if lh is not None and int(lh[0]) != 0:            # some rank's loss was non-finite: every rank takes the branch below together
            finite = False
        if bad:                                           # the DEVICE skipped the update (Adam's skip word, the same on every rank: it is counted
            self.steps_done -= 1                          # behind the gradient exchange) — the host only follows that decision
            for st in self.state.values():
                st["step"] = self.steps_done
            msg = (f"training step {self.steps_done + 1}: {bad} non-finite gradient words (losses {res}) — the update was skipped on the device, "
                   f"parameters / Adam moments / BatchNorm buffers are those of the previous step (precision {self.fwd.model.precision!r}, "
                   f"grad_scale {self.fwd.grad_scale:g}: an fp16 plane of the split-fp16 backward overflowed, or the forward met an activation beyond "
                   "|x| < 4094); lower the loss scale (on_nonfinite='skip' halves it) or train with set_precision('fp32')")
            if self.on_nonfinite == "raise":
                raise FloatingPointError(msg)
            self.skipped_steps += 1
            self.fwd.grad_scale *= 0.5
            recapture = self._recapture_pending = True
        elif not finite:
            # a non-finite LOSS whose gradients were all finite: the device applied the update — the health word decides, and
            # it is the only thing all ranks share (losses are rank-local).  Nothing is rolled back, re-scaled or re-captured: host and device
            # step counts stay equal, and no rank starts a warm-up step (with collectives) the others do not run.
            self.nonfinite_loss_steps += 1
            if self.on_nonfinite == "raise":
                raise FloatingPointError(f"training step {self.steps_done}: non-finite loss {res} with finite gradients — the update WAS applied "
                                         "(the device's health word counts gradient words only); optimiser state and step counts are consistent")
        return res, recapture

    def step(self, batch, iteration=0, dropout_masks=None, random_mask=None, grad_hook=None):
        """One optimisation step -> dict of the six losses + "all".  `iteration` is accepted for signature compatibility with the
        reference's train_val_fn (T:132) and unused: the caller computes the mask ratio and passes `random_mask` (T:163-165)."""
        with self._deferred_range_check():                # the operand-scale flags are read HERE with the losses, not at packing time
            out, ws = self._device_step(batch, dropout_masks, random_mask, grad_hook)
        try:
            res, _ = self._finish(out, ws)
        finally:
            self._parameters_moved()                      # the MFMA operand copies are rebuilt from the updated parameters
            if self._graph is not None:                   # hand-over from the captured step (a ragged last batch runs eagerly): the graph's
                self._step_counter.fill_(self.steps_done) # device-side step counter follows, so a later replay() continues the same count
        return res

    def _parameters_moved(self):
        """`emage_adam_multi` (and a graph replay) update the parameters through raw device pointers: advance their version counters so
        the staleness stamp of `_engine()` sees it (correctness no longer hangs on the call below), and drop the packed set."""
        model = self.fwd.model
        model.bump_versions([p for p in model.parameters() if p.requires_grad])
        model.invalidate_packed()
        self.fwd._pcache = None

    # ---- the step as ONE hipGraph -------------------------------------------------------------------------------------------------
    def capture(self, batch, random_mask, dropout_masks=None):
        """Capture the whole step — re-packing the MFMA operands from the current parameters, targets, three forwards with their
        backward passes, the dropout masks (drawn on the device from the in-graph step counter unless `dropout_masks` buffers are
        given), Adam, BatchNorm buffers — into one hipGraph over the GIVEN tensors: `batch` and `random_mask` (and `dropout_masks`)
        become the graph's input buffers (refill them in place between replays), the parameters its state.  `replay()` then runs a
        step at device speed instead of the ~10^4 Python-level launches of `step()`.  Both fp32-storage precisions: in f16x3 the
        split-fp16 operand scales are the ones chosen by the warm-up step's packing (a re-packing with known scales is a pure sequence
        of launches).  One eager warm-up step is run and undone.

        Multi-process runs (round 5): the step is captured WITH its collectives — the four bucket all-reduces (started mid-backward on the
        backend's own stream: they become parallel branches of the graph), SyncBatchNorm's all-gathers and small all-reduces — when the
        backend is "nccl" (= RCCL, whose kernels are stream-capturable), so a multi-rank step is the same graph replay as a single-rank
        one instead of ~10^4 eager launches.  Contract: every rank calls `capture()` (its warm-up step runs the collectives eagerly) and
        then `replay()` the same number of times, with the batch sizes it captured (SyncBatchNorm's global count is the warm-up step's)."""
        fwd, model = self.fwd, self.fwd.model
        if fwd.sync_bn or self._exchanging():
            import torch.distributed as tdist
            backend = tdist.get_backend(self.group) if (tdist.is_available() and tdist.is_initialized()) else None
            if backend is not None and str(backend) != "nccl":
                raise RuntimeError(f"Trainer.capture: the step's collectives can only be captured with the nccl (RCCL) backend, not {backend!r}; "
                                   "use Trainer.step")
        dev = model.device
        for name, t in list(batch.items()) + [("random_mask", random_mask)]:
            if not (torch.is_tensor(t) and t.is_cuda and t.device == dev and t.dtype == torch.float32):
                raise RuntimeError(f"Trainer.capture: {name} must be a float32 tensor on {dev} (it becomes an input buffer of the graph; a host "
                                   "tensor would be copied once at capture time and never again)")
        for fm in (dropout_masks or ()):
            for mk in fm:
                if not (mk.is_cuda and mk.dtype == torch.float32 and mk.is_contiguous()):
                    raise RuntimeError("Trainer.capture: dropout masks must be contiguous fp32 tensors on the device (they are the graph's input buffers)")
        params = model._flat_params()
        saved = {k: v.clone() for k, v in params.items()}
        done = self.steps_done                                         # a RE-capture (new loss scale / operand scales) keeps the optimiser state
        moments = {k: (st["exp_avg"].clone(), st["exp_avg_sq"].clone()) for k, st in self.state.items()}
        self._graph = None
        with self._deferred_range_check():
            self._device_step(batch, dropout_masks, random_mask)      # warm-up: lazy initialisation inside the library / the allocator, the exchange schedule
        torch.cuda.synchronize(dev)
        for k, v in saved.items():
            params[k].copy_(v)
        for k, st in self.state.items():                               # the moments live OUTSIDE the graph's memory (created by the first warm-up step)
            if k in moments:
                st["exp_avg"].copy_(moments[k][0])
                st["exp_avg_sq"].copy_(moments[k][1])
            else:
                st["exp_avg"].zero_()
                st["exp_avg_sq"].zero_()
            st["step"] = done
        self.steps_done = done
        self._step_counter = torch.full((1,), done, dtype=torch.int32, device=dev)
        self._graph_inputs = (batch, random_mask, dropout_masks)       # the graph reads these buffers at every replay: keep them alive
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            model.invalidate_packed()                                  # so that the packing of the current parameters is part of the graph
            fwd._pcache = None
            self._step_counter.add_(1)
            self._graph_out, self._graph_ws = self._device_step(batch, dropout_masks, random_mask, step_counter=self._step_counter)
        self._graph, self._recapture_pending = graph, False
        self._graph_operands = model._packed                           # what the captured launches read (and re-write at every replay)
        model.invalidate_packed()                                      # ... and nobody else: they hold the weights BEFORE the replay's update
        self.steps_done = done
        for st in self.state.values():
            st["step"] = done
        return self

    def replay(self):
        """One captured step on the current contents of the captured input buffers -> the loss dict of `step()`.  A step whose health word
        or operand-scale flags call for it (see `__init__`) is followed by a re-capture over the same input buffers."""
        if self._graph is None:
            raise RuntimeError("Trainer.replay: capture() first")
        if self._recapture_pending:                       # a step that raised behind a scale reset: the graph still carries the old scales
            self.capture(*self._graph_inputs)
        self._graph.replay()
        self.steps_done = int(self._step_counter) + (1 if int(self.health[0]) else 0)      # _finish() takes a skipped step back off
        for st in self.state.values():
            st["step"] = self.steps_done
        model = self.fwd.model
        model._packed = self._graph_operands              # the flags of the graph's own packing
        try:
            res, recapture = self._finish(self._graph_out, self._graph_ws)
        finally:
            self._parameters_moved()                      # a replay moves the parameters through raw pointers: an eval forward behind it
        if recapture:                                     # must re-pack (and never reads the graph's operand set)
            self.capture(*self._graph_inputs)
        return res

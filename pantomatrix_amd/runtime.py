"""Clip-batch runner: the whole `wav -> SMPL-X` hot path of test_emage_audio.py:16-53 as ONE hipGraph.

The reference drives ~600 small launches per 128-frame batch from Python; on MI355X the per-launch host cost
(~8 us through any Python binding) would dominate the kernels themselves.  The launch sequence of a batch is
static — it depends only on (batch, audio length, precision) — so it is captured once into a HIP graph
(stream capture via torch.cuda.CUDAGraph; every kernel is launched on torch's current stream, so the C-ABI
launches are captured like any other) and replayed per batch: `inference()` (sequential 64-frame windows, seed
frames carried through the VQ decode) + the final `EmageVQModel.decode(get_global_motion=True)`.
Outputs land in pinned host buffers with one synchronisation.

`sub_batches = k` splits the batch into k independent groups of clips, each with its own captured graph, replayed
on k streams at once: the launch chain of one window is ~300 dependent kernels of a few microseconds each, and a
second independent chain fills the ramp / drain gaps of the first.
"""
from __future__ import annotations

import torch

from . import ops


RESCALE_STEP = 4          # on_overflow="rescale": the twin's activation images hold x instead of 16 x (16x the range, 4 bits off the smallest values' lo plane)


def calibrate_activation_shift(model, vq_model, audio, speaker_id=None, margin: int = 1, max_shift: int = 8):
    """Choose the EMAGE_H2 activation shift of a checkpoint from a calibration batch: the smallest k at which `audio` (B, samples) runs the
    split-fp16 path without a non-finite value (the on-device health check of `ClipRunner`), plus `margin` bits of headroom when k > 0;
    k = 0 — the parity-green default — is kept whenever it suffices.  Sets the shift on both models and returns it; raises if even
    `max_shift` overflows (run such a checkpoint in fp32)."""
    audio = torch.as_tensor(audio, dtype=torch.float32)
    if audio.dim() == 1:
        audio = audio[None]
    for k in range(0, max_shift + 1):
        runner = ClipRunner(model, vq_model, audio.shape[0], audio.shape[1], use_graph=False, warmup=1, activation_shift=k)
        try:
            runner(audio.to(model.device), speaker_id)
        except FloatingPointError:
            continue
        k = min(k + (margin if k else 0), ops.MAX_ACT_SHIFT)
        model.set_activation_shift(k)
        vq_model.set_activation_shift(k)
        return k
    raise FloatingPointError(f"calibrate_activation_shift: non-finite values up to shift {max_shift} — run this checkpoint with set_precision('fp32')")


class ClipRunner:
    def __init__(self, model, vq_model, batch: int, n_samples: int, use_graph: bool = True, warmup: int = 2,
                 sub_batches: int = 1, main_priority: bool = False, on_overflow: str = "raise", split_k: bool = False,
                 activation_shift: int | None = None):
        """main_priority (experiment): capture on a HIGH-priority stream, so the launch chain issued on lane 0 (the critical
        path: motion encoder -> body stack -> decode) outranks the side lanes (face decoder, WavEncoders) when both have
        ready kernels — if the runtime's graph kernel nodes inherit the capturing stream's priority.
        on_overflow: what `__call__` does with a batch whose health counter is non-zero (in f16x3: an activation beyond the split-fp16
        range) — "raise" (FloatingPointError) or "fp32": run THAT batch again through an exact-fp32 twin of this runner (built on
        first use: +26 ms per affected batch instead of an error; `fallbacks` counts them); a batch that is non-finite in fp32 too raises;
        or "rescale" (f16x3): run that batch again on the SAME split-fp16 path with the activation images scaled down by 2^RESCALE_STEP
        (`activation_shift` + 4: |x| < 65 504 instead of 4 094; a twin runner built on first use, `rescales` counts them) and only a batch
        that overflows there too goes to the fp32 twin — a checkpoint with large pre-norm sums stays on the fast path.
        activation_shift: the EMAGE_H2 activation shift THIS runner's launches use (None: the model's current `activation_shift`)."""
        if on_overflow not in ("raise", "fp32", "rescale"):
            raise ValueError("on_overflow must be 'raise', 'fp32' or 'rescale'")
        self.model, self.vq = model, vq_model
        self.on_overflow, self.fallbacks, self._fp32_twin = on_overflow, 0, None
        self.rescales, self._rescaled_twin = 0, None
        self.shift = int(model.activation_shift if activation_shift is None else activation_shift)
        ops.h2_shifted(self.shift)
        self.precision = model.precision             # what THIS runner's launches compute in (the models may be re-packed later)
        self._args = dict(batch=batch, n_samples=n_samples, use_graph=use_graph, warmup=warmup, split_k=split_k)
        dev = model.device
        if dev.type != "cuda":
            raise RuntimeError("ClipRunner needs the models on an MI355X device")
        if batch % sub_batches:
            raise ValueError("batch must be divisible by sub_batches")
        if sub_batches > 1 and on_overflow != "raise":
            raise ValueError("on_overflow='fp32' needs sub_batches=1 (the exact-fp32 re-run is built per single-graph runner)")
        self.device, self.batch, self.sub = dev, batch, sub_batches
        self.children, self.streams = [], []
        if sub_batches > 1:
            self.children = [ClipRunner(model, vq_model, batch // sub_batches, n_samples, use_graph, warmup, 1, activation_shift=self.shift)
                             for _ in range(sub_batches)]
            self.streams = [torch.cuda.Stream(device=dev) for _ in range(sub_batches)]
            self.frames_out = self.children[0].frames_out
            self.host = tuple(torch.empty((batch,) + tuple(h.shape[1:]), dtype=h.dtype, pin_memory=True)
                              for h in self.children[0].host)
            return
        self.audio = torch.zeros(batch, n_samples, dtype=torch.float32, device=dev)
        self.speaker_id = torch.zeros(batch, 1, dtype=torch.long, device=dev)
        self.ref_trans = torch.zeros(1, 3, device=dev)
        self.nonfinite = torch.zeros(1, dtype=torch.int32, device=dev)
        self.nonfinite_host = torch.zeros(1, dtype=torch.int32, pin_memory=True)
        # split_k (round 6; OFF by default — measured SLOWER: one clip 4.40 vs 4.18 ms, 28 s 28.9 vs 27.2, four clips 4.87 vs 4.32, eight 5.20 vs 4.72,
        # profiles/r06_one_clip_split_k_ab.txt): few clips (<= 8: at most 512 rows per contraction) — the launches of a window occupy a few CUs
        # each and walk their K range alone; lent scratch (ops.SplitKScratch; this runner's own, one pair per stream lane) they split K inside
        # the launch.  The fix-up's own chain (write-through stores -> counter -> re-read of the slices: three trips to memory) costs what the
        # shorter K-loop saves, and the lanes already run the few-block launches side by side
        self.splitk = ops.SplitKScratch() if (split_k and batch * 64 <= ops.SPLITK_MAX_ROWS and model.precision == "f16x3" and getattr(model, "split_acts", False)) else None
        self.graph = None
        for _ in range(max(1, warmup)):            # packs weights, warms the allocator, validates shapes
            out = self._step()
        torch.cuda.synchronize(dev)
        if use_graph:
            self.graph = torch.cuda.CUDAGraph()
            kw = dict(stream=torch.cuda.Stream(device=dev, priority=-1)) if main_priority else {}
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local", **kw):   # other threads (an RCCL watchdog) may touch the runtime meanwhile
                out = self._step()
        self.out = out
        # the captured launches read the packed operand sets built by the warm-up: keep them alive for the life of the graph even if
        # the models are re-packed later (set_precision / load_state_dict / a training step)
        self._operands = (model._packed, [m._packed for m in vq_model._models()], dict(model._templates), dict(vq_model._templates))
        self.host = tuple(torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in out)
        self.frames_out = int(out[0].shape[1])
        self._checked_replays = 0

    def _step(self):
        # health check: inf / NaN among the network outputs the codes are taken from (the quantiser would launder them into valid
        # codes) and among the results — e.g. an activation beyond the f16x3 range
        self.nonfinite.zero_()
        self.model.health_pending = pending = []
        was = (self.model.activation_shift, self.vq.activation_shift)
        self.model.set_activation_shift(self.shift)          # this runner's own (a rescaled twin differs from the models' setting)
        self.vq.set_activation_shift(self.shift)
        try:
            with ops.splitk_scope(self.splitk):
                try:
                    codes = self.model.infer_codes(self.audio, self.speaker_id, self.vq)
                finally:
                    self.model.health_pending = None
                pred = self.vq.decode(**codes, get_global_motion=True, ref_trans=self.ref_trans)
        finally:
            self.model.set_activation_shift(was[0])
            self.vq.set_activation_shift(was[1])
        out = pred["motion_axis_angle"], pred["expression"], pred["trans"]
        ops.count_nonfinite_multi(pending + [t.contiguous() for t in out], self.nonfinite)      # ONE launch (round 5: 11)
        return out

    def run_device(self, audio=None, speaker_id=None):
        """Launch one batch on the current stream; returns device tensors (poses (B,T,165), expressions (B,T,100),
        trans (B,T,3)) that are overwritten by the next call."""
        assert self.sub == 1
        if audio is not None:
            self.audio.copy_(audio, non_blocking=True)
        if speaker_id is not None:
            self.speaker_id.copy_(speaker_id, non_blocking=True)
        if self.graph is not None:
            self.graph.replay()
        else:
            self.out = self._step()
        return self.out

    def _to_host(self, host_views):
        for h, d in zip(host_views, self.out):
            h.copy_(d, non_blocking=True)

    def __call__(self, audio=None, speaker_id=None):
        """One batch end to end, results as numpy arrays on the host (the arrays `beat_format_save` receives)."""
        if self.sub == 1:
            self.run_device(audio, speaker_id)
            self._to_host(self.host)
            self.nonfinite_host.copy_(self.nonfinite, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
            if int(self.nonfinite_host[0]) and self.on_overflow == "rescale" and self.precision == "f16x3" and self.shift + RESCALE_STEP <= ops.MAX_ACT_SHIFT:
                return self._run_rescaled()
            if int(self.nonfinite_host[0]) and self.on_overflow in ("fp32", "rescale") and self.precision != "fp32":
                return self._run_in_fp32()
            self._raise_if_nonfinite()
            return tuple(h.numpy() for h in self.host)
        main = torch.cuda.current_stream(self.device)
        start = torch.cuda.Event()
        start.record(main)
        n = self.batch // self.sub
        for i, (child, s) in enumerate(zip(self.children, self.streams)):
            s.wait_event(start)
            with torch.cuda.stream(s):
                child.run_device(None if audio is None else audio[i * n:(i + 1) * n],
                                 None if speaker_id is None else speaker_id[i * n:(i + 1) * n])
                child._to_host(tuple(h[i * n:(i + 1) * n] for h in self.host))
                child.nonfinite_host.copy_(child.nonfinite, non_blocking=True)
        for s in self.streams:
            s.synchronize()
        for child in self.children:
            child._raise_if_nonfinite()
        return tuple(h.numpy() for h in self.host)

    def _run_rescaled(self):
        """The batch in `self.audio` / `self.speaker_id` again through a twin of this runner whose activation images are scaled down by
        2^RESCALE_STEP (same packed weights, its own graph, built at the first overflow); a batch that overflows there too leaves the
        split-fp16 path through the twin's own fp32 re-run."""
        self.rescales += 1
        if self._rescaled_twin is None:
            self._rescaled_twin = ClipRunner(self.model, self.vq, on_overflow="fp32", activation_shift=self.shift + RESCALE_STEP, **self._args)
        out = self._rescaled_twin(self.audio, self.speaker_id)
        self.fallbacks += self._rescaled_twin.fallbacks - getattr(self, "_twin_fallbacks_seen", 0)
        self._twin_fallbacks_seen = self._rescaled_twin.fallbacks
        return out

    def _run_in_fp32(self):
        """The batch in `self.audio` / `self.speaker_id` again through an exact-fp32 twin runner (same weights; its own packed operands and
        graph, built at the first overflow); the models' precision setting is restored afterwards."""
        self.fallbacks += 1
        if self._fp32_twin is None:
            was = (self.model.precision, self.vq.precision)
            self.model.set_precision("fp32")
            self.vq.set_precision("fp32")
            try:
                self._fp32_twin = ClipRunner(self.model, self.vq, on_overflow="raise", **self._args)
            finally:
                self.model.set_precision(was[0])
                self.vq.set_precision(was[1])
        return self._fp32_twin(self.audio, self.speaker_id)

    def _raise_if_nonfinite(self):
        n = int(self.nonfinite_host[0])
        if n:
            raise FloatingPointError(f"{n} non-finite values among the network outputs / the generated motion (precision {self.precision!r}): in f16x3 an activation beyond "
                                     "|x| < 4094 overflows the fp16 planes — run this checkpoint with set_precision('fp32')")


class ClipPipeline:
    """`depth` ClipRunners (own buffers, own captured graph, shared read-only weights) on `depth` streams: batch i is launched on
    slot i % depth while earlier batches are still in flight.  A batch is a chain of ~300 DEPENDENT launches (the autoregressive
    windows; the cross-attention stack waits for the waveform features), each with a ramp and a drain during which most of the
    chip idles; consecutive batches are independent, so the next batch's kernels fill those gaps.  Results are returned in
    submission order; every batch is complete (poses / expressions / translation in pinned host buffers, health counter checked)
    when `collect()` hands it out.

    Host buffers: `depth + 1` sets in rotation, submission i writes set i % (depth + 1) — never the set of a batch that is still in
    flight or that the same `submit()` call is handing out (the slot's own buffers were overwritten by the batch launched
    right behind the one returned).  Arrays handed out stay valid until the NEXT `submit()` call."""

    def __init__(self, model, vq_model, batch: int, n_samples: int, depth: int = 2, use_graph: bool = True):
        self.runners = [ClipRunner(model, vq_model, batch, n_samples, use_graph=use_graph) for _ in range(depth)]
        self.device = self.runners[0].device
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(depth)]
        self.done = [torch.cuda.Event() for _ in range(depth)]
        self.depth, self.frames_out, self.batch = depth, self.runners[0].frames_out, batch
        r0 = self.runners[0]
        self.host_sets = [(tuple(torch.empty(h.shape, dtype=h.dtype, pin_memory=True) for h in r0.host),
                           torch.zeros(1, dtype=torch.int32, pin_memory=True)) for _ in range(depth + 1)]
        self._next, self._inflight, self._submitted = 0, [], 0

    def submit(self, audio=None, speaker_id=None):
        """Launch one batch (audio (B, L) float32 on the device or in pinned host memory); blocks only when `depth` batches are
        already in flight (then the oldest is collected first and returned)."""
        ready = self.collect() if len(self._inflight) == self.depth else None
        slot = self._next
        self._next = (slot + 1) % self.depth
        hset = self._submitted % (self.depth + 1)
        self._submitted += 1
        r, s = self.runners[slot], self.streams[slot]
        host, flag = self.host_sets[hset]
        s.wait_stream(torch.cuda.current_stream(self.device))       # the caller's writes to `audio` are ordered before the copy
        with torch.cuda.stream(s):
            r.run_device(audio, speaker_id)
            r._to_host(host)
            flag.copy_(r.nonfinite, non_blocking=True)
            self.done[slot].record(s)
        self._inflight.append((slot, hset))
        return ready

    def collect(self):
        """Wait for the OLDEST batch in flight; returns its (poses, expressions, trans) numpy arrays (views of pinned buffers, valid
        until the next `submit()` call)."""
        slot, hset = self._inflight.pop(0)
        self.done[slot].synchronize()
        host, flag = self.host_sets[hset]
        r = self.runners[slot]
        r.nonfinite_host.copy_(flag)
        r._raise_if_nonfinite()
        return tuple(h.numpy() for h in host)

    def drain(self):
        """Collect everything in flight -> list of result tuples in submission order (the batches in flight use distinct host sets: all of
        them stay valid until the next `submit()`)."""
        out = []
        while self._inflight:
            out.append(self.collect())
        return out


class LstmClipRunner:
    """One DisCo / CaMN forward (WavEncoder, input projections, one persistent `emage_lstm_layer` launch per LSTM layer — or, in
    the exact-fp32 mode, one `emage_lstm_step_pair` launch per time step —, output MLPs, rot-6D -> axis-angle) captured as ONE
    hipGraph for a fixed (batch, audio length).  `__call__(audio)` returns (motion (B,T,pose_dims), axis_angle (B,T,165)) as numpy
    arrays (views of pinned buffers, valid until the next call); inf / NaN in them, or a lost block of the persistent recurrence,
    raises."""

    def __init__(self, model, batch: int, n_samples: int, use_graph: bool = True, warmup: int = 1):
        dev = model.device
        if dev.type != "cuda":
            raise RuntimeError("LstmClipRunner needs the model on an MI355X device")
        self.model, self.device = model, dev
        self.audio = torch.zeros(batch, n_samples, dtype=torch.float32, device=dev)
        self.speaker_id = torch.zeros(batch, 1, dtype=torch.long, device=dev)
        self.nonfinite = torch.zeros(2, dtype=torch.int32, device=dev)       # [non-finite results, lost blocks of the recurrences]
        self.nonfinite_host = torch.zeros(2, dtype=torch.int32, pin_memory=True)
        self.graph = None
        for _ in range(max(1, warmup)):
            out = self._step()
        torch.cuda.synchronize(dev)
        if use_graph:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                out = self._step()
        self.out = out
        self._operands = model._packed               # what the captured launches read: alive as long as the graph is
        self.host = tuple(torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in out)
        self.frames_out = int(out[0].shape[1])
        self._checked_replays = 0

    def _step(self):
        o = self.model.forward(self.audio, self.speaker_id)
        b = self.audio.shape[0]
        out = o["motion"].reshape(b, o["motion"].shape[1], -1), o["motion_axis_angle"]
        self.nonfinite.zero_()                       # health check, as in ClipRunner: inf / NaN in the generated motion raises
        for t in out:
            ops.count_nonfinite(t.contiguous(), self.nonfinite[:1])
        # ... and so does a lost block of a persistent recurrence (its output is finite but wrong): the error words of this
        # forward's launches are folded into the second counter word inside the graph, so EVERY replay is checked
        self.model.fold_kernel_health(self.nonfinite[1:])
        return out

    def __call__(self, audio=None, speaker_id=None):
        if audio is not None:
            self.audio.copy_(audio, non_blocking=True)
        if speaker_id is not None:
            self.speaker_id.copy_(speaker_id, non_blocking=True)
        if self.graph is not None:
            self.graph.replay()
        else:
            self.out = self._step()
        for h, d in zip(self.host, self.out):
            h.copy_(d, non_blocking=True)
        self.nonfinite_host.copy_(self.nonfinite, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        if int(self.nonfinite_host[0]):
            raise FloatingPointError(f"{int(self.nonfinite_host[0])} non-finite values in the generated motion (precision {self.model.precision!r}): in f16x3 an "
                                     "activation beyond |x| < 4094 overflows the fp16 planes — run this checkpoint with set_precision('fp32')")
        if int(self.nonfinite_host[1]):
            from ._lib import EmageKernelError
            raise EmageKernelError("emage_lstm_layer: a block timed out waiting for its group (blocks not co-resident: another kernel holding CUs?); "
                                   "the motion of this batch is invalid")
        return tuple(h.numpy() for h in self.host)

    def check(self):
        """Raise if a persistent-recurrence launch of a replay reported a lost block (synchronises)."""
        self.model.check_kernels()

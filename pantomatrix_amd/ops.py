"""PyTorch-ROCm custom ops over the C ABI ("invoked from Python via PyTorch-ROCm custom ops").

Every entry point of include/emage_hip.h is registered as an operator of the `emage` torch.library namespace
(`torch.ops.emage.gemm`, `torch.ops.emage.attention`, ...).  Each op has ONE implementation, for the CUDA (= ROCm)
dispatch key: pointer / stride extraction and a launch into libemage_hip.so on torch's current HIP stream.  There is no
CPU kernel behind any of them, so dispatching an op on CPU tensors fails in the dispatcher — the path has no fallback.
The public functions below (what the model classes call) allocate outputs, fill in defaults and call the op.
PyTorch is plumbing here (device memory, streams, the dispatcher); every kernel is in libemage_hip.so."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import F32, BF16, F16X3, H2, check, EmageKernelError

_LIBRARY = torch.library.Library("emage", "DEF")


_RECORDER = [None]        # the active `Lockstep` (ops issued inside one of its chains are recorded, not launched), else None
_TRACE = [None]           # measurement hook (bench.py / tools): an object with tag() -> label of the calling scope, evaluated when an op is
                          # CALLED, and fire(kind, entries, launch) which must call launch() — it brackets every actual launch
_META = {}                # bookkeeping the next op call carries to the hook (gemm: k_real, the unpadded contraction length)
LOCKSTEP_CHECK = False    # debug mode of `lockstep()`: while chains are being recorded, any NON-emage torch operator that touches
                          # the storage of a recorded (= not yet launched) emage output raises — such an op would run at record time, ahead of
                          # its producer, and read or overwrite uninitialised memory.  Costs a TorchDispatchMode: tests only


def _written_args(op, args):
    """The tensor arguments an emage operator writes (its schema's `Tensor(a!)` annotations)."""
    out = []
    for a, v in zip(op.default._schema.arguments, args):
        if a.alias_info is not None and a.alias_info.is_write and torch.is_tensor(v):
            out.append(v)
    return out


def _storage_key(t):
    try:
        return t.untyped_storage().data_ptr()
    except Exception:  # noqa: BLE001  (meta / fake tensors)
        return None


class _PendingGuard(torch.utils._python_dispatch.TorchDispatchMode):
    """`LOCKSTEP_CHECK`: active between the first recorded op of a lockstep and its `run()`."""

    def __init__(self, pending):
        super().__init__()
        self.pending = pending

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if self.pending and func.namespace != "emage" and not getattr(func, "is_view", False):
            import torch.utils._pytree as pytree
            for v in pytree.tree_leaves((args, kwargs)):
                if torch.is_tensor(v) and v.numel() and _storage_key(v) in self.pending:
                    raise RuntimeError(f"ops.lockstep: {func} touches the output of emage::{self.pending[_storage_key(v)]}, which is recorded in a "
                                       "chain and not launched yet — only emage ops may consume a chain's tensors before the lockstep exits")
        return func(*args, **kwargs)


def _entry(name, op, args):
    meta = dict(_META)
    _META.clear()
    tr = _TRACE[0]
    if tr is not None:
        meta["tag"] = tr.tag()
    return (name, op, args, meta)


def _fire(kind, entries, launch):
    tr = _TRACE[0]
    return tr.fire(kind, entries, launch) if tr is not None else launch()


def _op(name, schema):
    """Register `emage::name` with `schema`; the decorated function is its CUDA implementation.  Returns the callable the public
    functions below use: the operator itself, or — inside a `lockstep()` chain — a recording of the call."""
    def deco(fn):
        _LIBRARY.define(name + schema)
        _LIBRARY.impl(name, fn, "CUDA")
        op = getattr(torch.ops.emage, name)

        def call(*args):
            e = _entry(name, op, args)
            rec = _RECORDER[0]
            if rec is not None:
                return rec.record(e)
            return _fire(name, [e], lambda: op(*args))

        call.op = op
        return call
    return deco


class Lockstep:
    """Several INDEPENDENT launch chains issued in lock step, so that their contractions meet in grouped launches
    (`emage_gemm_grouped`): the four VQ part decoders, the three refinement layers with their heads, ... run the same sequence of
    shapes on different weights.  Inside `with ops.lockstep() as ls:` every chain is written as before, under `with ls.chain():` —
    its emage ops are RECORDED instead of launched (their outputs are allocated as usual: a later op of the chain needs the tensor,
    not its values).  On exit the chains are walked together: each chain's ops up to its next EMAGE_H2 contraction are launched in
    order, then the contractions waiting at the head of the chains go out as ONE `emage_gemm_grouped` call (the library puts those
    that select the same tile configuration into one launch), and so on.  Per chain the launch order is unchanged; across chains there
    is no dependency by contract.  Only emage ops (and allocations / fills that PRECEDE their consumers) may appear inside a chain: a
    torch op that reads a chain's result would run at record time, before the producer.  Everything goes to the current stream."""

    def __init__(self, enabled=True):
        self.enabled, self.chains, self.cur = bool(enabled), [], None
        self.pending = {}                 # LOCKSTEP_CHECK: storage pointer -> name of the recorded op that will write it
        self.launches = []                # (kind, count) log of the last run: ("group", n) / ("gemm", 1) / (op name, 1)
        self.groups = []                  # the recorded contractions of every grouped call of the last run (tests: `grouped_launch_count`)

    def chain(self):
        import contextlib

        @contextlib.contextmanager
        def cm():
            if self.cur is not None:
                raise RuntimeError("ops.lockstep: chains do not nest")
            self.cur = []
            self.chains.append(self.cur)
            try:
                yield
            finally:
                self.cur = None
        return cm()

    def record(self, entry):
        if self.cur is None:
            raise RuntimeError(f"ops.lockstep: emage::{entry[0]} issued outside a chain (it would overtake the recorded launches)")
        self.cur.append(entry)
        if LOCKSTEP_CHECK:
            for t in _written_args(entry[1], entry[2]):
                k = _storage_key(t)
                if k:
                    self.pending[k] = entry[0]

    @staticmethod
    def _groupable(entry):
        return entry[0] == "gemm" and entry[2][0] & 0xff == H2

    def _issue(self, entry):
        _fire(entry[0], [entry], lambda: entry[1](*entry[2]))
        self.launches.append((entry[0], 1))

    def _issue_group(self, entries):
        _fire("gemm_grouped", entries, lambda: _gemm_grouped_entries(entries))
        self.launches.append(("group", len(entries)))
        self.groups.append(entries)

    def run(self):
        self.launches, self.groups = [], []
        ptr = [0] * len(self.chains)
        while True:
            heads = []
            for ci, ch in enumerate(self.chains):
                while ptr[ci] < len(ch) and not self._groupable(ch[ptr[ci]]):
                    self._issue(ch[ptr[ci]])
                    ptr[ci] += 1
                if ptr[ci] < len(ch):
                    heads.append(ch[ptr[ci]])
                    ptr[ci] += 1
            if not heads:
                break
            if len(heads) == 1 or not self.enabled:
                for e in heads:
                    self._issue(e)
            else:
                self._issue_group(heads)
        self.chains = []


def lockstep(enabled=True):
    """Context manager -> `Lockstep` (see there).  enabled=False: the chains are still recorded and walked in lock step, but every
    contraction is its own launch (the A/B form: same launch order, no grouping).  Nested use joins the outer lockstep's CURRENT chain."""
    import contextlib

    @contextlib.contextmanager
    def cm():
        outer = _RECORDER[0]
        if outer is not None:             # already recording: the inner chains are simply part of the outer chain
            class _Join:
                enabled, launches = outer.enabled, outer.launches

                @staticmethod
                def chain():
                    return contextlib.nullcontext()
            yield _Join()
            return
        ls = Lockstep(enabled)
        _RECORDER[0] = ls
        ok = False
        guard = _PendingGuard(ls.pending) if LOCKSTEP_CHECK else contextlib.nullcontext()
        try:
            with guard:
                yield ls
            ok = True
        finally:
            _RECORDER[0] = None
            ls.pending.clear()
            if ok:
                ls.run()
    return cm()

# storage type of activations per precision code; F16X3 is a GEMM-only operand mode over float32 storage
# H2: the pre-split storage of the split-fp16 mode (csrc/h2.h) — float32-sized elements, 32-byte groups of 8 columns = [8 fp16 hi | 8 fp16 lo]
class _TorchDtypes(dict):
    def __missing__(self, code):             # EMAGE_H2 with an activation shift (`h2_shifted`): the storage type of the plain code
        return self[code & 0xff]


TORCH_DTYPE = _TorchDtypes({F32: torch.float32, BF16: torch.bfloat16, F16X3: torch.float32, H2: torch.float32})
A_SCALE_F16X3 = 16.0    # activations are multiplied by this power of two before the fp16 hi/lo split (|x| < 4094 stays finite)
MAX_ACT_SHIFT = 12


def h2_shifted(shift):
    """The dtype code EMAGE_H2_SHIFT(k) (include/emage_hip.h): activation images hold x * 2^(4 - k) — a model whose activations pass 4094
    runs its split-fp16 path with k > 0 (|x| < 4094 * 2^k) instead of overflowing the fp16 hi plane; k = 0 is plain EMAGE_H2."""
    shift = int(shift)
    if not 0 <= shift <= MAX_ACT_SHIFT:
        raise ValueError(f"activation shift {shift}: 0..{MAX_ACT_SHIFT}")
    return H2 | (shift << 8)


def act_scale(dtype):
    """The scale of the activation images of a dtype code (16 for plain EMAGE_H2 / EMAGE_F16X3)."""
    return A_SCALE_F16X3 * 2.0 ** -(dtype >> 8)


def split_f16_weights(w2d, scale=None):
    """Host packing of an (N, K) fp32 weight matrix for EMAGE_F16X3 (include/emage_hip.h: emage_gemm), K % 32 == 0.
    Returns (packed (N, K) float32-typed buffer holding the fp16 planes, w_scale).  w_scale is the power of two that
    puts max|w| into [2^12, 2^13): both planes stay normal fp16 numbers for every weight above max|w| * 2^-16.
    scale given (a power of two chosen at an earlier packing of the same weight): no read-back of max|w| — the packing is then a
    pure sequence of device launches (a training step re-packs after every update, also inside a captured graph)."""
    n, k = w2d.shape
    assert k % 32 == 0
    w = w2d.to(torch.float32)
    if scale is None:
        scale = _f16_scale(w)
    if w.is_cuda and n:                  # on the device: ONE launch, the same bits as the tensor arithmetic below (tests/test_kernels_gpu.py)
        return pack_f16x3_weights(w, scale), scale
    ws = w * scale
    hi = ws.to(torch.float16)
    lo = (ws - hi.to(torch.float32)).to(torch.float16)

    def chunked(plane):      # k = 32*kt + 16*g2 + 4*g + e  ->  [kt][g][g2][e]: chunk g holds k = 4g..4g+3, 16+4g..16+4g+3
        return plane.view(n, k // 32, 2, 4, 4).permute(0, 1, 3, 2, 4).reshape(n, k // 32, 32)

    packed = torch.stack([chunked(hi), chunked(lo)], dim=2).reshape(n, 2 * k).contiguous()
    return packed.view(torch.float32), scale



def _f16_scale(w):
    import math
    mx = float(w.abs().max()) if w.numel() else 0.0
    return 2.0 ** (12 - math.floor(math.log2(mx))) if mx > 0 and math.isfinite(mx) else 1.0


F16_SCALE_LO, F16_SCALE_HI = 2.0 ** 10, 2.0 ** 14     # `_f16_scale` puts max |w * scale| into [2^12, 2^13); outside this band the scale is re-derived


def f16_scale_out_of_range(w, scale):
    """int32 device scalar: 1 when max |w| * scale has left [2^10, 2^14) — a weight packed with a power-of-two scale chosen at an EARLIER
    packing has since grown towards the fp16 range (the hi plane overflows at 2^16: two doublings of margin are left when this fires)
    or shrunk so far that its planes lose bits; NaN / inf count as out of range.  No host read-back (a training step re-packs inside a
    captured graph); the caller accumulates the flags and reads them where it reads its results."""
    mx = w.detach().abs().amax().to(torch.float32) * float(scale)
    return (~(((mx >= F16_SCALE_LO) & (mx < F16_SCALE_HI)) | (mx == 0))).to(torch.int32)       # an all-zero weight has no range to leave


def f16_scales_out_of_range(ws, scales):
    """`f16_scale_out_of_range` for a LIST of weights in a handful of launches (multi-tensor max-abs): int32 device scalar = how many of
    them have left the band.  A training step re-packs ~480 operands; one check per operand was ~2 500 tiny launches per step."""
    if not ws:
        return None
    mx = torch._foreach_norm([w.detach() for w in ws], float("inf"))
    mx = torch.stack(torch._foreach_mul(mx, [float(s) for s in scales])).to(torch.float32)     # scalars ride in the launch arguments: capture-safe
    return (~(((mx >= F16_SCALE_LO) & (mx < F16_SCALE_HI)) | (mx == 0))).sum().to(torch.int32)


def h2_pack(x, scale=A_SCALE_F16X3):
    """(..., C) fp32, C % 8 == 0 -> the EMAGE_H2 image (csrc/h2.h) as a float32-typed tensor of the same shape: every group of
    8 columns becomes [8 fp16 hi | 8 fp16 lo] with x * scale = hi + lo.  Host-side helper (weights, tests, tools); on the hot
    path the kernels' epilogues write this format themselves."""
    c = x.shape[-1]
    assert c % 8 == 0
    xs = x.to(torch.float32) * scale
    hi = xs.to(torch.float16)
    lo = (xs - hi.to(torch.float32)).to(torch.float16)
    g = torch.stack([hi.reshape(*x.shape[:-1], c // 8, 8), lo.reshape(*x.shape[:-1], c // 8, 8)], dim=-2)
    return g.reshape(*x.shape[:-1], 2 * c).contiguous().view(torch.float32)


def h2_unpack(t, scale=A_SCALE_F16X3):
    """Inverse of `h2_pack` (up to the 2^-22 relative residual of the split): float32-typed H2 image -> fp32 values."""
    c = t.shape[-1]
    assert c % 8 == 0
    g = t.contiguous().view(torch.float16).reshape(*t.shape[:-1], c // 8, 2, 8).to(torch.float32)
    return ((g[..., 0, :] + g[..., 1, :]) / scale).reshape(*t.shape[:-1], c)


def split_f16_weights_h2(w2d, scale=None):
    """Host packing of an (N, K) fp32 weight matrix for EMAGE_H2 (K % 32 == 0): the H2 image of W * w_scale, natural k order
    (the same layout the activations use).  Returns (packed (N, K) float32-typed, w_scale); `scale` as for `split_f16_weights`."""
    n, k = w2d.shape
    assert k % 32 == 0
    w = w2d.to(torch.float32)
    scale = _f16_scale(w) if scale is None else scale
    if w.is_cuda and n:                  # on the device: the activation-cast kernel (h2 images carry a fixed x16: the rest of the scale goes in front)
        return h2_cast(w if w.stride(1) == 1 else w.contiguous(), k, scale=scale / A_SCALE_F16X3), scale
    return h2_pack(w, scale), scale


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t):
    if not t.is_cuda:
        raise RuntimeError("pantomatrix_amd kernels need tensors on an MI355X device (no CPU fallback)")


def _ld(t):
    """Row stride (elements) of a 2-D view (rows, C) with unit inner stride."""
    assert t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
    return t.stride(0)


def round_up(x, m):
    return (x + m - 1) // m * m


def index_view(idx):
    """(rows, ld, tstride) of an int64 index tensor as the kernels address it (include/emage_hip.h: emage_vq_argmin_f32):
    a 1-D contiguous list -> (0, 0, 1); a 2-D (B, T) view with frame stride 1 or 0 (one id per clip expanded over T)
    and any clip stride is used in place — no gather / contiguous copy."""
    assert idx.dtype == torch.int64
    if idx.dim() == 1:
        assert idx.stride(0) == 1 or idx.numel() <= 1
        return 0, 0, 1
    assert idx.dim() == 2 and idx.stride(1) in (0, 1), (idx.shape, idx.stride())
    return idx.shape[1], idx.stride(0), idx.stride(1)


@_op("vq_argmin", "(Tensor z, Tensor codebook, Tensor(a!) idx) -> ()")
def _vq_argmin(z, codebook, idx):
    rows, ld, _ts = index_view(idx)
    n, d = z.shape
    check(_lib.load().emage_vq_argmin_f32(_ptr(z), _ld(z), _ptr(codebook), _ptr(idx), rows, ld, n, codebook.shape[0], d, _stream()), "vq_argmin")


def vq_argmin(z2d, codebook, out=None):
    """z2d (N,D) fp32, codebook (K,D) fp32 contiguous -> int64 nearest-code indices; `out`: an (N,) tensor or a (B, T)
    view (B*T == N) of a larger code buffer to fill in place."""
    _dev(z2d)
    assert z2d.dtype == torch.float32 and codebook.dtype == torch.float32 and codebook.is_contiguous()
    n, d = z2d.shape
    idx = torch.empty(n, dtype=torch.int64, device=z2d.device) if out is None else out
    assert idx.numel() == n and index_view(idx)[2] == 1
    _vq_argmin(z2d, codebook, idx)
    return idx


@_op("argmax_logsoftmax", "(Tensor logits, Tensor(a!) idx) -> ()")
def _argmax_logsoftmax(logits, idx):
    rows, ld, _ts = index_view(idx)
    n, c = logits.shape
    check(_lib.load().emage_argmax_logsoftmax_f32(_ptr(logits), _ld(logits), _ptr(idx), rows, ld, n, c, _stream()), "argmax_logsoftmax")


def argmax_logsoftmax(logits2d, out=None):
    _dev(logits2d)
    assert logits2d.dtype == torch.float32
    n, c = logits2d.shape
    idx = torch.empty(n, dtype=torch.int64, device=logits2d.device) if out is None else out
    assert idx.numel() == n and index_view(idx)[2] == 1
    _argmax_logsoftmax(logits2d, idx)
    return idx


@_op("gather_rows", "(Tensor table, Tensor idx, Tensor(a!) out, int dtype) -> ()")
def _gather_rows(table, idx, out, dtype):
    rows, ld, ts = index_view(idx)
    k, d = table.shape
    n, n_store = out.shape
    check(_lib.load().emage_gather_rows(_ptr(table), _ptr(idx), rows, ld, ts, _ptr(out), _ld(out), n_store, n, k, d, dtype, _stream()), "gather_rows")


def gather_rows(table, idx, dtype, n_store=None):
    """table (K,D) fp32 rows selected by an int64 index list / (B,T) view (see `index_view`) -> (N, n_store) in `dtype`."""
    _dev(table)
    k, d = table.shape
    n_store = d if n_store is None else n_store
    if idx.dim() > 2 or (idx.dim() == 2 and idx.stride(1) not in (0, 1)) or (idx.dim() == 1 and idx.numel() > 1 and idx.stride(0) != 1):
        idx = idx.reshape(-1).contiguous()
    out = torch.empty(idx.numel(), n_store, dtype=TORCH_DTYPE[dtype], device=table.device)
    _gather_rows(table, idx, out, dtype)
    return out


class SplitKScratch:
    """Scratch memory a caller lends to `gemm` launches of few rows so that they may split their K range inside the launch (include/emage_hip.h:
    emage_gemm_problem sk_ws / sk_count): one (workspace, zeroed tile counters) pair per (device, stream) — launches on one stream are ordered, so
    they share a pair; concurrent lanes get their own.  Owned by whoever owns the launch sequence (runtime.ClipRunner: one per captured graph, so two
    graphs in flight never share scratch); `with ops.splitk_scope(scratch):` makes it the one the model code hands to its contractions."""

    def __init__(self, mbytes=8, tiles=256):
        self.mbytes, self.tiles, self.pool = int(mbytes), int(tiles), {}

    def get(self, device):
        device = torch.device(device)
        key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        hit = self.pool.get(key)
        if hit is None:
            hit = self.pool[key] = (torch.empty(self.mbytes << 18, dtype=torch.float32, device=device), torch.zeros(self.tiles, dtype=torch.int32, device=device))
        return hit


_SPLITK = [None]
SPLITK_MAX_ROWS = 512        # rows up to which the model code offers its contractions the scratch (the library still decides per launch)


def splitk_scope(scratch):
    """Context: contractions of at most SPLITK_MAX_ROWS rows issued inside are lent `scratch` (a `SplitKScratch`, or None = none)."""
    import contextlib

    @contextlib.contextmanager
    def cm():
        prev, _SPLITK[0] = _SPLITK[0], scratch
        try:
            yield scratch
        finally:
            _SPLITK[0] = prev
    return cm()


def _gemm_fields(dtype, a, w, bias, slope, res, out, out_f32, out_t, n, cp, n_store, t_col0, t_rows, res_first, taps, stride, pad, lin, lout, m,
                 w_scale, a_scale, res_h2, ln_stats=None, ln_c=None, rs_stats=None, rs_gamma=None, rs_beta=None, st_out=None, ln_eps=0.0,
                 sk_ws=None, sk_count=None):
    """The arguments of the `emage::gemm` operator -> the fields of one emage_gemm call (the order of `emage_gemm_problem`)."""
    res_f32 = 1 if (res is not None and res.dtype == torch.float32 and not res_h2) else 0
    t_ld = out_t.shape[-1] if out_t is not None else 0
    for st in (ln_stats, rs_stats, st_out):             # (M, partials, 2) float32, contiguous: {mean, M2} over 32 columns each
        assert st is None or (st.dtype == torch.float32 and st.is_contiguous() and st.dim() == 3 and st.shape[2] == 2 and st.shape[0] >= m)
    return dict(A=_ptr(a), W=_ptr(w), bias=_ptr(bias), slope=_ptr(slope), res=_ptr(res), out=_ptr(out), out_f32=_ptr(out_f32), out_t=_ptr(out_t),
                lda=_ld(a), ldr=_ld(res) if res is not None else 0, res_is_f32=res_f32, res_first=1 if res_first else 0,
                ldo=_ld(out) if out is not None else 0, n_store=n_store, ldf=_ld(out_f32) if out_f32 is not None else 0,
                t_col0=t_col0, t_rows=t_rows, t_ld=t_ld, M=m, N=n, Cp=cp, taps=taps, stride=stride, pad=pad, Lin=lin, Lout=lout,
                a_scale=a_scale, w_scale=w_scale,
                ln_stats=_ptr(ln_stats), ln_c=_ptr(ln_c), rs_stats=_ptr(rs_stats), rs_gamma=_ptr(rs_gamma), rs_beta=_ptr(rs_beta), st_out=_ptr(st_out),
                ln_np=ln_stats.shape[1] if ln_stats is not None else 0, rs_np=rs_stats.shape[1] if rs_stats is not None else 0, ln_eps=float(ln_eps),
                sk_ws=_ptr(sk_ws), sk_ws_bytes=sk_ws.numel() * sk_ws.element_size() if sk_ws is not None else 0,
                sk_count=_ptr(sk_count), sk_tiles=sk_count.numel() if sk_count is not None else 0)


@_op("gemm", "(int dtype, Tensor a, Tensor w, Tensor? bias, Tensor? slope, Tensor? res, Tensor(a!)? out, Tensor(b!)? out_f32, "
             "Tensor(c!)? out_t, int n, int cp, int n_store, int t_col0, int t_rows, bool res_first, int taps, int stride, int pad, "
             "int lin, int lout, int m, float w_scale, float a_scale, bool res_h2, Tensor? ln_stats=None, Tensor? ln_c=None, "
             "Tensor? rs_stats=None, Tensor? rs_gamma=None, Tensor? rs_beta=None, Tensor(d!)? st_out=None, float ln_eps=0.0, "
             "Tensor(e!)? sk_ws=None, Tensor(f!)? sk_count=None) -> ()")
def _gemm(dtype, *args):
    f = _gemm_fields(dtype, *args)
    if f["ln_stats"] or f["rs_stats"] or f["st_out"] or f["sk_ws"]:     # LayerNorm fold / split-K scratch: the problem-struct entry point carries those fields
        arr, n = _problem_array([int(f[k] or 0) for k in _GEMM_PROBLEM_INTS], [float(f["a_scale"]), float(f["w_scale"]), float(f["ln_eps"])])
        check(_lib.load().emage_gemm_grouped(dtype, arr, n, _stream()), "gemm (LayerNorm fold)")
        return
    check(_lib.load().emage_gemm(dtype, f["A"], f["lda"], f["W"], f["bias"], f["slope"], f["res"], f["ldr"], f["res_is_f32"], f["res_first"],
                                 f["out"], f["ldo"], f["n_store"], f["out_f32"], f["ldf"], f["out_t"], f["t_col0"], f["t_rows"], f["t_ld"],
                                 f["M"], f["N"], f["Cp"], f["taps"], f["stride"], f["pad"], f["Lin"], f["Lout"], f["a_scale"], f["w_scale"], _stream()), "gemm")


@_op("gemm_ws", "(int dtype, Tensor a, Tensor w, Tensor? bias, Tensor? slope, Tensor? res, Tensor(a!)? out, Tensor(b!)? out_f32, "
                "Tensor(c!)? out_t, int n, int cp, int n_store, int t_col0, int t_rows, bool res_first, int taps, int stride, int pad, "
                "int lin, int lout, int m, float w_scale, float a_scale, bool res_h2, Tensor(d!) workspace) -> ()")
def _gemm_ws(dtype, *args):
    ws = args[-1]
    f = _gemm_fields(dtype, *args[:-1])
    check(_lib.load().emage_gemm_ws(dtype, f["A"], f["lda"], f["W"], f["bias"], f["slope"], f["res"], f["ldr"], f["res_is_f32"], f["res_first"],
                                    f["out"], f["ldo"], f["n_store"], f["out_f32"], f["ldf"], f["out_t"], f["t_col0"], f["t_rows"], f["t_ld"],
                                    f["M"], f["N"], f["Cp"], f["taps"], f["stride"], f["pad"], f["Lin"], f["Lout"], f["a_scale"], f["w_scale"],
                                    _ptr(ws), ws.numel() * ws.element_size(), _stream()), "gemm_ws")


# Descriptor-table operator: `tensors` lists every tensor the problems touch (dispatch key, aliasing: they may be written), `desc` holds
# per problem the 26 integer words of `emage_gemm_problem` in field order (device addresses first), `scales` its (a_scale, w_scale).
_GEMM_PROBLEM_PTRS = ("A", "W", "bias", "slope", "res", "out", "out_f32", "out_t", "ln_stats", "ln_c", "rs_stats", "rs_gamma", "rs_beta", "st_out", "sk_ws", "sk_count")
_GEMM_PROBLEM_INTS = ("A", "W", "bias", "slope", "res", "out", "out_f32", "out_t", "lda", "ldr", "res_is_f32", "res_first", "ldo", "n_store", "ldf",
                      "t_col0", "t_rows", "t_ld", "M", "N", "Cp", "taps", "stride", "pad", "Lin", "Lout",
                      "ln_stats", "ln_c", "rs_stats", "rs_gamma", "rs_beta", "st_out", "ln_np", "rs_np", "sk_ws", "sk_ws_bytes", "sk_count", "sk_tiles")
_GEMM_PROBLEM_SCALES = 3                  # floats per problem: a_scale, w_scale, ln_eps


def _problem_array(desc, scales):
    k = len(_GEMM_PROBLEM_INTS)
    n = len(desc) // k
    arr = (_lib.GemmProblem * n)()
    for i in range(n):
        for j, name in enumerate(_GEMM_PROBLEM_INTS):
            v = desc[i * k + j]
            setattr(arr[i], name, (v or None) if name in _GEMM_PROBLEM_PTRS else v)
        arr[i].a_scale, arr[i].w_scale, arr[i].ln_eps = scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]
    return arr, n


@_op("gemm_grouped", "(int dtype, Tensor(a!)[] tensors, int[] desc, float[] scales) -> ()")
def _gemm_grouped(dtype, tensors, desc, scales):
    arr, n = _problem_array(desc, scales)
    check(_lib.load().emage_gemm_grouped(dtype, arr, n, _stream()), "gemm_grouped")


def grouped_launch_count(entries):
    """Kernel launches `emage_gemm_grouped` makes of these recorded `emage::gemm` calls (measurement bookkeeping; launches nothing)."""
    dtype, desc, scales = entries[0][2][0], [], []
    for e in entries:
        f = _gemm_fields(*e[2])
        desc += [int(f[k] or 0) for k in _GEMM_PROBLEM_INTS]
        scales += [float(f["a_scale"]), float(f["w_scale"]), float(f["ln_eps"])]
    arr, n = _problem_array(desc, scales)
    rc = _lib.load().emage_gemm_grouped_launches(dtype, arr, n)
    if rc <= 0:
        check(rc or -1, "gemm_grouped_launches")
    return rc


def _gemm_grouped_entries(entries):
    """Recorded `emage::gemm` calls (Lockstep) of ONE dtype -> one `emage::gemm_grouped` call."""
    dtype = entries[0][2][0]
    tensors, desc, scales = [], [], []
    for _name, _op_, args, _meta in entries:
        assert args[0] == dtype
        f = _gemm_fields(*args)
        desc += [int(f[k] or 0) for k in _GEMM_PROBLEM_INTS]
        scales += [float(f["a_scale"]), float(f["w_scale"]), float(f["ln_eps"])]
        tensors += [t for t in list(args[1:9]) + list(args[24:30]) + list(args[31:33]) if torch.is_tensor(t)]
    _gemm_grouped.op(dtype, tensors, desc, scales)


def gemm_grouped(dtype, problems):
    """Several independent `gemm` problems (a list of dicts of `gemm`'s arguments: a, w, bias, slope, res, out, out_f32, out_t + its
    keywords) in as few launches as the library can make of them (include/emage_hip.h: emage_gemm_grouped); bit-identical to issuing
    them one by one."""
    with lockstep() as ls:
        for pr in problems:
            with ls.chain():
                pr = dict(pr)
                gemm(dtype, pr.pop("a"), pr.pop("w"), pr.pop("bias", None), pr.pop("slope", None), pr.pop("res", None), pr.pop("out", None),
                     pr.pop("out_f32", None), pr.pop("out_t", None), **pr)


def gemm(dtype, a, w, bias=None, slope=None, res=None, out=None, out_f32=None, out_t=None, *, n, cp,
         n_store=0, t_col0=0, t_rows=0, res_first=False, taps=1, stride=1, pad=0, lin=None, lout=None, m=None,
         k_real=None, w_scale=1.0, a_scale=None, res_h2=False, workspace=None, ln=None, res_ln=None, stats_out=None, ln_eps=1e-5, splitk=None):
    """See include/emage_hip.h:emage_gemm.  `a` (rows, lda) and `w` (n, taps*cp) are in `dtype`.  `k_real` (the
    unpadded contraction length) is bookkeeping for bench.py's algorithmic-flop count; the kernel ignores it.
    dtype F16X3: `a` is float32, `w` / `w_scale` come from `split_f16_weights`.  dtype H2: `a` / `out` are H2 images
    (float32-typed), `w` / `w_scale` from `split_f16_weights_h2`, `res` fp32 or (res_h2) an H2 image, `out_f32` / `out_t` fp32.
    workspace (a contiguous device tensor, scratch): emage_gemm_ws — split-K contractions store their K-slices as planes of it and add
    them in slice order (deterministic) instead of meeting through fp32 atomics."""
    _dev(a)
    m = a.shape[0] if m is None else m
    lin = m if lin is None else lin
    lout = m if lout is None else lout
    if k_real is not None:
        _META["k_real"] = k_real
    if workspace is not None:
        assert workspace.is_contiguous() and workspace.device == a.device
        _gemm_ws(dtype, a, w, bias, slope, res, out, out_f32, out_t, n, cp, n_store, t_col0, t_rows, bool(res_first), taps, stride, pad,
                 lin, lout, m, float(w_scale), float(act_scale(dtype) if a_scale is None else a_scale), bool(res_h2), workspace)
        return
    if ln is None and res_ln is None and stats_out is None and splitk is None:
        _gemm(dtype, a, w, bias, slope, res, out, out_f32, out_t, n, cp, n_store, t_col0, t_rows, bool(res_first), taps, stride, pad,
              lin, lout, m, float(w_scale), float(act_scale(dtype) if a_scale is None else a_scale), bool(res_h2))
        return
    # LayerNorm fold (include/emage_hip.h: emage_gemm_problem): ln = (row statistics of `a`, c) — `a` is the RAW pre-norm sum, `w` / `bias` the
    # folded W gamma / W beta + b; res_ln = (row statistics of `res`, gamma, beta) — `res` is the raw sum of a folded LayerNorm; stats_out: the
    # (M, n / 32, 2) partial row statistics of this launch's output
    # splitk = (scratch (float32, contiguous), counters (int32, ZERO, left zero)): lets a launch of few rows cut its K range into slices that meet
    # inside the launch (emage_gemm_problem: sk_ws / sk_count); launches on one stream may share the pair
    assert dtype & 0xff == H2 and workspace is None
    ln_stats, ln_c = ln if ln is not None else (None, None)
    rs_stats, rs_gamma, rs_beta = res_ln if res_ln is not None else (None, None, None)
    sk_ws, sk_count = splitk if splitk is not None else (None, None)
    assert sk_ws is None or (sk_ws.is_contiguous() and sk_count.is_contiguous() and sk_count.dtype == torch.int32)
    _gemm(dtype, a, w, bias, slope, res, out, out_f32, out_t, n, cp, n_store, t_col0, t_rows, bool(res_first), taps, stride, pad,
          lin, lout, m, float(w_scale), float(act_scale(dtype) if a_scale is None else a_scale), bool(res_h2),
          ln_stats, ln_c, rs_stats, rs_gamma, rs_beta, stats_out, float(ln_eps), sk_ws, sk_count)


@_op("wav_conv_in", "(int dtype, Tensor wav, Tensor w, Tensor? bias, Tensor? slope, Tensor(a!) out, int lout, int stride, int pad, "
                    "int nwin, int hop, int win_len) -> ()")
def _wav_conv_in(dtype, wav, w, bias, slope, out, lout, stride, pad, nwin, hop, win_len):
    c, taps = w.shape
    check(_lib.load().emage_wav_conv_in(dtype, _ptr(wav), wav.stride(0), win_len, nwin, hop, _ptr(w), _ptr(bias), _ptr(slope), _ptr(out), _ld(out),
                                        wav.shape[0], lout, c, taps, stride, pad, _stream()), "wav_conv_in")


def wav_conv_in(dtype, wav, w, bias, slope, out, lout, stride, pad, nwin=1, hop=0, win_len=None):
    """wav (B, L) fp32 view with unit sample stride (any clip stride).  nwin > 1: `nwin` windows of `win_len` samples per
    clip, window i at sample i*hop, written as output sequences i*B + b — the sliding windows are read in place."""
    _dev(wav)
    assert wav.dim() == 2 and wav.stride(1) == 1 and wav.dtype == torch.float32
    b, l = wav.shape
    win_len = l if win_len is None else win_len
    assert (nwin - 1) * hop + win_len <= l
    _wav_conv_in(dtype, wav, w, bias, slope, out, lout, stride, pad, nwin, hop, win_len)


@_op("conv_slab", "(int dtype, Tensor a, Tensor w, Tensor bias, Tensor slope, Tensor? res, Tensor(a!) out, int nseq, int l, int taps, int pad, "
                  "float w_scale, float a_scale) -> ()")
def _conv_slab(dtype, a, w, bias, slope, res, out, nseq, l, taps, pad, w_scale, a_scale):
    c = w.shape[0]
    check(_lib.load().emage_conv_slab(dtype, _ptr(a), _ld(a), _ptr(w), _ptr(bias), _ptr(slope), _ptr(res), _ld(res) if res is not None else 0,
                                      _ptr(out), _ld(out), nseq, l, c, taps, pad, a_scale, w_scale, _stream()), "conv_slab")


def conv_slab_supported(c, taps, stride):
    """Shapes the LDS-resident-slab convolution is built for (include/emage_hip.h: emage_conv_slab)."""
    return stride == 1 and c in (64, 128) and taps <= 16


def conv_slab(dtype, a, w, bias, slope, res, out, *, nseq, l, taps, pad, w_scale=1.0, a_scale=None):
    """Stride-1 Conv1d(C -> C, k = taps) + bias + shortcut + per-channel LeakyReLU with the input slab resident in LDS;
    a (nseq*l, lda) / res / out row views in the storage type of `dtype`, w the packed (C, taps*C) weights of emage_gemm."""
    _dev(a)
    _conv_slab(dtype, a, w, bias, slope, res, out, nseq, l, taps, pad, float(w_scale), float(A_SCALE_F16X3 if a_scale is None else a_scale))
    return out


@_op("wav_block0", "(int dtype, Tensor wav, Tensor w1, Tensor b1, float slope1, Tensor wds, Tensor bds, int stride1, int pad1, Tensor w2, Tensor bias2, "
                   "Tensor slope2, int taps2, int pad2, Tensor(a!) out, int l_out, int nwin, int hop, int win_len, float w_scale, float a_scale) -> ()")
def _wav_block0(dtype, wav, w1, b1, slope1, wds, bds, stride1, pad1, w2, bias2, slope2, taps2, pad2, out, l_out, nwin, hop, win_len, w_scale, a_scale):
    c, taps1 = w1.shape
    check(_lib.load().emage_wav_block0(dtype, _ptr(wav), wav.stride(0), win_len, nwin, hop, wav.shape[0], _ptr(w1), _ptr(b1), slope1, _ptr(wds), _ptr(bds),
                                       taps1, stride1, pad1, _ptr(w2), _ptr(bias2), _ptr(slope2), taps2, pad2, _ptr(out), _ld(out), l_out, c,
                                       a_scale, w_scale, _stream()), "wav_block0")


def wav_block0(dtype, wav, w1, b1, slope1, wds, bds, stride1, pad1, w2, bias2, slope2, taps2, pad2, out, l_out, *, nwin=1, hop=0, win_len=None,
               w_scale=1.0, a_scale=None):
    """WavEncoder block 0 in one launch: conv1 (+ folded BN + LeakyReLU) from the raw waveform into LDS, conv2 on it, the
    downsample shortcut added in the epilogue (P:283-294).  wav (B, L) fp32 window source as for `wav_conv_in`; w1 / wds
    (C, taps1) fp32 row views; w2 the packed (C, taps2*C) conv2 weights; out (nwin*B*l_out, ldo) in `dtype`'s storage."""
    _dev(wav)
    assert wav.dim() == 2 and wav.stride(1) == 1 and w1.is_contiguous() and wds.is_contiguous()
    win_len = wav.shape[1] if win_len is None else win_len
    _wav_block0(dtype, wav, w1, b1, float(slope1), wds, bds, stride1, pad1, w2, bias2, slope2, taps2, pad2, out, l_out, nwin, hop, win_len,
                float(w_scale), float(A_SCALE_F16X3 if a_scale is None else a_scale))
    return out


@_op("attention", "(int dtype, Tensor q, Tensor k, Tensor vt, int vt_rows, Tensor(a!) out, int b, int h, int tq, int tk, int hd) -> ()")
def _attention(dtype, q, k, vt, vt_rows, out, b, h, tq, tk, hd):
    check(_lib.load().emage_attention(dtype, _ptr(q), _ld(q), _ptr(k), _ld(k), _ptr(vt), vt.shape[-1], vt_rows, _ptr(out), _ld(out),
                                      b, h, tq, tk, hd, _stream()), "attention")


def attention(dtype, q, k, vt, vt_rows, out, b, h, tq, tk, hd):
    """q (B*Tq, ldq), k (B*Tk, ldk) 2-D views; vt a view into a (B, vt_rows, ldvt) buffer starting at this
    layer's first row; out (B*Tq, ldo)."""
    _dev(q)
    _attention(dtype, q, k, vt, vt_rows, out, b, h, tq, tk, hd)


@_op("attention_dropout", "(int dtype, Tensor q, Tensor k, Tensor vt, int vt_rows, Tensor(a!) out, int b, int h, int tq, int tk, int hd, Tensor pmask) -> ()")
def _attention_dropout(dtype, q, k, vt, vt_rows, out, b, h, tq, tk, hd, pmask):
    assert pmask.dtype == torch.float32 and pmask.is_contiguous() and tuple(pmask.shape) == (b, h, tq, tk)
    check(_lib.load().emage_attention_dropout(dtype, _ptr(q), _ld(q), _ptr(k), _ld(k), _ptr(vt), vt.shape[-1], vt_rows, _ptr(out), _ld(out),
                                              b, h, tq, tk, hd, _ptr(pmask), _stream()), "attention_dropout")


def attention_dropout(dtype, q, k, vt, vt_rows, out, b, h, tq, tk, hd, pmask):
    """`attention` of a TRAINING forward: the probabilities are multiplied by pmask (B, H, Tq, Tk) fp32 before P V
    (nn.MultiheadAttention's attention-probability dropout; the mask holds bernoulli / (1 - p))."""
    _dev(q)
    _attention_dropout(dtype, q, k, vt, vt_rows, out, b, h, tq, tk, hd, pmask)


@_op("bn_stats", "(Tensor x, Tensor(a!) mean, Tensor(b!) var, Tensor(c!)? running_mean, Tensor(d!)? running_var, float momentum, Tensor(e!) workspace) -> ()")
def _bn_stats(x, mean, var, running_mean, running_var, momentum, workspace):
    m, c = x.shape
    check(_lib.load().emage_bn_stats(_ptr(x), _ld(x), m, c, _ptr(workspace), workspace.numel() * workspace.element_size(), _ptr(mean), _ptr(var),
                                     _ptr(running_mean), _ptr(running_var), momentum, _stream()), "bn_stats")


def bn_stats(x, running_mean=None, running_var=None, momentum=0.1):
    """nn.BatchNorm1d (training) statistics of a channels-last fp32 (M, C) view: returns (mean, biased var) and updates
    the running buffers in place as torch does (unbiased variance, momentum)."""
    _dev(x)
    m, c = x.shape
    assert x.dtype == torch.float32 and x.stride(1) == 1
    nbytes = _lib.load().emage_bn_stats_workspace_bytes(m, c)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=x.device)
    mean, var = torch.empty(c, dtype=torch.float32, device=x.device), torch.empty(c, dtype=torch.float32, device=x.device)
    _bn_stats(x, mean, var, running_mean, running_var, float(momentum), ws)
    return mean, var


@_op("bn_apply", "(Tensor x, Tensor mean, Tensor var, Tensor gamma, Tensor beta, Tensor? sc, Tensor? sc_mean, Tensor? sc_var, Tensor? sc_gamma, "
                 "Tensor? sc_beta, float eps, float slope, Tensor(a!) out) -> ()")
def _bn_apply(x, mean, var, gamma, beta, sc, sc_mean, sc_var, sc_gamma, sc_beta, eps, slope, out):
    m, c = x.shape
    check(_lib.load().emage_bn_apply(_ptr(x), _ld(x), _ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta), _ptr(sc), _ld(sc) if sc is not None else 0,
                                     _ptr(sc_mean), _ptr(sc_var), _ptr(sc_gamma), _ptr(sc_beta), eps, slope, _ptr(out), _ld(out), m, c, _stream()), "bn_apply")


def bn_apply(x, stats, gamma, beta, out, *, slope=1.0, sc=None, sc_bn=None, eps=1e-5):
    """out = LeakyReLU(bn(x) + shortcut, slope) on fp32 (M, C) views; stats = (mean, var); sc = raw shortcut rows, or with
    sc_bn = (mean, var, gamma, beta) a shortcut that is batch-normalised itself (BasicBlock's downsample branch)."""
    _dev(x)
    sm, sv, sg, sb = sc_bn if sc_bn is not None else (None, None, None, None)
    _bn_apply(x, stats[0], stats[1], gamma, beta, sc, sm, sv, sg, sb, float(eps), float(slope), out)
    return out


@_op("mul_add", "(Tensor a, Tensor mask, int mask_t_rows, Tensor? b, Tensor(a!) out) -> ()")
def _mul_add(a, mask, mask_t_rows, b, out):
    m, c = a.shape
    check(_lib.load().emage_mul_add(_ptr(a), _ld(a), _ptr(mask), _ld(mask), mask_t_rows, _ptr(b), _ld(b) if b is not None else 0, _ptr(out), _ld(out),
                                    m, c, _stream()), "mul_add")


class PhiloxMask:
    """A dropout keep mask that exists as its KEY only: the tensor `dropout_mask(empty(shape), p, seed, mask_id, step)` would hold.
    `mul_add` draws it inside its kernel (forward and backward alike), so the mask never occupies memory; `materialize()` for consumers
    that need the tensor.  `view(m, c)` re-shapes like a contiguous tensor (the element order is the key's)."""

    def __init__(self, shape, p, seed, mask_id, step, device):
        self.shape, self.p, self.seed, self.mask_id, self.step, self.device = tuple(shape), float(p), int(seed), int(mask_id), step, device

    def view(self, *shape):
        n = 1
        for d in shape:
            n *= d
        m = 1
        for d in self.shape:
            m *= d
        assert n == m, (shape, self.shape)
        return PhiloxMask(shape, self.p, self.seed, self.mask_id, self.step, self.device)

    def materialize(self):
        return dropout_mask(torch.empty(self.shape, dtype=torch.float32, device=self.device), self.p, self.seed, self.mask_id, self.step)


@_op("mul_add_philox", "(Tensor a, float p, int seed, int mask_id, Tensor? step_dev, int step, int mask_t_rows, Tensor? b, Tensor(a!) out) -> ()")
def _mul_add_philox(a, p, seed, mask_id, step_dev, step, mask_t_rows, b, out):
    m, c = a.shape
    check(_lib.load().emage_mul_add_philox(_ptr(a), _ld(a), p, seed & 0xFFFFFFFFFFFFFFFF, mask_id & 0xFFFFFFFF, _ptr(step_dev), step, mask_t_rows,
                                           _ptr(b), _ld(b) if b is not None else 0, _ptr(out), _ld(out), m, c, _stream()), "mul_add_philox")


def mul_add(a, mask, b=None, out=None, *, mask_t_rows=0):
    """out = a * mask (+ b) on fp32 (M, C) views: dropout with a given mask, and the residual add behind it.  mask_t_rows = T:
    the mask rows are stored (T, B) while a / b / out run (B, T).  mask may be a `PhiloxMask` (C % 4 == 0): drawn inside the kernel."""
    _dev(a)
    if isinstance(mask, PhiloxMask):
        assert mask.shape == tuple(a.shape) and a.shape[1] % 4 == 0
        out = torch.empty_like(a) if out is None else out
        seed = mask.seed & 0xFFFFFFFFFFFFFFFF
        if seed >= 1 << 63:           # the op schema's `int` is a signed 64-bit word: the same 64 bits in two's complement
            seed -= 1 << 64
        if torch.is_tensor(mask.step):
            _mul_add_philox(a, mask.p, seed, mask.mask_id, mask.step, 0, int(mask_t_rows), b, out)
        else:
            _mul_add_philox(a, mask.p, seed, mask.mask_id, None, int(mask.step), int(mask_t_rows), b, out)
        return out
    assert mask.dtype == torch.float32 and mask.dim() == 2 and mask.shape == a.shape and mask.stride(1) == 1
    out = torch.empty_like(a) if out is None else out
    _mul_add(a, mask, int(mask_t_rows), b, out)
    return out


LOSS_WORKSPACE_BYTES = 8192      # include/emage_hip.h


@_op("mse_loss", "(Tensor pred, Tensor target, float weight, Tensor(a!) loss, Tensor(b!) workspace) -> ()")
def _mse_loss(pred, target, weight, loss, workspace):
    m, c = pred.shape
    check(_lib.load().emage_mse_loss(_ptr(pred), _ld(pred), _ptr(target), _ld(target), m, c, weight, _ptr(loss), _ptr(workspace), _stream()), "mse_loss")


@_op("nll_loss", "(Tensor logits, Tensor index, float weight, Tensor(a!) loss, Tensor(b!) workspace) -> ()")
def _nll_loss(logits, index, weight, loss, workspace):
    m, k = logits.shape
    check(_lib.load().emage_nll_loss(_ptr(logits), _ld(logits), _ptr(index), m, k, weight, _ptr(loss), _ptr(workspace), _stream()), "nll_loss")


def loss_workspace(device):
    return torch.zeros(LOSS_WORKSPACE_BYTES // 8, dtype=torch.float64, device=device)


def loss_check(workspace):
    """Raise if an `nll_loss` call on this workspace met a class index outside [0, K) (torch's NLLLoss raises there); reads one
    word from the device, so call it where the losses are read back anyway."""
    if int(workspace.view(torch.int32)[-2]) != 0:
        raise EmageKernelError("nll_loss: class index out of range")


def mse_loss(pred, target, weight, loss, workspace):
    """loss[0] += weight * mean((pred - target)^2): fp32 (M, C) views, `loss` a float64 device scalar (F.mse_loss)."""
    _dev(pred)
    assert pred.shape == target.shape and pred.dtype == target.dtype == torch.float32 and loss.dtype == torch.float64
    _mse_loss(pred, target, float(weight), loss, workspace)


def nll_loss(logits, index, weight, loss, workspace):
    """loss[0] += weight * mean_m(-log_softmax(logits[m])[index[m]]): logits fp32 (M, K), index int64 (M,) (NLLLoss(log_softmax))."""
    _dev(logits)
    assert logits.dtype == torch.float32 and index.dtype == torch.int64 and index.numel() == logits.shape[0] and index.is_contiguous()
    _nll_loss(logits, index, float(weight), loss, workspace)


# ---- backward building blocks (include/emage_hip.h) --------------------------------------------------------------------
@_op("transpose", "(Tensor x, Tensor(a!) out) -> ()")
def _transpose(x, out):
    m, n = x.shape
    check(_lib.load().emage_transpose_f32(_ptr(x), _ld(x), _ptr(out), _ld(out), m, n, _stream()), "transpose")


def transpose(x, out=None):
    """(M, N) fp32 view -> contiguous (N, M)."""
    _dev(x)
    m, n = x.shape
    out = torch.empty(n, m, dtype=torch.float32, device=x.device) if out is None else out
    _transpose(x, out)
    return out


@_op("col_sum", "(Tensor x, Tensor? y, Tensor(a!)? out, bool accumulate, Tensor(b!) workspace) -> ()")
def _col_sum(x, y, out, accumulate, workspace):
    m, c = x.shape
    check(_lib.load().emage_col_sum(_ptr(x), _ld(x), _ptr(y), _ld(y) if y is not None else 0, m, c, _ptr(out), int(accumulate), _ptr(workspace),
                                    workspace.numel() * 8, _stream()), "col_sum")


@_op("col_sum_finalize_multi", "(Tensor(a!)[] outs, Tensor[] partials, int[] desc) -> ()")
def _col_sum_finalize_multi(outs, partials, desc):
    n = len(outs)
    arr = (_lib.FinalizeEntry * n)()
    for i in range(n):
        arr[i].partial, arr[i].out = partials[i].data_ptr(), outs[i].data_ptr()
        arr[i].chunks, arr[i].C, arr[i].accumulate = desc[3 * i], desc[3 * i + 1], desc[3 * i + 2]
    check(_lib.load().emage_col_sum_finalize_multi(arr, n, _stream()), "col_sum_finalize_multi")


class FinalizeQueue:
    """The finalize steps of chunked column reductions (bias gradients out of `grad_prep`, LayerNorm / embedding gradients out of
    `col_sum`), QUEUED and issued 64 at a time as one `emage_col_sum_finalize_multi` launch instead of one 3-48-block launch each (~600 per
    training step).  An entry whose destination overlaps a queued one flushes first, so every element still receives its contributions in
    issue order.  `flush()` before anything reads the destinations (training.TrainForward.flush_param_grads).  The launch carries its
    table by value in the kernel arguments: no device table, nothing whose lifetime a captured graph would depend on."""
    CAP = 64

    def __init__(self):
        self.entries = []

    def add(self, partial, chunks, c, out, accumulate):
        lo = out.data_ptr()
        hi = lo + ((out.numel() - 1) * out.stride(-1) + 1) * 4 if out.numel() else lo
        if any(lo < b and a < hi for a, b, *_ in self.entries):
            self.flush()
        assert out.dtype == torch.float32 and (out.dim() == 1 and out.stride(0) == 1 or out.numel() == 1), "destination: a contiguous fp32 vector"
        self.entries.append((lo, hi, partial, int(chunks), int(c), out, bool(accumulate)))
        if len(self.entries) >= self.CAP:
            self.flush()

    def flush(self):
        if not self.entries:
            return
        desc = []
        for _lo, _hi, _partial, chunks, c, _out, accumulate in self.entries:
            desc += [chunks, c, int(accumulate)]
        _col_sum_finalize_multi([e[5] for e in self.entries], [e[2] for e in self.entries], desc)
        self.entries = []


def col_sum_chunks(m):
    return _lib.load().emage_col_sum_chunks(int(m))


def col_sum(x, y=None, out=None, accumulate=False, defer=None):
    """out[c] (+)= sum_m x[m, c] (* y[m, c]) over fp32 (M, C) views.  defer (a `FinalizeQueue`, needs `out`): only the float64 chunk
    partials are computed now; the finalize is queued."""
    _dev(x)
    m, c = x.shape
    if out is None:                       # the finalize launch WRITES every column unless `accumulate`: no zero fill
        out, accumulate, defer = torch.empty(c, dtype=torch.float32, device=x.device), False, None
    ws = torch.empty(((m + 63) // 64) * c, dtype=torch.float64, device=x.device)         # one partial per chunk of >= 64 rows (csrc/train.hip STAT_CHUNK)
    if defer is not None:
        _col_sum(x, y, None, False, ws)
        defer.add(ws, col_sum_chunks(m), c, out, accumulate)
        return out
    _col_sum(x, y, out, accumulate, ws)
    return out


@_op("act_backward", "(Tensor dy, Tensor y, float slope, Tensor(a!) out) -> ()")
def _act_backward(dy, y, slope, out):
    m, c = dy.shape
    check(_lib.load().emage_act_backward(_ptr(dy), _ld(dy), _ptr(y), _ld(y), slope, _ptr(out), _ld(out), m, c, _stream()), "act_backward")


def act_backward(dy, y, slope, out=None):
    """dy * (y > 0 ? 1 : slope) from the activation's saved output."""
    _dev(dy)
    out = torch.empty(dy.shape, dtype=torch.float32, device=dy.device) if out is None else out
    _act_backward(dy, y, float(slope), out)
    return out


@_op("grad_prep", "(Tensor dy, Tensor? y, float slope, float scale, Tensor(a!)? out_h, Tensor(b!)? out_t, Tensor(c!)? bias_grad, bool accumulate, "
                  "Tensor(d!)? workspace) -> ()")
def _grad_prep(dy, y, slope, scale, out_h, out_t, bias_grad, accumulate, workspace):
    m, c = dy.shape
    check(_lib.load().emage_grad_prep(_ptr(dy), _ld(dy), _ptr(y), _ld(y) if y is not None else 0, slope, m, c, scale,
                                      _ptr(out_h), _ld(out_h) if out_h is not None else 0, out_h.shape[1] if out_h is not None else 0,
                                      _ptr(out_t), _ld(out_t) if out_t is not None else 0, out_t.shape[1] if out_t is not None else 0,
                                      _ptr(bias_grad), int(accumulate), _ptr(workspace), workspace.numel() * 8 if workspace is not None else 0, _stream()),
          "grad_prep")


def grad_prep(dy, y=None, slope=0.0, scale=1.0, n_store=None, m_store=None, bias_grad=None, accumulate=False, want_bias=True, defer=None):
    """One pass over the gradient dy (M, C) of a Linear's output -> (dpre_h, dpre_t, bias gradient): the EMAGE_H2 images of
    scale * dpre (M, n_store) and of its transpose (C, m_store), dpre = dy * (y > 0 ? 1 : slope) when the saved output `y` is given, and
    the column sums of dpre — written to / added to (`accumulate`) `bias_grad`, or to a fresh tensor.  n_store / m_store None: that image
    is not wanted.  Replaces act_backward + two h2_cast + col_sum (four reads of dy) in the split-fp16 backward."""
    _dev(dy)
    m, c = dy.shape
    out_h = torch.empty(m, n_store, dtype=torch.float32, device=dy.device) if n_store is not None else None
    out_t = torch.empty(c, m_store, dtype=torch.float32, device=dy.device) if m_store is not None else None
    ws = None
    if want_bias:
        if bias_grad is None:
            bias_grad, accumulate = torch.empty(c, dtype=torch.float32, device=dy.device), False
        ws = torch.empty(((m + 63) // 64) * c, dtype=torch.float64, device=dy.device)
    else:
        bias_grad = None
    if defer is not None and bias_grad is not None:      # the bias gradient's finalize is queued (`FinalizeQueue`): partials only in this launch
        _grad_prep(dy, y, float(slope), float(scale), out_h, out_t, None, False, ws)
        defer.add(ws, (m + 63) // 64, c, bias_grad, accumulate)
        return out_h, out_t, bias_grad
    _grad_prep(dy, y, float(slope), float(scale), out_h, out_t, bias_grad, bool(accumulate), ws)
    return out_h, out_t, bias_grad


@_op("layernorm_backward", "(Tensor x, Tensor gamma, Tensor dy, float eps, Tensor(a!) dx, Tensor(b!) dy_xhat) -> ()")
def _layernorm_backward(x, gamma, dy, eps, dx, dy_xhat):
    m, c = x.shape
    check(_lib.load().emage_layernorm_backward(_ptr(x), _ld(x), _ptr(gamma), _ptr(dy), _ld(dy), eps, _ptr(dx), _ld(dx), _ptr(dy_xhat), _ld(dy_xhat),
                                               m, c, _stream()), "layernorm_backward")


# LayerNorm backward as ONE kernel that also reduces the affine gradients (emage_layernorm_backward_affine, 16 rows per block) + a finalize
# launch, instead of dx / dy * xhat from the row kernel and two column sums: built, equal (tests), and SLOWER — the captured training step
# 103.5 -> 106.4 ms in an A/B on one box (profiles/r04_train_step_ab_fused_layernorm_backward.txt): 224 blocks of 4 waves walking 4 rows each
# are a longer dependent chain than 896 blocks of one row per wave plus two bandwidth-bound reductions.  With 4 rows per block (one per wave:
# the row kernel's parallelism, 4x the partials) it is still 0.6 ms behind the separate launches (105.0 vs 105.7 ms).  Off;
# tools/bench_train_step.py --ln-fused {0, 4, 16}
FUSED_LAYERNORM_BACKWARD = False         # False | 16 (= True) | 4: rows per block of the fused kernel


@_op("layernorm_backward_affine", "(Tensor x, Tensor gamma, Tensor dy, float eps, Tensor(a!) dx, Tensor(b!) dgamma, Tensor(c!) dbeta, bool accumulate, "
                                  "int rows_per_block, Tensor(d!) workspace) -> ()")
def _layernorm_backward_affine(x, gamma, dy, eps, dx, dgamma, dbeta, accumulate, rows_per_block, workspace):
    m, c = x.shape
    check(_lib.load().emage_layernorm_backward_affine(_ptr(x), _ld(x), _ptr(gamma), _ptr(dy), _ld(dy), eps, _ptr(dx), _ld(dx), _ptr(dgamma), _ptr(dbeta),
                                                      int(accumulate), m, c, rows_per_block, _ptr(workspace), workspace.numel() * 8, _stream()),
          "layernorm_backward_affine")


def layernorm_backward(x, gamma, dy, eps=1e-5, dgamma=None, dbeta=None, defer=None):
    """-> (dx, dgamma, dbeta) of LayerNorm(x) * gamma + beta for fp32 (M, C) rows.  dgamma / dbeta given: the affine gradients are ADDED
    to them (the parameter's gradient accumulator: no temporary, no separate add).  C <= 1024: dx and the float64 partials of both affine
    gradients come from ONE kernel (emage_layernorm_backward_affine) + a finalize launch; wider rows: dx, dy * xhat, two column sums."""
    _dev(x)
    m, c = x.shape
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    if c <= 1024 and FUSED_LAYERNORM_BACKWARD:
        accumulate = dgamma is not None
        if not accumulate:
            dgamma, dbeta = torch.empty(c, dtype=torch.float32, device=x.device), torch.empty(c, dtype=torch.float32, device=x.device)
        rows = 4 if int(FUSED_LAYERNORM_BACKWARD) == 4 else 16
        ws = torch.empty(((m + rows - 1) // rows) * 2 * c, dtype=torch.float64, device=x.device)
        _layernorm_backward_affine(x, gamma, dy, float(eps), dx, dgamma, dbeta, accumulate, rows, ws)
        return dx, dgamma, dbeta
    t = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _layernorm_backward(x, gamma, dy, float(eps), dx, t)
    if dgamma is None:
        return dx, col_sum(t), col_sum(dy)
    return dx, col_sum(t, out=dgamma, accumulate=True, defer=defer), col_sum(dy, out=dbeta, accumulate=True, defer=defer)


@_op("attention_backward", "(Tensor q, Tensor k, Tensor vt, int vt_rows, Tensor? pmask, Tensor d_out, Tensor(a!) dq, Tensor(b!) dk, Tensor(c!) dv, "
                           "int b, int h, int tq, int tk, int hd) -> ()")
def _attention_backward(q, k, vt, vt_rows, pmask, d_out, dq, dk, dv, b, h, tq, tk, hd):
    check(_lib.load().emage_attention_backward(_ptr(q), _ld(q), _ptr(k), _ld(k), _ptr(vt), vt.shape[-1], vt_rows, _ptr(pmask), _ptr(d_out), _ld(d_out),
                                               _ptr(dq), _ld(dq), _ptr(dk), _ld(dk), _ptr(dv), _ld(dv), b, h, tq, tk, hd, _stream()), "attention_backward")


def attention_backward(q, k, vt, vt_rows, pmask, d_out, dq, dk, dv, b, h, tq, tk, hd):
    """Backward of `attention` / `attention_dropout` (fp32 operands, same layouts): fills dq (B*Tq, >= h*hd), dk, dv (B*Tk, >= h*hd)."""
    _dev(q)
    _attention_backward(q, k, vt, vt_rows, pmask, d_out, dq, dk, dv, b, h, tq, tk, hd)


@_op("mse_loss_grad", "(Tensor pred, Tensor target, float weight, Tensor(a!) grad) -> ()")
def _mse_loss_grad(pred, target, weight, grad):
    m, c = pred.shape
    check(_lib.load().emage_mse_loss_grad(_ptr(pred), _ld(pred), _ptr(target), _ld(target), m, c, weight, _ptr(grad), _ld(grad), _stream()), "mse_loss_grad")


@_op("nll_loss_grad", "(Tensor logits, Tensor index, float weight, Tensor(a!) grad) -> ()")
def _nll_loss_grad(logits, index, weight, grad):
    m, k = logits.shape
    check(_lib.load().emage_nll_loss_grad(_ptr(logits), _ld(logits), _ptr(index), m, k, weight, _ptr(grad), _ld(grad), _stream()), "nll_loss_grad")


def mse_loss_grad(pred, target, weight):
    _dev(pred)
    g = torch.empty(pred.shape, dtype=torch.float32, device=pred.device)
    _mse_loss_grad(pred, target, float(weight), g)
    return g


def nll_loss_grad(logits, index, weight):
    _dev(logits)
    g = torch.empty(logits.shape, dtype=torch.float32, device=logits.device)
    _nll_loss_grad(logits, index, float(weight), g)
    return g


@_op("im2col_t", "(Tensor x, int c, int taps, int stride, int pad, int lin, int lout, int nseq, Tensor(a!) out) -> ()")
def _im2col_t(x, c, taps, stride, pad, lin, lout, nseq, out):
    check(_lib.load().emage_im2col_t(_ptr(x), _ld(x), c, taps, stride, pad, lin, lout, nseq, _ptr(out), out.stride(0), _stream()), "im2col_t")


def im2col_t(x, c, taps, stride, pad, lin, lout, nseq, mp):
    """-> (taps*c, mp) fp32, columns m = seq*lout + l (zero beyond nseq*lout): the transposed im2col of channels-last rows x."""
    _dev(x)
    out = torch.zeros(taps * c, mp, dtype=torch.float32, device=x.device)
    _im2col_t(x, c, taps, stride, pad, lin, lout, nseq, out)
    return out


@_op("im2col_t_h2", "(Tensor x, int c, int taps, int stride, int pad, int lin, int lout, int nseq, Tensor(a!) out) -> ()")
def _im2col_t_h2(x, c, taps, stride, pad, lin, lout, nseq, out):
    check(_lib.load().emage_im2col_t_h2(_ptr(x), _ld(x), c, taps, stride, pad, lin, lout, nseq, _ptr(out), out.stride(0), _stream()), "im2col_t_h2")


def im2col_t_h2(x, c, taps, stride, pad, lin, lout, nseq, mp):
    """`im2col_t` as an EMAGE_H2 image (taps*c, mp), mp = rup64(nseq*lout): the W operand of the split-fp16 dW contraction."""
    _dev(x)
    out = torch.empty(taps * c, mp, dtype=torch.float32, device=x.device)
    _im2col_t_h2(x, c, taps, stride, pad, lin, lout, nseq, out)
    return out


@_op("col2im", "(Tensor dcol, int c, int taps, int stride, int pad, int lin, int lout, int nseq, Tensor(a!) dx) -> ()")
def _col2im(dcol, c, taps, stride, pad, lin, lout, nseq, dx):
    check(_lib.load().emage_col2im(_ptr(dcol), dcol.stride(0), c, taps, stride, pad, lin, lout, nseq, _ptr(dx), _ld(dx), _stream()), "col2im")


def col2im(dcol, c, taps, stride, pad, lin, lout, nseq):
    """dcol (nseq*lout, taps*c) -> dx (nseq*lin, c): the adjoint of im2col."""
    _dev(dcol)
    dx = torch.empty(nseq * lin, c, dtype=torch.float32, device=dcol.device)
    _col2im(dcol, c, taps, stride, pad, lin, lout, nseq, dx)
    return dx


@_op("bn_backward", "(Tensor x, Tensor mean, Tensor var, Tensor gamma, float eps, Tensor dy, Tensor(a!) dx, Tensor(b!) dgamma, Tensor(c!) dbeta, "
                    "Tensor(d!) workspace) -> ()")
def _bn_backward(x, mean, var, gamma, eps, dy, dx, dgamma, dbeta, workspace):
    m, c = x.shape
    check(_lib.load().emage_bn_backward(_ptr(x), _ld(x), _ptr(mean), _ptr(var), _ptr(gamma), eps, _ptr(dy), _ld(dy), _ptr(dx), _ld(dx), _ptr(dgamma),
                                        _ptr(dbeta), m, c, _ptr(workspace), workspace.numel() * 8, _stream()), "bn_backward")


def bn_backward(x, stats, gamma, dy, eps=1e-5):
    """Training-mode BatchNorm backward on fp32 (M, C) views -> (dx, dgamma, dbeta); stats = (mean, biased var) of the forward."""
    _dev(x)
    m, c = x.shape
    nbytes = _lib.load().emage_bn_stats_workspace_bytes(m, c)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=x.device)
    dx = torch.empty(m, c, dtype=torch.float32, device=x.device)
    dg, db = torch.empty(c, dtype=torch.float32, device=x.device), torch.empty(c, dtype=torch.float32, device=x.device)
    _bn_backward(x, stats[0], stats[1], gamma, float(eps), dy, dx, dg, db, ws)
    return dx, dg, db


@_op("bn_backward_sums", "(Tensor x, Tensor mean, Tensor var, float eps, Tensor dy, Tensor(a!) sum_dy_xhat, Tensor(b!) sum_dy, Tensor(c!) workspace) -> ()")
def _bn_backward_sums(x, mean, var, eps, dy, sum_dy_xhat, sum_dy, workspace):
    m, c = x.shape
    check(_lib.load().emage_bn_backward_sums(_ptr(x), _ld(x), _ptr(mean), _ptr(var), eps, _ptr(dy), _ld(dy), _ptr(sum_dy_xhat), _ptr(sum_dy), m, c,
                                             _ptr(workspace), workspace.numel() * 8, _stream()), "bn_backward_sums")


@_op("bn_backward_apply", "(Tensor x, Tensor mean, Tensor var, Tensor gamma, float eps, Tensor dy, Tensor sum_dy_xhat, Tensor sum_dy, int count, Tensor(a!) dx) -> ()")
def _bn_backward_apply(x, mean, var, gamma, eps, dy, sum_dy_xhat, sum_dy, count, dx):
    m, c = x.shape
    check(_lib.load().emage_bn_backward_apply(_ptr(x), _ld(x), _ptr(mean), _ptr(var), _ptr(gamma), eps, _ptr(dy), _ld(dy), _ptr(sum_dy_xhat), _ptr(sum_dy),
                                              count, _ptr(dx), _ld(dx), m, c, _stream()), "bn_backward_apply")


def bn_backward_sums(x, stats, dy, eps=1e-5):
    """-> (sum_m dy * xhat, sum_m dy) per channel over THIS rank's rows (SyncBatchNorm all-reduces them before `bn_backward_apply`)."""
    _dev(x)
    m, c = x.shape
    nbytes = _lib.load().emage_bn_stats_workspace_bytes(m, c)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=x.device)
    sdx, sd = torch.empty(c, dtype=torch.float32, device=x.device), torch.empty(c, dtype=torch.float32, device=x.device)
    _bn_backward_sums(x, stats[0], stats[1], float(eps), dy, sdx, sd, ws)
    return sdx, sd


def bn_backward_apply(x, stats, gamma, dy, sums, count, eps=1e-5):
    """dx of training-mode BatchNorm from (global) sums over `count` rows."""
    _dev(x)
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _bn_backward_apply(x, stats[0], stats[1], gamma, float(eps), dy, sums[0], sums[1], int(count), dx)
    return dx


@_op("wav_conv_in_backward", "(Tensor dy, Tensor wav, int lout, int taps, int stride, int pad, Tensor(a!) dw, Tensor(b!) workspace) -> ()")
def _wav_conv_in_backward(dy, wav, lout, taps, stride, pad, dw, workspace):
    b, l = wav.shape
    check(_lib.load().emage_wav_conv_in_backward(_ptr(dy), _ld(dy), _ptr(wav), wav.stride(0), l, b, lout, dy.shape[1], taps, stride, pad, _ptr(dw),
                                                 _ptr(workspace), workspace.numel() * 8, _stream()), "wav_conv_in_backward")


def wav_conv_in_backward(dy, wav, lout, taps, stride, pad):
    """Weight gradient (C, taps) of the first WavEncoder layer for output gradient dy (B*lout, C) and the waveform wav (B, L)."""
    _dev(dy)
    m, c = dy.shape
    nbytes = _lib.load().emage_wav_conv_in_backward_workspace_bytes(m, c, taps)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=dy.device)
    dw = torch.empty(c, taps, dtype=torch.float32, device=dy.device)
    _wav_conv_in_backward(dy, wav, lout, taps, stride, pad, dw, ws)
    return dw


@_op("count_nonfinite", "(Tensor x, Tensor(a!) counter) -> ()")
def _count_nonfinite(x, counter):
    check(_lib.load().emage_count_nonfinite(_ptr(x), x.numel(), _ptr(counter), _stream()), "count_nonfinite")


@_op("count_nonfinite_multi", "(Tensor[] xs, Tensor(a!) counter) -> ()")
def _count_nonfinite_multi(xs, counter):
    n = len(xs)
    ptrs, sizes = (C.c_void_p * n)(*[x.data_ptr() for x in xs]), (C.c_long * n)(*[x.numel() for x in xs])
    check(_lib.load().emage_count_nonfinite_multi(ptrs, sizes, n, _ptr(counter), _stream()), "count_nonfinite_multi")


COUNT_NONFINITE_MAX = 16


def count_nonfinite_multi(xs, counter):
    """counter[0] += the number of inf / NaN among the contiguous fp32 tensors `xs`, one launch per 16 of them."""
    xs = [x for x in xs if x.numel()]
    for x in xs:
        _dev(x)
        assert x.dtype == torch.float32 and x.is_contiguous()
    for i in range(0, len(xs), COUNT_NONFINITE_MAX):
        _count_nonfinite_multi(xs[i:i + COUNT_NONFINITE_MAX], counter)


def count_nonfinite(x, counter):
    """counter (int32, 1 element) += number of inf / NaN values in the contiguous fp32 tensor x."""
    _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and counter.dtype == torch.int32
    _count_nonfinite(x, counter)


@_op("adam_step", "(Tensor(a!) param, Tensor grad, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, int step, float lr, float beta1, float beta2, float eps, "
                  "float weight_decay) -> ()")
def _adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay):
    check(_lib.load().emage_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), step, lr, beta1, beta2, eps, weight_decay,
                                      _stream()), "adam_step")


@_op("adam_step_dev", "(Tensor(a!) param, Tensor grad, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, Tensor step, float lr, float beta1, float beta2, float eps, "
                      "float weight_decay) -> ()")
def _adam_step_dev(param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay):
    check(_lib.load().emage_adam_step_dev(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), _ptr(step), lr, beta1, beta2, eps,
                                          weight_decay, _stream()), "adam_step_dev")


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr=1.5e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam's update of one contiguous fp32 parameter tensor, in place (param, exp_avg, exp_avg_sq).  `step`: the
    1-based step count as a Python int, or a one-element int32 device tensor (read by the kernel: for hipGraph-captured steps)."""
    _dev(param)
    for t in (param, grad, exp_avg, exp_avg_sq):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == param.numel()
    if torch.is_tensor(step):
        assert step.dtype == torch.int32 and step.numel() == 1
        _adam_step_dev(param, grad, exp_avg, exp_avg_sq, step, float(lr), float(beta1), float(beta2), float(eps), float(weight_decay))
    else:
        _adam_step(param, grad, exp_avg, exp_avg_sq, int(step), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay))


@_op("dropout_mask", "(Tensor(a!) out, float p, int seed, int mask_id, Tensor? step_dev, int step) -> ()")
def _dropout_mask(out, p, seed, mask_id, step_dev, step):
    check(_lib.load().emage_dropout_mask(_ptr(out), out.numel(), p, seed & 0xFFFFFFFFFFFFFFFF, mask_id & 0xFFFFFFFF, _ptr(step_dev), step, _stream()), "dropout_mask")


def dropout_mask(out, p, seed, mask_id, step):
    """Fill the contiguous fp32 tensor `out` with a dropout keep mask bernoulli(1 - p) / (1 - p) drawn on the device (Philox4x32-10 keyed
    by `seed`, counter (element / 4, mask_id, step)); `step` an int or a one-element int32 device tensor (read by the kernel: captured
    training steps draw fresh masks on every replay).  See include/emage_hip.h: emage_dropout_mask."""
    _dev(out)
    assert out.dtype == torch.float32 and out.is_contiguous()
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    if seed >= 1 << 63:           # the op schema's `int` is a signed 64-bit word: pass the same 64 bits in two's complement
        seed -= 1 << 64
    if torch.is_tensor(step):
        assert step.dtype == torch.int32 and step.numel() == 1
        _dropout_mask(out, float(p), seed, int(mask_id), step, 0)
    else:
        _dropout_mask(out, float(p), seed, int(mask_id), None, int(step))
    return out


def philox_dropout_reference(n, p, seed, mask_id, step):
    """numpy restatement of emage_dropout_mask (Philox4x32-10, csrc/train.hip) — tests and the CPU stand-ins."""
    import numpy as np
    nb = (n + 3) // 4
    b = np.arange(nb, dtype=np.uint64)
    c = [(b & np.uint64(0xFFFFFFFF)).astype(np.uint64), (b >> np.uint64(32)).astype(np.uint64),
         np.full(nb, mask_id & 0xFFFFFFFF, dtype=np.uint64), np.full(nb, step & 0xFFFFFFFF, dtype=np.uint64)]
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c[0], np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & m32, p1 & m32, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & m32, p0 & m32]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & m32, (k1 + np.uint64(0xBB67AE85)) & m32
    words = np.stack(c, axis=1).reshape(-1)[:n]
    u = (words >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return np.where(u >= np.float32(p), np.float32(1.0 / (1.0 - p)), np.float32(0.0)).astype(np.float32)


class AdamTable:
    """Descriptor tables of `adam_multi` for a fixed list of (param, grad, exp_avg, exp_avg_sq) fp32 tensors (contiguous, same numel per
    quadruple): built once, valid as long as the tensors' storage does not move."""

    def __init__(self, quads, device):
        chunk = _lib.load().emage_adam_multi_chunk()
        rows, bt, bc = [], [], []
        for i, (p, g, m, v) in enumerate(quads):
            n = p.numel()
            for t in (p, g, m, v):
                assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n and t.device == p.device
            rows.append([p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n])
            nb = (n + chunk - 1) // chunk
            bt += [i] * nb
            bc += list(range(nb))
        self.table = torch.tensor(rows, dtype=torch.int64).to(device)
        self.block_tensor = torch.tensor(bt, dtype=torch.int32).to(device)
        self.block_chunk = torch.tensor(bc, dtype=torch.int32).to(device)
        self.n_blocks = len(bt)
        self.keep = quads                      # the tensors the table points at stay alive with it


@_op("adam_multi", "(Tensor table, Tensor block_tensor, Tensor block_chunk, int n_blocks, Tensor? step_dev, int step, float lr, float beta1, float beta2, "
                   "float eps, float weight_decay, float grad_scale, bool zero_grad, Tensor? skip) -> ()")
def _adam_multi(table, block_tensor, block_chunk, n_blocks, step_dev, step, lr, beta1, beta2, eps, weight_decay, grad_scale, zero_grad, skip):
    check(_lib.load().emage_adam_multi(_ptr(table), _ptr(block_tensor), _ptr(block_chunk), n_blocks, _ptr(step_dev), step, lr, beta1, beta2, eps,
                                       weight_decay, grad_scale, int(zero_grad), _ptr(skip), _stream()), "adam_multi")


def adam_multi(tab: "AdamTable", step, lr=1.5e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0, zero_grad=False, skip=None):
    """torch.optim.Adam's update of every tensor of `tab` in ONE launch (include/emage_hip.h: emage_adam_multi); `step` an int or a one-element
    int32 device tensor.  skip: one int32 on the device — non-zero (a count of non-finite gradient words) leaves parameters and moments
    untouched (the gradients are still cleared when zero_grad)."""
    _dev(tab.table)
    if skip is not None:
        assert skip.dtype == torch.int32 and skip.numel() == 1 and skip.is_cuda
    if torch.is_tensor(step):
        _adam_multi(tab.table, tab.block_tensor, tab.block_chunk, tab.n_blocks, step, 0, float(lr), float(beta1), float(beta2), float(eps),
                    float(weight_decay), float(grad_scale), bool(zero_grad), skip)
    else:
        _adam_multi(tab.table, tab.block_tensor, tab.block_chunk, tab.n_blocks, None, int(step), float(lr), float(beta1), float(beta2), float(eps),
                    float(weight_decay), float(grad_scale), bool(zero_grad), skip)


@_op("layernorm", "(int dtype, Tensor x, Tensor gamma, Tensor beta, float eps, Tensor? add, Tensor(a!)? y_f32, Tensor(b!)? y) -> ()")
def _layernorm(dtype, x, gamma, beta, eps, add, y_f32, y):
    m, c = x.shape
    ldy = _ld(y_f32) if y_f32 is not None else _ld(y)
    check(_lib.load().emage_layernorm(dtype, _ptr(x), _ld(x), _ptr(gamma), _ptr(beta), eps, _ptr(add), _ld(add) if add is not None else 0,
                                      _ptr(y_f32), _ptr(y), ldy, m, c, _stream()), "layernorm")


def layernorm(dtype, x, gamma, beta, eps=1e-5, add=None, y_f32=None, y=None):
    """x / add / y are in `dtype` (the residual stream's storage type)."""
    _dev(x)
    m, c = x.shape
    assert x.dtype == TORCH_DTYPE[dtype] and (add is None or add.dtype == x.dtype)
    if y_f32 is not None and y is not None:
        assert _ld(y) == _ld(y_f32)
    _layernorm(dtype, x, gamma, beta, float(eps), add, y_f32, y)


@_op("add", "(int dtype, Tensor a, Tensor b, Tensor? c, Tensor(a!)? out_f32, Tensor(b!)? out, int mod_b, int mod_c, int f32_mask) -> ()")
def _add(dtype, a, b, c, out_f32, out, mod_b, mod_c, f32_mask):
    m, n = a.shape
    ldo = _ld(out_f32) if out_f32 is not None else _ld(out)
    check(_lib.load().emage_add(dtype, _ptr(a), _ld(a), _ptr(b), _ld(b), mod_b, _ptr(c), _ld(c) if c is not None else 0, mod_c,
                                f32_mask, _ptr(out_f32), _ptr(out), ldo, m, n, _stream()), "add")


def add(dtype, a, b, c=None, out_f32=None, out=None, mod_b=0, mod_c=0, h2_operands=()):
    """out = a + b[m % mod_b] (+ c); each operand may be fp32 or `dtype` (detected from the tensor; dtype H2: every operand
    is float32 unless its position (0 = a, 1 = b, 2 = c) is listed in `h2_operands`; `out` is then an H2 image)."""
    _dev(a)
    m, n = a.shape
    mask = 0
    for bit, t in enumerate((a, b, c)):
        if t is not None:
            assert t.dtype in (torch.float32, TORCH_DTYPE[dtype])
            if dtype & 0xff == H2:
                mask |= 0 if bit in h2_operands else (1 << bit)
            else:
                mask |= (1 << bit) if t.dtype == torch.float32 else 0
    if out_f32 is not None and out is not None:
        assert _ld(out) == _ld(out_f32)
    _add(dtype, a, b, c, out_f32, out, mod_b, mod_c, mask)


@_op("pack_motion", "(int dtype, Tensor motion, Tensor mask, Tensor emb, Tensor? seed, Tensor(a!) out) -> ()")
def _pack_motion(dtype, motion, mask, emb, seed, out):
    b, t, c = motion.shape
    ldb = motion.stride(0) if b > 1 else t * c
    pre, ld_seed = 0, 0
    if seed is not None:
        pre = seed.shape[1]
        ld_seed = seed.stride(0) if b > 1 else pre * c
    check(_lib.load().emage_pack_motion(dtype, _ptr(motion), _ptr(mask), ldb, _ptr(emb), _ptr(seed), ld_seed, pre,
                                        _ptr(out), _ld(out), out.shape[1], b, t, c, _stream()), "pack_motion")


def pack_motion(dtype, motion, mask, emb, n_store, seed=None):
    """motion / mask: (B, T, C) fp32 views with contiguous frames (strides (ldb, C, 1), the same ldb for both): a window
    of a longer clip tensor is read in place.  seed: optional (B, pre, C) view spliced into the first `pre` frames
    where the mask is set (inference()'s seed carry-over, M:386-391).  -> (B*T, n_store) in `dtype`."""
    _dev(motion)
    b, t, c = motion.shape
    for x in (motion, mask):
        assert x.dtype == torch.float32 and x.shape == (b, t, c) and x.stride(2) == 1 and (x.stride(1) == c or t == 1), (x.shape, x.stride())
    assert b == 1 or mask.stride(0) == motion.stride(0)
    if seed is not None:
        pre = seed.shape[1]
        assert seed.dtype == torch.float32 and seed.shape == (b, pre, c) and seed.stride(2) == 1 and (seed.stride(1) == c or pre == 1)
    out = torch.empty(b * t, n_store, dtype=TORCH_DTYPE[dtype], device=motion.device)
    _pack_motion(dtype, motion, mask, emb, seed, out)
    return out


@_op("cast_pad", "(int dtype, Tensor src, Tensor(a!) out) -> ()")
def _cast_pad(dtype, src, out):
    m, c = src.shape
    check(_lib.load().emage_cast_pad(dtype, _ptr(src), _ld(src), _ptr(out), _ld(out), out.shape[1], m, c, _stream()), "cast_pad")


def cast_pad(dtype, src2d, n_store, out=None):
    """fp32 (M, C) view -> (M, n_store) in `dtype` with a zero tail.  dtype H2 with `out` aliasing `src2d` (same view) converts
    in place."""
    _dev(src2d)
    m, c = src2d.shape
    if out is None:
        out = torch.empty(m, n_store, dtype=TORCH_DTYPE[dtype], device=src2d.device)
    else:
        assert out.shape == (m, n_store) and out.stride(1) == 1
    _cast_pad(dtype, src2d, out)
    return out


@_op("h2_cast", "(Tensor src, Tensor(a!) out, float scale, bool transpose) -> ()")
def _h2_cast(src, out, scale, transpose):
    m, c = src.shape
    check(_lib.load().emage_h2_cast(_ptr(src), _ld(src), _ptr(out), _ld(out), out.shape[1], m, c, scale, int(transpose), _stream()), "h2_cast")


def h2_cast(src2d, n_store, scale=1.0, transpose=False):
    """fp32 (M, C) view -> the EMAGE_H2 image of src * scale (a power of two): (M, n_store), or transposed (C, n_store) with
    out[c][m] = src[m][c] * scale; zero tails.  The operands of the training step's backward contractions."""
    _dev(src2d)
    m, c = src2d.shape
    out = torch.empty(c if transpose else m, n_store, dtype=torch.float32, device=src2d.device)
    _h2_cast(src2d, out, float(scale), bool(transpose))
    return out


@_op("f16x3_pack_weights", "(Tensor w, Tensor(a!) out, float scale) -> ()")
def _f16x3_pack_weights(w, out, scale):
    n, k = w.shape
    check(_lib.load().emage_f16x3_pack_weights(_ptr(w), _ld(w), _ptr(out), _ld(out), n, k, scale, _stream()), "f16x3_pack_weights")


def pack_f16x3_weights(w2d, scale):
    """fp32 (N, K) device weights, K % 32 == 0 -> the EMAGE_F16X3 operand image of w * scale (what `split_f16_weights` builds with
    tensor arithmetic), float32-typed (N, K)."""
    _dev(w2d)
    w = w2d if (w2d.stride(1) == 1 and w2d.stride(0) % 4 == 0 and w2d.data_ptr() % 16 == 0) else w2d.contiguous()
    out = torch.empty(w.shape[0], w.shape[1], dtype=torch.float32, device=w.device)
    _f16x3_pack_weights(w, out, float(scale))
    return out


@_op("rot6d_to_axis_angle", "(Tensor rot6d, Tensor(a!) out) -> ()")
def _rot6d_to_axis_angle(x, out):
    check(_lib.load().emage_rot6d_to_axis_angle(_ptr(x), _ptr(out), x.numel() // 6, _stream()), "rot6d_to_axis_angle")


@_op("axis_angle_to_rot6d", "(Tensor aa, Tensor(a!) out) -> ()")
def _axis_angle_to_rot6d(x, out):
    check(_lib.load().emage_axis_angle_to_rot6d(_ptr(x), _ptr(out), x.numel() // 3, _stream()), "axis_angle_to_rot6d")


def rot6d_to_axis_angle(rot6d):
    _dev(rot6d)
    x = rot6d.contiguous().float()
    out = torch.empty(x.shape[:-1] + (3,), dtype=torch.float32, device=x.device)
    _rot6d_to_axis_angle(x, out)
    return out


def axis_angle_to_rot6d(aa):
    _dev(aa)
    x = aa.contiguous().float()
    out = torch.empty(x.shape[:-1] + (6,), dtype=torch.float32, device=x.device)
    _axis_angle_to_rot6d(x, out)
    return out


@_op("merge_parts", "(Tensor? face, Tensor? upper, Tensor? hands, Tensor? lower, Tensor(a!) aa, Tensor(b!)? motion, Tensor(c!) expr) -> ()")
def _merge_parts(face, upper, hands, lower, aa, motion, expr):
    ld = lambda t: _ld(t) if t is not None else 0
    check(_lib.load().emage_merge_parts(_ptr(face), ld(face), _ptr(upper), ld(upper), _ptr(hands), ld(hands), _ptr(lower), ld(lower),
                                        _ptr(aa), _ptr(motion), _ptr(expr), aa.shape[0], _stream()), "merge_parts")


def merge_parts(face, upper, hands, lower, m, device, want_motion=True):
    """2-D fp32 part tensors (M, ld) or None -> (axis_angle (M,165), motion (M,337), expression (M,100))."""
    aa = torch.empty(m, 165, dtype=torch.float32, device=device)
    motion = torch.empty(m, 337, dtype=torch.float32, device=device) if want_motion else None
    expr = torch.empty(m, 100, dtype=torch.float32, device=device)
    _merge_parts(face, upper, hands, lower, aa, motion, expr)
    return aa, motion, expr


@_op("velocity_to_position", "(Tensor vel, int col0, Tensor init, float dt, Tensor(a!) trans) -> ()")
def _velocity_to_position(vel, col0, init, dt, trans):
    b, t, _ = trans.shape
    ld_init = 0 if init.shape[0] == 1 and b > 1 else init.stride(0)
    check(_lib.load().emage_velocity_to_position(_ptr(vel), _ld(vel), col0, _ptr(init), ld_init, dt, _ptr(trans), b, t, _stream()), "velocity_to_position")


def velocity_to_position(vel2d, col0, init, dt, b, t):
    """init: (B, 3) fp32 view (any clip stride) or (1, 3) = one start position for every clip (clip stride 0)."""
    _dev(vel2d)
    assert init.dtype == torch.float32 and init.dim() == 2 and init.shape[1] == 3 and init.stride(1) == 1 and init.shape[0] in (1, b)
    trans = torch.empty(b, t, 3, dtype=torch.float32, device=vel2d.device)
    _velocity_to_position(vel2d, col0, init, float(dt), trans)
    return trans


# ---- DisCo / CaMN (include/emage_hip.h: emage_lstm_step ...) -------------------------------------------------------------
@_op("lstm_step", "(int dtype, Tensor h_prev, Tensor w_hh, Tensor gates_x, Tensor(a!) cstate, Tensor(b!) h_out, float w_scale, float a_scale) -> ()")
def _lstm_step(dtype, h_prev, w_hh, gates_x, cstate, h_out, w_scale, a_scale):
    b, h = cstate.shape
    ld = lambda t: t.stride(0) if b > 1 else max(t.shape[1], t.stride(0))
    check(_lib.load().emage_lstm_step(dtype, _ptr(h_prev), ld(h_prev), _ptr(w_hh), w_scale, a_scale,
                                      _ptr(gates_x), ld(gates_x), _ptr(cstate), ld(cstate), _ptr(h_out), ld(h_out), b, h, _stream()), "lstm_step")


def lstm_step(dtype, h_prev, w_hh, gates_x, cstate, h_out, *, w_scale=1.0, a_scale=None):
    """One time step of one LSTM direction for the whole batch.  h_prev (B, H) / gates_x (B, 4H) / h_out (B, H): fp32 row
    views with unit column stride (step t of a (B, T, .) tensor is a strided view); cstate (B, H) updated in place."""
    _dev(h_prev)
    b, h = cstate.shape
    for t in (h_prev, gates_x, cstate, h_out):
        assert t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.shape[0] == b
    assert h_prev.shape[1] == h and h_out.shape[1] == h and gates_x.shape[1] == 4 * h
    _lstm_step(dtype, h_prev, w_hh, gates_x, cstate, h_out, float(w_scale), float(A_SCALE_F16X3 if a_scale is None else a_scale))


@_op("lstm_step_pair", "(int dtype, Tensor h_prev0, Tensor h_prev1, Tensor w_hh0, Tensor w_hh1, Tensor gates_x0, Tensor gates_x1, Tensor(a!) cstate0, "
                       "Tensor(b!) cstate1, Tensor(c!) h_out0, Tensor(d!) h_out1, float w_scale0, float w_scale1, float a_scale) -> ()")
def _lstm_step_pair(dtype, h_prev0, h_prev1, w_hh0, w_hh1, gates_x0, gates_x1, cstate0, cstate1, h_out0, h_out1, w_scale0, w_scale1, a_scale):
    b, h = cstate0.shape
    ld = lambda t: t.stride(0) if b > 1 else max(t.shape[1], t.stride(0))
    assert ld(gates_x0) == ld(gates_x1) and ld(cstate0) == ld(cstate1) and ld(h_out0) == ld(h_out1)
    check(_lib.load().emage_lstm_step_pair(dtype, _ptr(h_prev0), _ptr(h_prev1), ld(h_prev0), ld(h_prev1), _ptr(w_hh0), _ptr(w_hh1), w_scale0, w_scale1,
                                           a_scale, _ptr(gates_x0), _ptr(gates_x1), ld(gates_x0), _ptr(cstate0), _ptr(cstate1), ld(cstate0),
                                           _ptr(h_out0), _ptr(h_out1), ld(h_out0), b, h, _stream()), "lstm_step_pair")


def lstm_step_pair(dtype, fwd, bwd, *, a_scale=None):
    """One time step of BOTH directions of a layer in one launch.  fwd / bwd: (h_prev, w_hh, gates_x, cstate, h_out, w_scale)
    of the forward direction at its step and of the backward direction at its step (see `lstm_step`)."""
    _dev(fwd[0])
    _lstm_step_pair(dtype, fwd[0], bwd[0], fwd[1], bwd[1], fwd[2], bwd[2], fwd[3], bwd[3], fwd[4], bwd[4], float(fwd[5]), float(bwd[5]),
                    float(A_SCALE_F16X3 if a_scale is None else a_scale))


LSTM_SYNC_WORDS_PER_LAUNCH, LSTM_SYNC_ERROR_WORD = 544, 512      # include/emage_hip.h


def lstm_layer_supported(dtype, hidden):
    """Whether `lstm_layer` (the persistent recurrence) takes this layer on the CURRENT device: split-f16 precision, H = 256 or
    512, and enough co-resident blocks (2 * H/16 CUs; a compute partition or a CU-masked queue may have fewer — the per-step
    launches produce the same bits there)."""
    return dtype == F16X3 and hidden in (256, 512) and _lib.load().emage_lstm_layer_sync_words(1, int(hidden)) > 0


def lstm_layer_sync(b, hidden, device):
    """The int32 scratch tensor `lstm_layer` needs for a batch of b clips (arrival counters + error words)."""
    n = _lib.load().emage_lstm_layer_sync_words(int(b), int(hidden))
    if n <= 0:
        raise EmageKernelError(f"lstm_layer: unsupported batch / hidden size ({b}, {hidden})")
    return torch.zeros(n, dtype=torch.int32, device=device)


@_op("lstm_layer", "(int dtype, Tensor gates_x, Tensor w_hh0, Tensor w_hh1, float w_scale0, float w_scale1, float a_scale, Tensor(a!) hseq, "
                   "Tensor(b!) sync) -> ()")
def _lstm_layer(dtype, gates_x, w_hh0, w_hh1, w_scale0, w_scale1, a_scale, hseq, sync):
    b, t, h2 = hseq.shape
    assert gates_x.dim() == 3 and gates_x.shape[0] == b and gates_x.shape[1] == t and gates_x.stride(2) == 1 and hseq.stride(2) == 1
    check(_lib.load().emage_lstm_layer(dtype, _ptr(gates_x), gates_x.stride(0), gates_x.stride(1), _ptr(w_hh0), _ptr(w_hh1), w_scale0, w_scale1, a_scale,
                                       _ptr(hseq), hseq.stride(0), hseq.stride(1), b, t, h2 // 2, _ptr(sync), sync.numel(), _stream()), "lstm_layer")


def lstm_layer(dtype, gates_x, w_hh, w_scale, hseq, sync, *, a_scale=None):
    """The whole recurrence of one bidirectional LSTM layer (zero initial state) in one launch per <= 256 clips:
    gates_x (B, T, 8H) fp32 (input projection incl. biases, columns dir * 4H + 4u + g), w_hh / w_scale the two directions'
    packed recurrent weights, hseq (B, T, 2H) fp32 out.  Bit-identical to T `lstm_step_pair` launches.  `sync` from
    `lstm_layer_sync`; call `lstm_layer_check(sync)` once the stream is synchronised."""
    _dev(gates_x)
    _lstm_layer(dtype, gates_x, w_hh[0], w_hh[1], float(w_scale[0]), float(w_scale[1]), float(A_SCALE_F16X3 if a_scale is None else a_scale), hseq, sync)
    return hseq


@_op("lstm_layer_health", "(Tensor sync, Tensor(a!) counter) -> ()")
def _lstm_layer_health(sync, counter):
    check(_lib.load().emage_lstm_layer_health(_ptr(sync), sync.numel(), _ptr(counter), _stream()), "lstm_layer_health")


def lstm_layer_health(sync, counter):
    """counter (int32 device scalar) += number of `lstm_layer` launches recorded in `sync` that lost a block; stream-ordered behind
    them (capturable): the runners read the counter with the D2H copy they do anyway, on EVERY replay."""
    _dev(sync)
    assert sync.dtype == torch.int32 and counter.dtype == torch.int32
    _lstm_layer_health(sync, counter)


def lstm_layer_check(sync):
    """Raise if a block of an `lstm_layer` launch gave up waiting for its group (reads the error words: synchronises)."""
    if bool(sync.view(-1, LSTM_SYNC_WORDS_PER_LAUNCH)[:, LSTM_SYNC_ERROR_WORD].any()):
        raise EmageKernelError("emage_lstm_layer: a block timed out waiting for its group (blocks not co-resident?); the layer output is invalid")


@_op("softmax2_mix", "(Tensor sel, Tensor c1, Tensor c2, Tensor(a!) out) -> ()")
def _softmax2_mix(sel, c1, c2, out):
    m, c = c1.shape
    check(_lib.load().emage_softmax2_mix(_ptr(sel), _ld(sel), _ptr(c1), _ld(c1), _ptr(c2), _ld(c2), _ptr(out), _ld(out), m, c, _stream()),
          "softmax2_mix")


def softmax2_mix(sel, c1, c2, out):
    """out = softmax(sel[:, :2])[0] * c1 + softmax(sel[:, :2])[1] * c2   (fp32 2-D views)."""
    _dev(sel)
    m, c = c1.shape
    _softmax2_mix(sel, c1, c2, out)
    return out


@_op("lstm_inputs", "(Tensor(a!) out, Tensor? speaker_table, Tensor? speaker_id, Tensor? seed_motion, int pose_dims, int seed_frames, "
                    "Tensor src_map, int b, int t) -> ()")
def _lstm_inputs(out, speaker_table, speaker_id, seed_motion, pose_dims, seed_frames, src_map, b, t):
    f = 0 if speaker_table is None else speaker_table.shape[1]
    ld_seed = 0
    if seed_motion is not None:
        ld_seed = seed_motion.stride(0) if b > 1 else seed_motion.shape[1] * pose_dims
    check(_lib.load().emage_lstm_inputs(_ptr(speaker_table), _ptr(speaker_id), f, _ptr(seed_motion), ld_seed, pose_dims, seed_frames,
                                        _ptr(src_map), _ptr(out), _ld(out), out.shape[1], b, t, _stream()), "lstm_inputs")


def lstm_inputs(out, speaker_table, speaker_id, seed_motion, pose_dims, seed_frames, src_map, b, t):
    """Fill `out` (B*T, n_store) = a column block of the LSTM input rows with [speaker | seed | is-seed | 0...]."""
    _dev(out)
    f = 0 if speaker_table is None else speaker_table.shape[1]
    ld_seed = 0
    if seed_motion is not None:
        assert seed_motion.dtype == torch.float32 and seed_motion.dim() == 3 and seed_motion.shape[2] == pose_dims
        assert seed_motion.stride(2) == 1 and seed_motion.stride(1) == pose_dims
        ld_seed = seed_motion.stride(0) if b > 1 else seed_motion.shape[1] * pose_dims
    assert src_map.dtype == torch.int32 and src_map.numel() == t
    _lstm_inputs(out, speaker_table, speaker_id, seed_motion, pose_dims, seed_frames, src_map, b, t)
    return out


@_op("rot6d_scatter", "(Tensor rot6d, Tensor slot_of_joint, Tensor(a!) out) -> ()")
def _rot6d_scatter(rot6d, slot_of_joint, out):
    m = rot6d.shape[0]
    check(_lib.load().emage_rot6d_scatter(_ptr(rot6d), _ld(rot6d), _ptr(slot_of_joint), _ptr(out), m, slot_of_joint.numel(), _stream()), "rot6d_scatter")


def rot6d_scatter(rot6d2d, slot_of_joint, n_joints=55):
    """rot6d2d (M, n_sel*6) fp32 view -> (M, n_joints*3) axis-angle, joint j from slot slot_of_joint[j] (or zeros)."""
    _dev(rot6d2d)
    m = rot6d2d.shape[0]
    out = torch.empty(m, n_joints * 3, dtype=torch.float32, device=rot6d2d.device)
    _rot6d_scatter(rot6d2d, slot_of_joint, out)
    return out

"""Per-kernel parity: every C-ABI entry point of libemage_hip.so, called through pantomatrix_amd.ops on the
MI355X, against its CPU restatement (tests/fake_ops.py == the oracle's arithmetic) on the same seeded inputs.
Tolerances are written per test; index outputs are compared exactly."""
import math

import numpy as np
import pytest
import torch

import fake_ops as F
from oracle import emage_oracle as orc
from pantomatrix_amd import ops
from pantomatrix_amd._lib import BF16, F32, F16X3, H2

pytestmark = pytest.mark.gpu
DEV = "cuda"
TD = {F32: torch.float32, BF16: torch.bfloat16, F16X3: torch.float32}


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _cmp(name, got, ref, atol, rtol=0.0):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = int((~(err <= tol)).sum())           # a NaN on either side counts
    assert bad == 0, f"{name}: {bad}/{err.numel()} outside tol, max err {float(err.max()):.3e}, ref max {float(ref.abs().max()):.3e}"


GEMM_CASES = [
    # name, M-structure (nb, lin, lout), cin, n, taps, stride, pad, flags
    ("linear_small", (1, 300, 300), 128, 200, 1, 1, 0, dict(bias=True, slope=0.1)),
    ("linear_768x2304", (4, 64, 64), 768, 2304, 1, 1, 0, dict(bias=True)),
    ("linear_res_f32", (3, 50, 50), 1536, 768, 1, 1, 0, dict(bias=True, res="f32", want="f32")),
    ("linear_n64", (2, 700, 700), 64, 64, 1, 1, 0, dict(bias=True, slope=0.0)),
    ("linear_vt", (3, 40, 40), 768, 2304, 1, 1, 0, dict(bias=True, vt=1536)),
    ("conv3_337", (3, 37, 37), 337, 256, 3, 1, 1, dict(bias=True, slope=0.2, n_store=256)),
    ("conv3_res", (2, 64, 64), 256, 256, 3, 1, 1, dict(bias=True, res="lo", n_store=256)),
    ("conv3_out106", (2, 29, 29), 256, 106, 3, 1, 1, dict(bias=True, slope=0.2, n_store=128)),
    ("conv15_s6", (2, 1241, 205), 64, 256, 15, 6, 0, dict(bias=True, slope_vec=True)),
    ("conv15_s1_resfirst", (2, 300, 300), 64, 64, 15, 1, 7, dict(bias=True, slope=0.01, res="lo", res_first=True)),
    ("conv15_s3", (3, 205, 64), 128, 512, 15, 3, 0, dict(bias=True)),
    ("big_m", (64, 64, 64), 768, 768, 1, 1, 0, dict(bias=True, res="f32", want="both")),
]


@pytest.mark.parametrize("dtype", [F32, BF16, F16X3], ids=["fp32", "bf16", "f16x3"])
@pytest.mark.parametrize("case", GEMM_CASES, ids=[c[0] for c in GEMM_CASES])
def test_gemm(case, dtype):
    name, (nb, lin, lout), cin, n, taps, stride, pad, fl = case
    g = _g(hash(name) % 1000)
    td = TD[dtype]
    cp = ops.round_up(cin, 64)
    m = nb * lout
    a = torch.zeros(nb * lin, cp)
    a[:, :cin] = torch.randn(nb * lin, cin, generator=g)
    w = torch.zeros(n, taps, cp)
    w[:, :, :cin] = torch.randn(n, taps, cin, generator=g) / math.sqrt(cin * taps)
    a, w = a.to(td), w.reshape(n, taps * cp).to(td)
    w_scale = 1.0
    if dtype == F16X3:
        w, w_scale = ops.split_f16_weights(w)
    bias = torch.randn(n, generator=g) * 0.1 if fl.get("bias") else None
    slope = None
    if "slope" in fl:
        slope = torch.full((n,), float(fl["slope"]))
    if fl.get("slope_vec"):
        slope = torch.cat([torch.full((n // 2,), 0.01), torch.ones(n - n // 2)])
    res = None
    if fl.get("res") == "f32":
        res = torch.randn(m, n, generator=g)
    elif fl.get("res") == "lo":
        res = torch.randn(m, n + 64, generator=g).to(td)[:, 32:32 + n]      # strided, offset view
    n_store = fl.get("n_store", 0)
    want = fl.get("want", "lo")
    vt0 = fl.get("vt")
    ncol = vt0 if vt0 else n

    def run(mod, dev):
        mv = lambda t: None if t is None else t.to(dev)
        res_d = res.to(dev) if fl.get("res") == "f32" else None
        if fl.get("res") == "lo":
            base = torch.zeros(m, n + 64, dtype=td)
            base[:, 32:32 + n] = res
            res_d = base.to(dev)[:, 32:32 + n]
        out = torch.full((m, max(ncol, n_store)), 7.0, dtype=td, device=dev) if want in ("lo", "both") else None
        out_f = torch.full((m, ncol), 7.0, device=dev) if want in ("f32", "both") else None
        out_t = None
        if vt0:
            tp = ops.round_up(lout, 32)
            out_t = torch.zeros(nb, n - vt0, tp, dtype=td, device=dev)
        mod.gemm(dtype, mv(a), mv(w), mv(bias), mv(slope), res_d, out, out_f, out_t, n=n, cp=cp, n_store=n_store,
                 t_col0=vt0 or 0, t_rows=lout if vt0 else 0, res_first=fl.get("res_first", False),
                 taps=taps, stride=stride, pad=pad, lin=lin, lout=lout, m=m, w_scale=w_scale)
        return out, out_f, out_t

    got = run(ops, DEV)
    torch.cuda.synchronize()
    ref = run(F, "cpu")
    # operands are identical (bf16-rounded where bf16); only the fp32 accumulation order differs
    # (F16X3: the fake restates the split arithmetic and accumulates wide; the kernel accumulates in fp32)
    atol = {F32: 2e-4, BF16: 2e-2, F16X3: 2e-5}[dtype]
    for nm, gt, rf in zip(("out", "out_f32", "out_t"), got, ref):
        if gt is not None:
            _cmp(f"{name}.{nm}", gt, rf, atol=atol if (nm != "out_f32" or dtype == F16X3) else 2e-4,
                 rtol=1e-2 if dtype == BF16 and nm != "out_f32" else (1e-5 if dtype == F16X3 else 1e-4))


@pytest.mark.parametrize("m,k,n", [(4096, 768, 768), (4096, 1536, 768), (512, 256, 2304)])
def test_gemm_f16x3_is_fp32_grade(m, k, n):
    """The split-f16 GEMM against the TRUE product (fp64 of the unsplit operands): its error must be of the order of
    the exact-fp32 MFMA kernel's own rounding error, orders of magnitude below bf16 — this is what lets the fast
    mode keep the reference's VQ code indices (north_star: bit-exact indices, 1e-3 on rotations)."""
    g = _g(m + k + n)
    a = torch.randn(m, k, generator=g) * torch.logspace(-2, 1.5, m).view(m, 1)      # rows from 0.01 to 30 in scale
    w = torch.randn(n, k, generator=g) / math.sqrt(k)
    ref = a.double() @ w.double().t()
    scale = ref.abs().mean(dim=1, keepdim=True)                                     # per-row magnitude of the result
    errs = {}
    for dtype in (F32, F16X3, BF16):
        wd, ws = (ops.split_f16_weights(w) if dtype == F16X3 else (w.to(TD[dtype]), 1.0))
        out = torch.empty(m, n, device=DEV)
        ops.gemm(dtype, a.to(TD[dtype]).to(DEV), wd.to(DEV), None, None, None, None, out, None, n=n, cp=k, w_scale=ws)
        errs[dtype] = float(((out.cpu().double() - ref).abs() / scale).max())
    print(f"max row-relative error vs fp64: fp32 MFMA {errs[F32]:.2e}, split-f16 {errs[F16X3]:.2e}, bf16 {errs[BF16]:.2e}")
    assert errs[F16X3] < 4 * errs[F32] + 2e-6
    assert errs[F16X3] < 1e-3 * errs[BF16]


@pytest.mark.parametrize("dtype", [F32, BF16, F16X3], ids=["fp32", "bf16", "f16x3"])
@pytest.mark.parametrize("b,tq,tk", [(3, 64, 64), (2, 10, 11), (2, 24, 25), (1, 64, 128), (2, 60, 60), (5, 33, 33)])
def test_attention(dtype, b, tq, tk):
    g = _g(b * 1000 + tq * 10 + tk)
    td, h, hd, d = TD[dtype], 4, 192, 768
    q = torch.randn(b * tq, 2 * d, generator=g).to(td)[:, :d]            # ld = 2d views, like the qk buffer
    k = torch.randn(b * tk, 2 * d, generator=g).to(td)[:, d:]
    tp = ops.round_up(tk, 32)
    vt_all = torch.zeros(b, 2 * d, tp, dtype=td)
    vt_all[:, :, :tk] = torch.randn(b, 2 * d, tk, generator=g).to(td)
    out_c = torch.zeros(b * tq, d, dtype=td)
    F.attention(dtype, q, k, vt_all[:, d:], 2 * d, out_c, b, h, tq, tk, hd)
    qb = torch.zeros(b * tq, 2 * d, dtype=td); qb[:, :d] = q
    kb = torch.zeros(b * tk, 2 * d, dtype=td); kb[:, d:] = k
    qd, kd, vd = qb.to(DEV)[:, :d], kb.to(DEV)[:, d:], vt_all.to(DEV)
    out_d = torch.zeros(b * tq, d, dtype=td, device=DEV)
    ops.attention(dtype, qd, kd, vd[:, d:], 2 * d, out_d, b, h, tq, tk, hd)
    # split-f16 MFMA on fp32 tensors: the same fp32-grade tolerance as the exact-fp32 MFMA path
    _cmp("attention", out_d, out_c, atol=2e-2 if dtype == BF16 else 2e-5)


@pytest.mark.parametrize("dtype", [F32, BF16], ids=["fp32", "bf16"])
def test_layernorm_add_pack_cast_gather(dtype):
    g = _g(3)
    td = TD[dtype]
    x = (torch.randn(1000, 768, generator=g) * 3 + 0.5).to(td)            # residual stream is stored in `dtype`
    gamma, beta = 1 + 0.1 * torch.randn(768, generator=g), 0.1 * torch.randn(768, generator=g)
    addt = torch.randn(1000, 768, generator=g).to(td)
    for add in (None, addt):
        yf_c, y_c = torch.zeros(1000, 768), torch.zeros(1000, 768, dtype=td)
        F.layernorm(dtype, x, gamma, beta, 1e-5, add, yf_c, y_c)
        yf, y = torch.zeros(1000, 768, device=DEV), torch.zeros(1000, 768, dtype=td, device=DEV)
        ops.layernorm(dtype, x.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5, None if add is None else add.to(DEV), yf, y)
        _cmp("layernorm.f32", yf, yf_c, atol=2e-5)
        _cmp("layernorm.lo", y, y_c, atol=1e-6 if dtype == F32 else 4e-2)
    # add: fp32 a, fp32 broadcast table b, compute-dtype c
    a, b2, c = torch.randn(640, 768, generator=g), torch.randn(64, 768, generator=g), torch.randn(640, 768, generator=g).to(td)
    of_c, o_c = torch.zeros(640, 768), torch.zeros(640, 768, dtype=td)
    F.add(dtype, a, b2, c, of_c, o_c, mod_b=64)
    of, o = torch.zeros(640, 768, device=DEV), torch.zeros(640, 768, dtype=td, device=DEV)
    ops.add(dtype, a.to(DEV), b2.to(DEV), c.to(DEV), of, o, mod_b=64)
    assert torch.equal(of.cpu(), of_c) and torch.equal(o.cpu(), o_c)
    o2_c, o2 = torch.zeros(640, 768, dtype=td), torch.zeros(640, 768, dtype=td, device=DEV)
    F.add(dtype, c, c, None, None, o2_c)
    ops.add(dtype, c.to(DEV), c.to(DEV), None, None, o2)
    assert torch.equal(o2.cpu(), o2_c)
    # a 26-frame window cut out of 5 longer clips (read in place), with and without the 4-frame seed splice (M:386-391)
    clip_m, clip_k = torch.randn(5, 70, 337, generator=g), (torch.rand(5, 70, 337, generator=g) > 0.5).float()
    emb = torch.randn(337, generator=g)
    prev = torch.randn(5, 17, 337, generator=g)               # a previous decode: its last 4 frames are the seed (a strided view)
    dm, dk, dprev = clip_m.to(DEV), clip_k.to(DEV), prev.to(DEV)
    for seed in (False, True):
        ref = F.pack_motion(dtype, clip_m[:, 30:56], clip_k[:, 30:56], emb, 384, seed=prev[:, 13:] if seed else None)
        got = ops.pack_motion(dtype, dm[:, 30:56], dk[:, 30:56], emb.to(DEV), 384, seed=dprev[:, 13:] if seed else None)
        assert torch.equal(got.cpu(), ref), seed
    one = ops.pack_motion(dtype, dm[:1, :1], dk[:1, :1], emb.to(DEV), 384)     # a single frame of a single clip
    assert torch.equal(one.cpu(), F.pack_motion(dtype, clip_m[:1, :1], clip_k[:1, :1], emb, 384))
    src = torch.randn(77, 106, generator=g)
    assert torch.equal(ops.cast_pad(dtype, src.to(DEV), 128).cpu(), F.cast_pad(dtype, src, 128))
    table, idx = torch.randn(256, 256, generator=g), torch.randint(0, 256, (513,), generator=g)
    assert torch.equal(ops.gather_rows(table.to(DEV), idx.to(DEV), dtype, 256).cpu(), F.gather_rows(table, idx, dtype, 256))
    # index views: a (B, T) window of a longer (B, L) code buffer, and one id per clip broadcast over T frames
    codes = torch.randint(0, 256, (6, 128), generator=g)
    win = codes.to(DEV)[:, 47:64]
    assert torch.equal(ops.gather_rows(table.to(DEV), win, dtype, 320).cpu(), F.gather_rows(table, codes[:, 47:64], dtype, 320))
    spk = torch.randint(0, 256, (6, 1), generator=g)
    assert torch.equal(ops.gather_rows(table.to(DEV), spk.to(DEV).expand(6, 64), dtype).cpu(), F.gather_rows(table, spk.expand(6, 64), dtype))


def test_vq_argmin_large_n_kernel_equals_small_n_kernel():
    """N >= 16 384 takes the 64-rows-per-block kernel with the code tiles staged in LDS: the same per-(row, code)
    arithmetic, hence the same indices as the 16-row kernel run on slices of the same input (and as torch on the CPU)."""
    g = _g(77)
    n = 40000 + 7                                   # not a multiple of 64
    z, cb = torch.randn(n, 256, generator=g), torch.randn(250, 256, generator=g)      # K not a multiple of 16
    zd, cbd = z.to(DEV), cb.to(DEV)
    big = ops.vq_argmin(zd, cbd)
    small = torch.cat([ops.vq_argmin(zd[i:i + 8192], cbd) for i in range(0, n, 8192)])
    assert torch.equal(big, small)
    ref = orc.vq_nearest(z[:4096].unsqueeze(0), cb).reshape(-1)
    assert torch.equal(big[:4096].cpu(), ref)


@pytest.mark.parametrize("n,k,d", [(4096, 256, 256), (7680, 256, 256), (33, 256, 256), (500, 100, 64), (200, 37, 240)])
def test_vq_argmin_indices_exact(n, k, d):
    g = _g(n + k)
    z, cb = torch.randn(n, d, generator=g), torch.randn(k, d, generator=g)
    ref = orc.vq_nearest(z.unsqueeze(0), cb).reshape(-1)
    got = ops.vq_argmin(z.to(DEV), cb.to(DEV)).cpu()
    assert got.dtype == torch.int64
    if not torch.equal(got, ref):
        dist = orc.vq_distances(z, cb)
        bad = (got != ref).nonzero().reshape(-1)
        gap = (dist[bad, got[bad]] - dist[bad, ref[bad]]).abs()
        raise AssertionError(f"{bad.numel()} / {n} indices differ; distance gaps {gap[:8].tolist()}")
    # invariant: every codebook row maps to itself; exact ties pick the first index
    assert torch.equal(ops.vq_argmin(cb.to(DEV), cb.to(DEV)).cpu(), torch.arange(k))
    cb2 = torch.cat([cb, cb], 0)[: min(2 * k, 4096)]
    assert torch.equal(ops.vq_argmin(z[:64].to(DEV), cb2.to(DEV)).cpu(), ref[:64])
    if n % 11 == 0:      # written in place into a (B, T) window of a longer code buffer; the rest of the buffer untouched
        buf = torch.full((11, n // 11 + 9), -7, dtype=torch.int64, device=DEV)
        ops.vq_argmin(z.to(DEV), cb.to(DEV), out=buf[:, 5:5 + n // 11])
        assert torch.equal(buf[:, 5:5 + n // 11].cpu(), ref.view(11, -1))
        assert int((buf.cpu() == -7).sum()) == 11 * 9


def test_nan_surfaces_like_torch():
    """torch.argmin / torch.max return the position of the first NaN: a NaN upstream must not turn into a valid-looking code."""
    g = _g(12)
    z, cb = torch.randn(40, 256, generator=g), torch.randn(256, 256, generator=g)
    z[3, 100] = float("nan")                     # every distance of row 3 is NaN -> index 0
    cb2 = cb.clone()
    cb2[77, 5] = float("nan")                    # code 77 is NaN for every row -> index 77
    for zz, cc in ((z, cb), (z[:3], cb2)):
        ref = torch.argmin(orc.vq_distances(zz, cc), dim=1)
        assert torch.equal(ops.vq_argmin(zz.to(DEV), cc.to(DEV)).cpu(), ref)
    x = torch.randn(8, 256, generator=g)
    x[2, 40] = float("nan")
    ref = torch.max(torch.log_softmax(x, dim=1), dim=1)[1]
    assert torch.equal(ops.argmax_logsoftmax(x.to(DEV)).cpu(), ref)


def test_argmax_logsoftmax_exact():
    g = _g(9)
    x = torch.randn(4096, 256, generator=g) * 3
    x[5, 17] = x[5, 200] = 50.0                   # exact tie -> first index
    ref = torch.max(torch.log_softmax(x, dim=1), dim=1)[1]
    got = ops.argmax_logsoftmax(x.to(DEV)).cpu()
    assert torch.equal(got, ref) and int(got[5]) == 17
    buf = torch.full((64, 128), -7, dtype=torch.int64, device=DEV)            # 64 clips x 64 frames into columns 60..123 of (64, 128)
    ops.argmax_logsoftmax(x.to(DEV), out=buf[:, 60:124])
    assert torch.equal(buf[:, 60:124].cpu(), ref.view(64, 64)) and int((buf.cpu() == -7).sum()) == 64 * 64


def test_wav_conv_in():
    g = _g(4)
    wav = 0.1 * torch.randn(3, 34112, generator=g)
    w, bias = torch.randn(256, 15, generator=g) / 4, torch.randn(256, generator=g) * 0.1
    slope = torch.cat([torch.full((64,), 0.01), torch.ones(64)] * 2)
    for dtype in (F32, BF16):
        out_c = torch.zeros(3 * 7460, 256, dtype=TD[dtype])
        F.wav_conv_in(dtype, wav, w, bias, slope, out_c, 7460, 5, 1600)
        out = torch.zeros(3 * 7460, 256, dtype=TD[dtype], device=DEV)
        ops.wav_conv_in(dtype, wav.to(DEV), w.to(DEV), bias.to(DEV), slope.to(DEV), out, 7460, 5, 1600)
        _cmp("wav_conv_in", out, out_c, atol=1e-5 if dtype == F32 else 1e-2)
    # two sliding windows per clip read in place from longer clips (inference(): hop = 60 frames, M:393-394)
    clips = 0.1 * torch.randn(3, 34112 + 31980 + 11, generator=g)
    for dtype in (F32, BF16):
        out_c = torch.zeros(2 * 3 * 7460, 256, dtype=TD[dtype])
        F.wav_conv_in(dtype, clips, w, bias, slope, out_c, 7460, 5, 1600, nwin=2, hop=31980, win_len=34112)
        out = torch.zeros(2 * 3 * 7460, 256, dtype=TD[dtype], device=DEV)
        ops.wav_conv_in(dtype, clips.to(DEV), w.to(DEV), bias.to(DEV), slope.to(DEV), out, 7460, 5, 1600, nwin=2, hop=31980, win_len=34112)
        _cmp("wav_conv_in.windows", out, out_c, atol=1e-5 if dtype == F32 else 1e-2)


@pytest.mark.parametrize("c,lw", [(32, 136000), (64, 34112), (128, 21003), (8, 9000)])
def test_wav_conv_in_narrow_encoders(c, lw):
    """The first layer at the widths of DisCo / CaMN (32 .. 128 channels; round 6: rows per block grow as the channels shrink — 512 at C = 32 —
    so that a thread's register-resident filters are loaded once per 16 rows, not per 2): ragged lengths, the last block partly empty, against
    the CPU restatement; per output the same fmaf chain as before."""
    g = _g(40 + c)
    wav = 0.1 * torch.randn(2, lw, generator=g)
    w, bias = torch.randn(c, 15, generator=g) / 4, torch.randn(c, generator=g) * 0.1
    slope = torch.full((c,), 0.01)
    lout = (lw + 2 * 1600 - 15) // 5 + 1
    out_c = torch.zeros(2 * lout, c)
    F.wav_conv_in(F32, wav, w, bias, slope, out_c, lout, 5, 1600)
    out = torch.full((2 * lout, c), 7.0, device=DEV)
    ops.wav_conv_in(F32, wav.to(DEV), w.to(DEV), bias.to(DEV), slope.to(DEV), out, lout, 5, 1600)
    _cmp(f"wav_conv_in.c{c}", out, out_c, atol=1e-5)


def test_rotations_merge_scan(golden_dir):
    import os
    g = _g(5)
    gold = np.load(os.path.join(golden_dir, "rotations.npz"))
    d6 = torch.randn(4, 50, 6, generator=g)
    aa = torch.randn(4, 50, 3, generator=g) * torch.tensor([1.0, 0.3, 1e-4, 0.0]).view(4, 1, 1)
    np.testing.assert_allclose(ops.rot6d_to_axis_angle(d6.to(DEV)).cpu().numpy(), gold["rot6d_to_aa"], atol=2e-4, rtol=0)   # near-pi rotations amplify 1-ulp sin/atan2 differences
    np.testing.assert_allclose(ops.axis_angle_to_rot6d(aa.to(DEV)).cpu().numpy(), gold["aa_to_rot6d"], atol=1e-5, rtol=0)
    assert torch.equal(ops.axis_angle_to_rot6d(torch.zeros(2, 3, device=DEV)).cpu(), torch.tensor([[1.0, 0, 0, 0, 1, 0]] * 2))
    m = 300
    face, upper, hands, lower = (torch.randn(m, c, generator=g) for c in (106, 78, 180, 61))
    for parts in ((face, upper, hands, lower), (None, upper, None, lower), (face, None, hands, None)):
        ref = F.merge_parts(*parts, m, "cpu")
        got = ops.merge_parts(*[None if p is None else p.to(DEV) for p in parts], m, DEV)
        for nm, gt, rf in zip(("aa", "motion", "expr"), got, ref):
            _cmp("merge." + nm, gt, rf, atol=1e-3)   # w = sqrt(1+trace)/2 near angle pi turns 1-ulp differences into ~3e-4
    vel = torch.randn(5 * 120, 61, generator=g)
    init = torch.randn(5, 3, generator=g)
    ref = F.velocity_to_position(vel, 54, init, 1 / 30, 5, 120)
    got = ops.velocity_to_position(vel.to(DEV), 54, init.to(DEV), 1 / 30, 5, 120)
    assert torch.equal(got.cpu(), ref), float((got.cpu() - ref).abs().max())
    one = torch.randn(1, 3, generator=g)                                      # one start position shared by every clip (M:198-200)
    assert torch.equal(ops.velocity_to_position(vel.to(DEV), 54, one.to(DEV), 1 / 30, 5, 120).cpu(),
                       F.velocity_to_position(vel, 54, one, 1 / 30, 5, 120))
    strided = torch.randn(5, 7, 3, generator=g)                               # ref_trans (B, T', 3): frame 0 of every clip, as a view
    assert torch.equal(ops.velocity_to_position(vel.to(DEV), 54, strided.to(DEV)[:, 0, :], 1 / 30, 5, 120).cpu(),
                       F.velocity_to_position(vel, 54, strided[:, 0, :], 1 / 30, 5, 120))
    for b, t in ((3, 1), (2, 841), (1, 6000)):       # single frame, a 28 s clip, and one beyond the LDS-staged scan's reach
        vel = torch.randn(b * t, 61, generator=g)
        init = torch.randn(b, 3, generator=g)
        ref = F.velocity_to_position(vel, 54, init, 1 / 30, b, t)
        got = ops.velocity_to_position(vel.to(DEV), 54, init.to(DEV), 1 / 30, b, t)
        assert torch.equal(got.cpu(), ref), (b, t, float((got.cpu() - ref).abs().max()))


def test_bad_arguments_raise():
    from pantomatrix_amd._lib import EmageKernelError
    with pytest.raises(EmageKernelError):
        ops.vq_argmin(torch.zeros(4, 6, device=DEV), torch.zeros(3, 6, device=DEV))       # D % 4 != 0
    with pytest.raises(RuntimeError):
        ops.vq_argmin(torch.zeros(4, 8), torch.zeros(3, 8))                                 # CPU tensors: no fallback


@pytest.mark.parametrize("cfg", [0, 3, 18, 25, 27, 32, 33, 34, 36, 37, 45, 46, 47, 48])
@pytest.mark.parametrize("dtype", [F32, BF16, F16X3], ids=["fp32", "bf16", "f16x3"])
def test_gemm_every_tile_configuration(cfg, dtype):
    """Each compiled tile configuration (register-staged 0 / 3, LDS-DMA ring kernels, two-K-tiles-per-slot 45-48) on the awkward cases."""
    if dtype == F16X3 and cfg in (0, 3):
        pytest.skip("the register-staged kernels have no split-f16 form")
    if dtype == BF16 and cfg >= 45:
        pytest.skip("two K-tiles per ring slot exist for the fp32-storage modes only")
    from pantomatrix_amd import _lib
    lib = _lib.use_tools(True)          # the product library carries only the configurations its heuristic selects
    try:
        assert lib.emage_set_tuning(0, cfg) == 0
        for case in GEMM_CASES:
            if case[0] in ("linear_small", "linear_vt", "conv3_337", "conv3_out106", "conv15_s6", "conv15_s1_resfirst", "linear_res_f32"):
                test_gemm(case, dtype)
    finally:
        lib.emage_set_tuning(0, -1)
        _lib.use_tools(False)


@pytest.mark.parametrize("dtype", [F32, BF16, F16X3], ids=["fp32", "bf16", "f16x3"])
@pytest.mark.parametrize("c,nseq,l,res", [(64, 3, 1241, True), (64, 2, 200, False), (128, 5, 205, True), (128, 1, 64, True), (64, 1, 7460, True)])
def test_conv_slab_equals_gemm_bitwise(dtype, c, nseq, l, res):
    """The LDS-resident-slab convolution computes exactly what emage_gemm computes for the same stride-1 k = 15 conv
    (same K order, same MFMA grouping, same epilogue association): bit-identical outputs in every precision, for
    sequences that are not a multiple of the 128-position tile and with / without the shortcut operand."""
    g = _g(c + nseq + l)
    td = TD[dtype]
    a = torch.randn(nseq * l, c, generator=g).to(td)
    w = (torch.randn(c, 15, c, generator=g) / math.sqrt(15 * c)).reshape(c, 15 * c)
    wp, ws = (ops.split_f16_weights(w) if dtype == F16X3 else (w.to(td), 1.0))
    bias, slope = torch.randn(c, generator=g) * 0.1, torch.full((c,), 0.01)
    sc = torch.randn(nseq * l, 2 * c, generator=g).to(td)[:, c:] if res else None       # a strided view, like the stacked conv1 output
    ad, wd, bd, sd = a.to(DEV), wp.to(DEV), bias.to(DEV), slope.to(DEV)
    scd = None
    if res:
        base = torch.zeros(nseq * l, 2 * c, dtype=td)
        base[:, c:] = sc
        scd = base.to(DEV)[:, c:]
    ref = torch.zeros(nseq * l, c, dtype=td, device=DEV)
    ops.gemm(dtype, ad, wd, bd, sd, scd, ref, None, None, n=c, cp=c, res_first=True, taps=15, stride=1, pad=7, lin=l, lout=l, m=nseq * l, w_scale=ws)
    got = torch.full((nseq * l, c), 7.0, dtype=td, device=DEV)
    ops.conv_slab(dtype, ad, wd, bd, sd, scd, got, nseq=nseq, l=l, taps=15, pad=7, w_scale=ws)
    torch.cuda.synchronize()
    assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())
    cpu = torch.zeros(nseq * l, c, dtype=td)
    F.conv_slab(dtype, a, wp, bias, slope, sc, cpu, nseq=nseq, l=l, taps=15, pad=7, w_scale=ws)
    _cmp("conv_slab vs cpu", got, cpu, atol={F32: 2e-4, BF16: 3e-2, F16X3: 2e-5}[dtype], rtol=1e-2 if dtype == BF16 else 1e-5)


@pytest.mark.parametrize("dtype", [F32, BF16, F16X3], ids=["fp32", "bf16", "f16x3"])
@pytest.mark.parametrize("nclip,nwin", [(3, 1), (2, 2)])
def test_wav_block0_equals_unfused_bitwise(dtype, nclip, nwin):
    """WavEncoder block 0 in one launch (conv1 from the waveform into LDS, conv2, shortcut in the epilogue) against the
    unfused emage_wav_conv_in + emage_gemm sequence: bit-identical, for plain clips and for sliding windows read in place."""
    g = _g(nclip * 10 + nwin)
    td = TD[dtype]
    c, win, hop = 64, 34112, 31980
    wav = 0.1 * torch.randn(nclip, win + (nwin - 1) * hop + 5, generator=g)
    w_first = torch.randn(2 * c, 15, generator=g) / 4                   # [conv1 | shortcut], eval BatchNorm already folded
    b_first = torch.randn(2 * c, generator=g) * 0.1
    s_first = torch.cat([torch.full((c,), 0.01), torch.ones(c)])
    w2 = (torch.randn(c, 15, c, generator=g) / math.sqrt(15 * c)).reshape(c, 15 * c)
    w2p, ws = (ops.split_f16_weights(w2) if dtype == F16X3 else (w2.to(td), 1.0))
    b2, s2 = torch.randn(c, generator=g) * 0.1, torch.full((c,), 0.01)
    lout, nseq = 7460, nclip * nwin
    wd = wav.to(DEV)
    y0 = torch.zeros(nseq * lout, 2 * c, dtype=td, device=DEV)
    storage = F32 if dtype == F16X3 else dtype                          # the first layer runs in the mode's storage type
    ops.wav_conv_in(storage, wd, w_first.to(DEV), b_first.to(DEV), s_first.to(DEV), y0, lout, 5, 1600, nwin=nwin, hop=hop, win_len=win)
    ref = torch.zeros(nseq * lout, c, dtype=td, device=DEV)
    ops.gemm(dtype, y0[:, :c], w2p.to(DEV), b2.to(DEV), s2.to(DEV), y0[:, c:], ref, None, None, n=c, cp=c, res_first=True, taps=15, stride=1, pad=7,
             lin=lout, lout=lout, m=nseq * lout, w_scale=ws)
    got = torch.full((nseq * lout, c), 7.0, dtype=td, device=DEV)
    wf = w_first.to(DEV)
    ops.wav_block0(dtype, wd, wf[:c], b_first.to(DEV)[:c], 0.01, wf[c:], b_first.to(DEV)[c:], 5, 1600, w2p.to(DEV), b2.to(DEV), s2.to(DEV), 15, 7,
                   got, lout, nwin=nwin, hop=hop, win_len=win, w_scale=ws)
    torch.cuda.synchronize()
    assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())


# ---- EMAGE_H2: pre-split activation storage (csrc/h2.h) -------------------------------------------------------------------
from pantomatrix_amd._lib import H2  # noqa: E402

H2_GEMM_CASES = [c for c in GEMM_CASES if c[0] not in ("conv15_s6", "conv15_s1_resfirst", "conv15_s3")] + [
    ("h2_kv_all", (4, 64, 64), 768, 3072, 1, 1, 0, dict(bias=True, vt=1536)),
    ("h2_ragged_tail", (2, 65, 65), 512, 1536, 1, 1, 0, dict(bias=True, vt=768)),
    # the K / V projection of all eight cross-attention layers: weights past the eight L2s -> tiles in dispatch order (round 5)
    ("h2_kv_all_8_layers_dispatch_order", (3, 70, 70), 768, 12288, 1, 1, 0, dict(bias=True)),
    ("h2_out106_f32", (2, 29, 29), 106, 106, 3, 1, 1, dict(bias=True, want="f32")),
    ("h2_res_h2", (3, 64, 64), 256, 256, 3, 1, 1, dict(bias=True, res="h2", n_store=256)),
    ("h2_small_m", (1, 17, 17), 256, 256, 3, 1, 1, dict(bias=True, slope=0.2)),
    # bare contractions with a long K (the weight gradients of a training step): 144 / 288 tiles -> one slice per block (split-K with fp32
    # atomics below 100 tiles: the narrow heads' gradients, covered by the training-step tests)
    ("h2_splitk_144_tiles_one_slice", (12, 64, 64), 3584, 768, 1, 1, 0, dict(want="f32")),
    ("h2_splitk_288_tiles_one_slice", (12, 64, 64), 3584, 1536, 1, 1, 0, dict(want="f32")),
]


def _run_h2_gemm(case, mod, dev, cfg=None):
    name, (nb, lin, lout), cin, n, taps, stride, pad, fl = case
    g = _g(hash(name) % 1000)
    cp = ops.round_up(cin, 64)
    m = nb * lout
    a = torch.zeros(nb * lin, cp)
    a[:, :cin] = torch.randn(nb * lin, cin, generator=g)
    w = torch.zeros(n, taps, cp)
    w[:, :, :cin] = torch.randn(n, taps, cin, generator=g) / math.sqrt(cin * taps)
    w_h2, ws = ops.split_f16_weights_h2(w.reshape(n, taps * cp))
    bias = torch.randn(n, generator=g) * 0.1 if fl.get("bias") else None
    slope = torch.full((n,), float(fl["slope"])) if "slope" in fl else None
    n8 = ops.round_up(n, 8)
    res, res_h2 = None, False
    if fl.get("res") in ("f32", "lo"):
        res = torch.randn(m, n8, generator=g)[:, :n]
    elif fl.get("res") == "h2":
        res, res_h2 = ops.h2_pack(torch.randn(m, n8, generator=g)), True
    n_store = fl.get("n_store", 0)
    want = fl.get("want", "lo")
    vt0 = fl.get("vt")
    ncol = vt0 if vt0 else n
    mv = lambda t: None if t is None else t.to(dev)
    out = torch.full((m, ops.round_up(max(ncol, n_store), 8)), 7.0, device=dev) if want in ("lo", "both") else None
    out_f = torch.full((m, ncol), 7.0, device=dev) if want in ("f32", "both") else None
    out_t = torch.zeros(nb, n - vt0, ops.round_up(lout, 32), device=dev) if vt0 else None
    mod.gemm(H2, mv(ops.h2_pack(a)), mv(w_h2), mv(bias), mv(slope), mv(res), out, out_f, out_t, n=n, cp=cp, n_store=n_store,
             t_col0=vt0 or 0, t_rows=lout if vt0 else 0, taps=taps, stride=stride, pad=pad, lin=lin, lout=lout, m=m, w_scale=ws, res_h2=res_h2)
    vals = None if out is None else ops.h2_unpack(out)
    return vals, out_f, out_t, max(ncol, n_store)


@pytest.mark.parametrize("case", H2_GEMM_CASES, ids=[c[0] for c in H2_GEMM_CASES])
def test_gemm_h2(case):
    """EMAGE_H2 contraction (pre-split operands, csrc/gemm_h2.hip) against the CPU restatement of the same three-product
    arithmetic: row-major, zero-filled tail, float32 copy, transposed (V^T) destination, float32 / H2 residuals."""
    got = _run_h2_gemm(case, ops, DEV)
    torch.cuda.synchronize()
    ref = _run_h2_gemm(case, F, "cpu")
    for nm, gt, rf in zip(("out", "out_f32", "out_t"), got[:3], ref[:3]):
        if gt is not None:
            width = got[3] if nm == "out" else gt.shape[-1]
            _cmp(f"{case[0]}.{nm}", gt[..., :width], rf[..., :width], atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("cfg", [100, 101, 102, 103, 105, 107, 108, 110, 111, 113, 115, 116, 119, 120, 121, 122, 123, 124, 125, 127, 130, 131, 141, 144, 148, 150, 160, 164, 166, 170, 171,
                                 180, 181, 183, 184, 185, 186, 188, 189])
def test_gemm_h2_every_tile_configuration(cfg):
    """Every EMAGE_H2 tile configuration (wave grids, loader waves, register-pipelined / interleaved K-loops, odd fragment
    counts) on a ragged shape with a transposed tail and on a convolution with a zero-filled channel tail."""
    from pantomatrix_amd import _lib
    lib = _lib.use_tools(True)
    cases = [("h2cfg_vt", (3, 70, 70), 768, 2304, 1, 1, 0, dict(bias=True, vt=1536)),
             ("h2cfg_conv", (3, 37, 37), 337, 106, 3, 1, 1, dict(bias=True, slope=0.2, n_store=128, res=None, want="both"))]
    try:
        for case in cases:
            lib.emage_set_tuning(4, cfg)
            try:
                got = _run_h2_gemm(case, ops, DEV)
            except Exception as e:  # noqa: BLE001  (a tile that does not divide t_col0 is refused, not wrong)
                assert "EINVAL" in str(e) and case[0] == "h2cfg_vt", (cfg, e)
                continue
            torch.cuda.synchronize()
            lib.emage_set_tuning(4, -1)
            ref = _run_h2_gemm(case, F, "cpu")
            for nm, gt, rf in zip(("out", "out_f32", "out_t"), got[:3], ref[:3]):
                if gt is not None:
                    width = got[3] if nm == "out" else gt.shape[-1]
                    _cmp(f"cfg{cfg}.{case[0]}.{nm}", gt[..., :width], rf[..., :width], atol=2e-5, rtol=1e-5)
    finally:
        lib.emage_set_tuning(4, -1)
        _lib.use_tools(False)


def test_gemm_h2_dispatch_order_tiles_change_no_bit():
    """Round 5: a biased projection whose weights exceed the eight L2s (the K / V projection of all cross-attention layers, 12 288 x 768) walks
    its tiles in dispatch order instead of XCD-aware runs (gemm_h2.hip `tile_order`).  Tile order only: the same bits as the runs
    (tools library, emage_set_tuning key 5 bit 4194304 = runs everywhere), and a narrower projection is untouched by the bit."""
    from pantomatrix_amd import _lib
    lib = _lib.use_tools(True)
    try:
        for case in (("kv_all", (5, 50, 50), 768, 12288, 1, 1, 0, dict(bias=True)), ("narrow", (5, 50, 50), 768, 768, 1, 1, 0, dict(bias=True))):
            outs = []
            for variant in (0, 4194304):
                lib.emage_set_tuning(5, variant)
                outs.append(_run_h2_gemm(case, ops, DEV))
            torch.cuda.synchronize()
            for nm, a, b in zip(("out", "out_f32", "out_t"), outs[0][:3], outs[1][:3]):
                if a is not None:
                    assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (case[0], nm)
    finally:
        lib.emage_set_tuning(5, 0)
        _lib.use_tools(False)


def test_gemm_h2_two_ktiles_per_slot_change_no_bit():
    """Round 5 (VERDICT next #1b, "BK = 64"): the KPB = 2 configurations — two 32-k sub-tiles per ring slot and s_barrier — issue the MFMAs
    of every accumulator in the order of their KPB = 1 twins: the same bits, on a residual GEMM, a ragged V^T GEMM and a convolution.
    Likewise the ring-of-8 tile for grids of at most one tile per CU (188: measured slower, tools only) against the three-blocks-per-CU tile (120)."""
    from pantomatrix_amd import _lib
    lib = _lib.use_tools(True)
    cases = [("kpb_res", (64, 64, 64), 768, 768, 1, 1, 0, dict(bias=True, res="h2")),
             ("kpb_vt", (3, 70, 70), 768, 2304, 1, 1, 0, dict(bias=True, vt=1536)),
             ("kpb_conv", (3, 37, 37), 337, 106, 3, 1, 1, dict(bias=True, slope=0.2, n_store=128, res=None, want="both"))]
    try:
        for one, two in ((100, 180), (170, 181), (120, 183), (113, 184), (120, 186), (120, 188), (120, 189)):
            for case in cases:
                outs = []
                for cfg in (one, two):
                    lib.emage_set_tuning(4, cfg)
                    try:
                        outs.append(_run_h2_gemm(case, ops, DEV))
                    except Exception as e:  # noqa: BLE001
                        assert "EINVAL" in str(e) and case[0] == "kpb_vt", (cfg, e)
                        outs.append(None)
                torch.cuda.synchronize()
                if outs[0] is None or outs[1] is None:
                    assert outs[0] is None and outs[1] is None, (one, two, case[0])
                    continue
                for nm, a, b in zip(("out", "out_f32", "out_t"), outs[0][:3], outs[1][:3]):
                    if a is not None:
                        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (one, two, case[0], nm)
    finally:
        lib.emage_set_tuning(4, -1)
        _lib.use_tools(False)


def test_h2_elementwise_producers():
    """Every producer of EMAGE_H2 images against the CPU restatement: LayerNorm (float32 twin + H2 copy), add (float32 and H2
    operands), pack_motion, cast_pad (also in place), gather_rows, attention."""
    g = _g(11)
    x = torch.randn(1000, 768, generator=g) * 3 + 0.5
    gamma, beta = 1 + 0.1 * torch.randn(768, generator=g), 0.1 * torch.randn(768, generator=g)
    addt = torch.randn(1000, 768, generator=g)
    for add in (None, addt):
        yf_c, y_c = torch.zeros(1000, 768), torch.zeros(1000, 768)
        F.layernorm(H2, x, gamma, beta, 1e-5, add, yf_c, y_c)
        yf, y = torch.zeros(1000, 768, device=DEV), torch.zeros(1000, 768, device=DEV)
        ops.layernorm(H2, x.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5, None if add is None else add.to(DEV), yf, y)
        _cmp("layernorm.f32", yf, yf_c, atol=2e-5)
        # the H2 copy is the split of the kernel's own float32 result, bit for bit
        assert torch.equal(y.cpu().view(torch.int32), ops.h2_pack(yf.cpu()).view(torch.int32))
        # and identical to what the float32 LayerNorm kernel writes
        y32 = torch.zeros(1000, 768, device=DEV)
        ops.layernorm(F32, x.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5, None if add is None else add.to(DEV), None, y32)
        assert torch.equal(y32, yf)
    a, b2, c = torch.randn(640, 768, generator=g), torch.randn(64, 768, generator=g), torch.randn(640, 768, generator=g)
    c_h2 = ops.h2_pack(c)
    of_c, o_c = torch.zeros(640, 768), torch.zeros(640, 768)
    F.add(H2, a, b2, c_h2, of_c, o_c, mod_b=64, h2_operands=(2,))
    of, o = torch.zeros(640, 768, device=DEV), torch.zeros(640, 768, device=DEV)
    ops.add(H2, a.to(DEV), b2.to(DEV), c_h2.to(DEV), of, o, mod_b=64, h2_operands=(2,))
    assert torch.equal(of.cpu(), of_c) and torch.equal(o.cpu().view(torch.int32), o_c.view(torch.int32))
    clip_m, clip_k = torch.randn(5, 70, 337, generator=g), (torch.rand(5, 70, 337, generator=g) > 0.5).float()
    emb, prev = torch.randn(337, generator=g), torch.randn(5, 17, 337, generator=g)
    for seed in (False, True):
        ref = F.pack_motion(H2, clip_m[:, 30:56], clip_k[:, 30:56], emb, 384, seed=prev[:, 13:] if seed else None)
        got = ops.pack_motion(H2, clip_m.to(DEV)[:, 30:56], clip_k.to(DEV)[:, 30:56], emb.to(DEV), 384, seed=prev.to(DEV)[:, 13:] if seed else None)
        assert torch.equal(got.cpu().view(torch.int32), ref.view(torch.int32)), seed
    src = torch.randn(77, 106, generator=g)
    assert torch.equal(ops.cast_pad(H2, src.to(DEV), 128).cpu().view(torch.int32), F.cast_pad(H2, src, 128).view(torch.int32))
    wide = torch.randn(90, 512, generator=g)                     # in place on a column block of a wider buffer
    buf = wide.to(DEV)
    ops.cast_pad(H2, buf[:, :256], 256, out=buf[:, :256])
    assert torch.equal(buf[:, :256].cpu().contiguous().view(torch.int32), ops.h2_pack(wide[:, :256]).view(torch.int32)) and torch.equal(buf[:, 256:].cpu(), wide[:, 256:])
    table, idx = torch.randn(256, 256, generator=g), torch.randint(0, 256, (513,), generator=g)
    assert torch.equal(ops.gather_rows(table.to(DEV), idx.to(DEV), H2, 320).cpu().view(torch.int32), F.gather_rows(table, idx, H2, 320).view(torch.int32))


@pytest.mark.parametrize("b,tq,tk", [(3, 64, 64), (2, 10, 11), (1, 64, 128), (5, 33, 33)])
def test_attention_h2_output_equals_f16x3(b, tq, tk):
    """EMAGE_H2 attention = the split-f16 attention with its output stored as an H2 image: same values bit for bit."""
    g = _g(b * 1000 + tq * 10 + tk)
    h, hd, d = 4, 192, 768
    q, k = torch.randn(b * tq, d, generator=g).to(DEV), torch.randn(b * tk, d, generator=g).to(DEV)
    tp = ops.round_up(tk, 32)
    vt = torch.zeros(b, d, tp)
    vt[:, :, :tk] = torch.randn(b, d, tk, generator=g)
    vt = vt.to(DEV)
    o32, oh = torch.zeros(b * tq, d, device=DEV), torch.zeros(b * tq, d, device=DEV)
    ops.attention(F16X3, q, k, vt, d, o32, b, h, tq, tk, hd)
    ops.attention(H2, q, k, vt, d, oh, b, h, tq, tk, hd)
    assert torch.equal(oh.cpu().view(torch.int32), ops.h2_pack(o32.cpu()).view(torch.int32))


@pytest.mark.parametrize("b,tq,tk", [(3, 64, 64), (2, 10, 11), (2, 24, 25), (2, 60, 60), (5, 33, 33), (2, 100, 64), (1, 37, 48)])
def test_attention_lds_staged_kv_is_bit_identical_to_the_register_path(b, tq, tk):
    """Split-f16 attention with K / V^T staged once per workgroup in LDS (the product path for Tk <= 64) against the round-2 form in
    which every query-tile wave fetches and splits K / V^T itself (tools library, emage_set_tuning key 6): the same operands in the
    same MFMA order, so the same bits — float32 output, EMAGE_H2 output and the dropout variant of the training forward; ragged
    query tiles (surplus waves stage and leave), ragged key counts, several query groups per (batch, head)."""
    from pantomatrix_amd import _lib
    g = _g(b * 1000 + tq * 10 + tk)
    h, hd, d = 4, 192, 768
    q = torch.randn(b * tq, 2 * d, generator=g).to(DEV)[:, :d]
    k = torch.randn(b * tk, 2 * d, generator=g).to(DEV)[:, d:]
    tp = ops.round_up(tk, 32)
    vt = torch.zeros(b, 2 * d, tp)
    vt[:, :, :tk] = torch.randn(b, 2 * d, tk, generator=g)
    vt = vt.to(DEV)
    pmask = ((torch.rand(b, h, tq, tk, generator=g) >= 0.1).float() / 0.9).to(DEV)
    lib = _lib.use_tools(True)
    try:
        outs = []
        for variant in (0, 1):
            lib.emage_set_tuning(6, variant)
            o32, oh, od = (torch.zeros(b * tq, d, device=DEV) for _ in range(3))
            ops.attention(F16X3, q, k, vt[:, d:], 2 * d, o32, b, h, tq, tk, hd)
            ops.attention(H2, q, k, vt[:, d:], 2 * d, oh, b, h, tq, tk, hd)
            ops.attention_dropout(F16X3, q, k, vt[:, d:], 2 * d, od, b, h, tq, tk, hd, pmask)
            torch.cuda.synchronize()
            outs.append((o32, oh, od))
    finally:
        lib.emage_set_tuning(6, 0)
        _lib.use_tools(False)
    for new, old, name in zip(outs[0], outs[1], ("float32 out", "h2 out", "dropout")):
        assert torch.equal(new.view(torch.int32), old.view(torch.int32)), name
    assert bool(torch.isfinite(outs[0][0]).all()) and float(outs[0][0].abs().max()) > 0


# ---- emage_gemm_grouped: independent problems of one tile configuration in one launch (csrc/gemm_h2.hip: gemm_h2_group_kernel) -----------
# (split-K contractions add their slices with fp32 atomics: launched one by one, not bit-reproducible — not part of the bit-identity list)
GROUP_CASES = [c for c in H2_GEMM_CASES if not c[0].startswith("h2_splitk")] + [
    ("g_out_proj", (64, 64, 64), 768, 768, 1, 1, 0, dict(bias=True, res="f32", want="both")),
    ("g_out_proj_h2res", (64, 64, 64), 768, 768, 1, 1, 0, dict(bias=True, res="h2", want="f32")),
    ("g_ffn1", (64, 64, 64), 768, 1536, 1, 1, 0, dict(bias=True, slope=0.0)),
    ("g_ffn1_b", (64, 64, 64), 768, 1536, 1, 1, 0, dict(bias=True, slope=0.0)),
    ("g_kv", (64, 64, 64), 768, 1536, 1, 1, 0, dict(bias=True, vt=768, want="f32")),
    ("g_kv_b", (64, 64, 64), 768, 1536, 1, 1, 0, dict(bias=True, vt=768, want="f32")),
    ("g_kv_all", (64, 64, 64), 768, 6144, 1, 1, 0, dict(bias=True, vt=3072, want="f32")),        # many tiles per CU: the 128x192 configuration
    ("g_kv_all_b", (64, 64, 64), 768, 6144, 1, 1, 0, dict(bias=True, vt=3072, want="f32")),
    ("g_dec_a", (64, 120, 120), 256, 256, 3, 1, 1, dict(bias=True, slope=0.2, n_store=256)),
    ("g_dec_b", (64, 120, 120), 256, 256, 3, 1, 1, dict(bias=True, slope=0.2, n_store=256)),
    ("g_dec_c", (64, 12, 12), 256, 256, 3, 1, 1, dict(bias=True, res="h2", n_store=256)),
    ("g_dec_out", (64, 120, 120), 256, 180, 3, 1, 1, dict(bias=True, want="f32")),
]


def _h2_problem(case):
    """Device operands + keyword arguments of one EMAGE_H2 `ops.gemm` call and a maker of fresh (poisoned) output buffers."""
    name, (nb, lin, lout), cin, n, taps, stride, pad, fl = case
    g = _g(hash(name) % 1000)
    cp = ops.round_up(cin, 64)
    m = nb * lout
    a = torch.zeros(nb * lin, cp)
    a[:, :cin] = torch.randn(nb * lin, cin, generator=g)
    w = torch.zeros(n, taps, cp)
    w[:, :, :cin] = torch.randn(n, taps, cin, generator=g) / math.sqrt(cin * taps)
    w_h2, ws = ops.split_f16_weights_h2(w.reshape(n, taps * cp))
    bias = torch.randn(n, generator=g) * 0.1 if fl.get("bias") else None
    slope = torch.full((n,), float(fl["slope"])) if "slope" in fl else None
    n8 = ops.round_up(n, 8)
    res, res_h2 = None, False
    if fl.get("res") in ("f32", "lo"):
        res = torch.randn(m, n8, generator=g)[:, :n]
    elif fl.get("res") == "h2":
        res, res_h2 = ops.h2_pack(torch.randn(m, n8, generator=g)), True
    n_store, want, vt0 = fl.get("n_store", 0), fl.get("want", "lo"), fl.get("vt")
    ncol = vt0 if vt0 else n
    mv = lambda t: None if t is None else t.to(DEV)
    operands = (mv(ops.h2_pack(a)), mv(w_h2), mv(bias), mv(slope), mv(res))
    kw = dict(n=n, cp=cp, n_store=n_store, t_col0=vt0 or 0, t_rows=lout if vt0 else 0, taps=taps, stride=stride, pad=pad, lin=lin, lout=lout, m=m,
              w_scale=ws, res_h2=res_h2)

    def outputs():
        out = torch.full((m, ops.round_up(max(ncol, n_store), 8)), 7.0, device=DEV) if want in ("lo", "both") else None
        out_f = torch.full((m, ncol), 7.0, device=DEV) if want in ("f32", "both") else None
        out_t = torch.full((nb, n - vt0, ops.round_up(lout, 32)), 7.0, device=DEV) if vt0 else None
        return out, out_f, out_t
    return operands, kw, outputs


def test_gemm_grouped_is_bit_identical_to_single_launches():
    """`emage_gemm_grouped` (through `ops.lockstep`: recorded contractions issued as one grouped call) on a mix
    of every EMAGE_H2 test case and model-sized problems — four tile configurations, more problems of one configuration than fit one
    launch (8), convolutions, V^T destinations, H2 / float32 residuals, ragged M and N — writes exactly the bytes the single launches
    write; `ops.gemm_grouped` (the direct form) likewise; and the library reports fewer launches than problems."""
    probs = [_h2_problem(c) for c in GROUP_CASES]
    single = []
    for operands, kw, outputs in probs:
        o = outputs()
        ops.gemm(H2, *operands, *o, **kw)
        single.append(o)
    grouped = [outputs() for _operands, _kw, outputs in probs]
    with ops.lockstep() as ls:
        for (operands, kw, _), o in zip(probs, grouped):
            with ls.chain():
                ops.gemm(H2, *operands, *o, **kw)
    torch.cuda.synchronize()
    assert ls.launches == [("group", len(probs))]
    n_launch = ops.grouped_launch_count(ls.groups[0])
    assert 4 <= n_launch < len(probs) // 2, n_launch
    for case, so, go in zip(GROUP_CASES, single, grouped):
        for nm, a, b in zip(("out", "out_f32", "out_t"), so, go):
            if a is not None:
                assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (case[0], nm)
    direct = [outputs() for _operands, _kw, outputs in probs[-6:]]
    ops.gemm_grouped(H2, [dict(a=op[0], w=op[1], bias=op[2], slope=op[3], res=op[4], out=o[0], out_f32=o[1], out_t=o[2], **kw)
                          for (op, kw, _), o in zip(probs[-6:], direct)])
    for so, go in zip(single[-6:], direct):
        for a, b in zip(so, go):
            if a is not None:
                assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    # lock step with grouping disabled: the same walk, every contraction its own launch
    with ops.lockstep(False) as ls:
        for (operands, kw, outputs) in probs[:3]:
            with ls.chain():
                ops.gemm(H2, *operands, *outputs(), **kw)
    assert ls.launches == [("gemm", 1)] * 3
    # an invalid problem is refused before anything is launched
    from pantomatrix_amd._lib import EmageKernelError
    operands, kw, outputs = probs[0]
    with pytest.raises(EmageKernelError):
        ops.gemm_grouped(H2, [dict(a=operands[0], w=operands[1], out=outputs()[0], **{**kw, "cp": kw["cp"] + 8})] * 2)


@pytest.mark.parametrize("m", [4096, 200, 64])
def test_gemm_layernorm_fold(m):
    """The LayerNorm fold of emage_gemm_problem (round 6): a sub-layer chain of a post-norm transformer layer —
        s1 = x + a W_o^T + b_o (st_out)  ->  q = LN(s1) W_q^T + b_q  and  [Q | K | V^T] = LN(s1) W_qkv^T + b  (ln fold, row-major AND V^T tiles)
        ->  s2 = LN(s1) + a2 W_2^T + b_2 (folded residual, st_out again)
    against float64 torch (nn.LayerNorm semantics) and against the CPU stand-in of the same arithmetic; single launches and, at M = 4096,
    the grouped form (bit-identical to the single launches)."""
    from pantomatrix_amd.modeling_emage_audio import _Packed
    d, t = 768, 8
    g = _g(300 + m)
    x = torch.randn(m, d, generator=g) * 1.5 + 0.3
    a = torch.randn(m, d, generator=g)
    a2 = torch.randn(m, d, generator=g)
    lin = lambda n: (torch.randn(n, d, generator=g) / math.sqrt(d), torch.randn(n, generator=g) * 0.1)
    (wo, bo), (wq, bq), (wqkv, bqkv), (w2, b2) = lin(d), lin(d), lin(3 * d), lin(d)
    gamma, beta = 1.0 + 0.2 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    # float64 reference
    s1 = x.double() + a.double() @ wo.double().t() + bo.double()
    ln1 = torch.nn.functional.layer_norm(s1, (d,), gamma.double(), beta.double(), 1e-5)
    q_ref = ln1 @ wq.double().t() + bq.double()
    qkv_ref = ln1 @ wqkv.double().t() + bqkv.double()
    s2_ref = ln1 + a2.double() @ w2.double().t() + b2.double()
    fold = lambda w, b: _Packed._fold_norm(type("P", (), {"p": {"n.weight": gamma, "n.bias": beta}})(), w, b, "n")

    def run(mod, dev, grouped=False):
        mv = lambda v: v.to(dev)
        pack = lambda w: tuple(mv(v) if torch.is_tensor(v) else v for v in ops.split_f16_weights_h2(w))
        wo_p, wo_s = pack(wo)
        (wq_f, bq_f, cq), (wqkv_f, bqkv_f, cqkv) = fold(wq, bq), fold(wqkv, bqkv)
        wq_p, wq_s = pack(wq_f)
        wqkv_p, wqkv_s = pack(wqkv_f)
        w2_p, w2_s = pack(w2)
        s1_img, st1 = torch.zeros(m, d, device=dev), torch.zeros(m, d // 32, 2, device=dev)
        mod.gemm(H2, mv(ops.h2_pack(a)), wo_p, mv(bo), None, mv(ops.h2_pack(x)), s1_img, None, None, n=d, cp=d, w_scale=wo_s, res_h2=True, stats_out=st1)
        q = torch.zeros(m, d, device=dev)
        qk, vt = torch.zeros(m, 2 * d, device=dev), torch.zeros(m // t, d, 32, device=dev)
        s2_img, st2 = torch.zeros(m, d, device=dev), torch.zeros(m, d // 32, 2, device=dev)
        calls = [lambda: mod.gemm(H2, s1_img, wq_p, mv(bq_f), None, None, None, q, None, n=d, cp=d, w_scale=wq_s, ln=(st1, mv(cq))),
                 lambda: mod.gemm(H2, s1_img, wqkv_p, mv(bqkv_f), None, None, None, qk, vt, n=3 * d, cp=d, w_scale=wqkv_s, t_col0=2 * d, t_rows=t, ln=(st1, mv(cqkv))),
                 lambda: mod.gemm(H2, mv(ops.h2_pack(a2)), w2_p, mv(b2), None, s1_img, s2_img, None, None, n=d, cp=d, w_scale=w2_s, res_h2=True,
                                  res_ln=(st1, mv(gamma), mv(beta)), stats_out=st2)]
        if grouped:
            with ops.lockstep() as ls:
                for c in calls:
                    with ls.chain():
                        c()
            assert ls.launches == [("group", 3)]
        else:
            for c in calls:
                c()
        return ops.h2_unpack(s1_img), st1, q, qk, vt[:, :, :t], ops.h2_unpack(s2_img), st2

    got = run(ops, DEV)
    torch.cuda.synchronize()
    ref = run(F, "cpu")
    names = ("s1", "st1", "q", "qk", "vt", "s2", "st2")
    for nm, gt, rf in zip(names, got, ref):
        if nm.startswith("st"):                     # {mean, M2} per 32 columns: M2 is a sum of 32 squares of O(1) values
            _cmp(f"fold.{nm}.mean", gt[..., 0], rf[..., 0], atol=2e-6)
            _cmp(f"fold.{nm}.m2", gt[..., 1], rf[..., 1], atol=1e-4, rtol=1e-5)
        else:
            _cmp(f"fold.{nm}", gt, rf, atol=3e-5, rtol=1e-5)
    _cmp("fold.q vs float64", got[2], q_ref, atol=4e-5, rtol=1e-5)
    _cmp("fold.qk vs float64", got[3], qkv_ref[:, :2 * d], atol=4e-5, rtol=1e-5)
    _cmp("fold.vt vs float64", got[4], qkv_ref[:, 2 * d:].reshape(m // t, t, d).permute(0, 2, 1), atol=4e-5, rtol=1e-5)
    _cmp("fold.s2 vs float64", got[5], s2_ref, atol=4e-5, rtol=1e-5)
    # the statistics the consumers merge equal the rows' LayerNorm statistics
    mu, rstd = F._merge_row_stats(got[1].cpu(), 1e-5)
    _cmp("fold.mu", mu, s1.mean(dim=1), atol=2e-6)
    _cmp("fold.rstd", rstd, 1.0 / torch.sqrt(s1.var(dim=1, unbiased=False) + 1e-5), atol=0.0, rtol=2e-6)
    if m == 4096:
        again = run(ops, DEV, grouped=True)
        torch.cuda.synchronize()
        for nm, a_, b_ in zip(names, got, again):
            assert torch.equal(a_, b_), nm


SPLITK_CASES = [
    # name, M-structure (nb, lin, lout), cin, n, taps, stride, pad, flags
    ("sk_out_proj", (1, 64, 64), 768, 768, 1, 1, 0, dict(bias=True, res="h2", want="f32")),
    ("sk_qkv_vt", (1, 64, 64), 768, 2304, 1, 1, 0, dict(bias=True, vt=1536, want="f32")),
    ("sk_ffn2", (1, 64, 64), 1536, 768, 1, 1, 0, dict(bias=True, res="h2", want="both")),
    ("sk_conv3", (1, 64, 64), 256, 256, 3, 1, 1, dict(bias=True, slope=0.2, n_store=256)),
    ("sk_conv3_res_t17", (3, 17, 17), 256, 256, 3, 1, 1, dict(bias=True, res="lo", n_store=256)),
    ("sk_head", (2, 100, 100), 768, 256, 1, 1, 0, dict(bias=True, want="both")),
    ("sk_ragged_n", (1, 70, 70), 512, 106, 1, 1, 0, dict(bias=True, slope=0.1, n_store=128)),
]


@pytest.mark.parametrize("case", SPLITK_CASES, ids=[c[0] for c in SPLITK_CASES])
def test_gemm_split_k_fixup(case):
    """Round 6: launches of few rows (ONE clip: M = 64) lent scratch memory (`splitk=`) cut their K range into slices that meet INSIDE the launch —
    the last block of a tile adds the slices in slice order and runs the ordinary epilogue (emage_gemm_problem: sk_ws / sk_count).  Every
    output form against the CPU restatement and the unsplit launch (fp32 summation order differs: 2e-5), the same bits on every run, the
    counters left at zero."""
    name, (nb, lin, lout), cin, n, taps, stride, pad, fl = case
    scratch = torch.empty(8 << 20, dtype=torch.float32, device=DEV)
    counters = torch.zeros(256, dtype=torch.int32, device=DEV)

    class WithScratch:
        @staticmethod
        def gemm(*a, **k):
            return ops.gemm(*a, splitk=(scratch, counters), **k)

    got = _run_h2_gemm(case, WithScratch, DEV)
    again = _run_h2_gemm(case, WithScratch, DEV)
    plain = _run_h2_gemm(case, ops, DEV)
    torch.cuda.synchronize()
    assert int(counters.abs().sum()) == 0
    ref = _run_h2_gemm(case, F, "cpu")
    for nm, gt, ag, pl, rf in zip(("out", "out_f32", "out_t"), got[:3], again[:3], plain[:3], ref[:3]):
        if gt is not None:
            width = got[3] if nm == "out" else gt.shape[-1]
            _cmp(f"{name}.{nm}", gt[..., :width], rf[..., :width], atol=2e-5, rtol=1e-5)
            _cmp(f"{name}.{nm} vs unsplit", gt[..., :width], pl[..., :width], atol=2e-5, rtol=1e-5)
            assert torch.equal(gt, ag), (name, nm)
    # the launch really was split: with the K range in one piece the bits are those of the plain launch, with slices they (almost surely) are not
    differs = any(gt is not None and not torch.equal(gt, pl) for gt, pl in zip(got[:3], plain[:3]))
    assert differs, "the launch was not split"


def test_count_nonfinite_multi_equals_the_single_launches():
    """`emage_count_nonfinite_multi` (round 6: the clip runner's health check is ONE launch): the count over several tensors of very different
    sizes (an empty chunk tail, one element, more than one 64 K chunk, more tensors than one launch takes) equals the sum of the single launches."""
    g = _g(77)
    sizes = [1, 7, 65536, 65537, 300000, 4096 * 256, 3, 120 * 165 * 64] + [100 + i for i in range(12)]
    xs = []
    for i, n in enumerate(sizes):
        x = torch.randn(n, generator=g)
        idx = torch.randint(0, n, (min(n, 1 + i),), generator=g)
        x[idx] = float("inf") if i % 2 else float("nan")
        xs.append(x.to(DEV))
    want = torch.zeros(1, dtype=torch.int32, device=DEV)
    for x in xs:
        ops.count_nonfinite(x, want)
    got = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.count_nonfinite_multi(xs, got)
    assert int(got) == int(want) == sum(int((~torch.isfinite(x)).sum()) for x in xs)
    clean = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.count_nonfinite_multi([torch.randn(1000, generator=g).to(DEV), torch.zeros(70000, device=DEV)], clean)
    assert int(clean) == 0


def test_gemm_ws_split_k_is_deterministic_and_fp32_grade():
    """VERDICT round 4 next #2b: `emage_gemm_ws` — the weight-gradient shapes of the training step (bare contractions, K = 3 584 rows of a
    56-clip batch, few tiles) cut into K-slices stored as planes of a caller-owned workspace and added in slice order: (i) fp32-grade
    against float64, like the single-slice kernel; (ii) the SAME BITS on every run (the fp32-atomic form of emage_gemm is not);
    (iii) accumulating form (res == out_f32): added onto the destination's contents; (iv) a workspace too small for two slices falls back
    to emage_gemm's behaviour; ragged N; (v) shapes that are not split are untouched by the workspace."""
    g = torch.Generator().manual_seed(5)
    ws = torch.empty(8 << 20, dtype=torch.float32, device=DEV)
    for n_out, k_in, kc in ((768, 768, 3584), (768, 1536, 3584), (256, 768, 3584), (768, 250, 2048), (1536, 768, 3584)):
        a = torch.randn(n_out, kc, generator=g).to(DEV)
        x = torch.randn(k_in, kc, generator=g).to(DEV)
        a_h2, x_h2 = ops.h2_pack(a), ops.h2_pack(x)
        ref = a.double() @ x.double().t()
        tol = 2e-5 * float(ref.abs().max()) * 4
        ldo = ops.round_up(k_in, 4)
        outs = []
        for rep in range(3):
            out = torch.full((n_out, ldo), float("nan"), device=DEV)[:, :k_in]
            ops.gemm(H2, a_h2, x_h2, None, None, None, None, out, None, n=k_in, cp=kc, w_scale=16.0, a_scale=16.0, workspace=ws)
            outs.append(out.clone())
        assert float((outs[0].double() - ref).abs().max()) <= tol, (n_out, k_in)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (n_out, k_in)
        base = torch.randn(n_out, ldo, generator=g).to(DEV)[:, :k_in]
        acc = torch.empty(n_out, ldo, device=DEV)[:, :k_in]             # rows on 16-byte boundaries, like the gradient buckets' views
        acc.copy_(base)
        ops.gemm(H2, a_h2, x_h2, None, None, acc, None, acc, None, n=k_in, cp=kc, w_scale=16.0, a_scale=16.0, workspace=ws)
        assert float((acc.double() - (base.double() + ref)).abs().max()) <= tol, (n_out, k_in)
        small = torch.empty(1024, dtype=torch.float32, device=DEV)          # not even one plane: emage_gemm's path
        out_s = torch.empty(n_out, ldo, device=DEV)[:, :k_in]
        ops.gemm(H2, a_h2, x_h2, None, None, None, None, out_s, None, n=k_in, cp=kc, w_scale=16.0, a_scale=16.0, workspace=small)
        assert float((out_s.double() - ref).abs().max()) <= tol, (n_out, k_in)
    # not a split-K shape (an activation GEMM with bias): the workspace changes no bit
    a = ops.h2_pack(torch.randn(4096, 768, generator=g).to(DEV))
    w, wsc = ops.split_f16_weights_h2((torch.randn(768, 768, generator=g) / 27.0).to(DEV))
    bias = torch.randn(768, generator=g).to(DEV)
    o1, o2 = torch.empty(4096, 768, device=DEV), torch.empty(4096, 768, device=DEV)
    ops.gemm(H2, a, w, bias, None, None, o1, None, None, n=768, cp=768, w_scale=wsc)
    ops.gemm(H2, a, w, bias, None, None, o2, None, None, n=768, cp=768, w_scale=wsc, workspace=ws)
    assert torch.equal(o1.view(torch.int32), o2.view(torch.int32))


def test_weight_packing_on_the_device_equals_the_tensor_arithmetic():
    """`ops.split_f16_weights` / `split_f16_weights_h2` on device tensors run as ONE launch per operand (`emage_f16x3_pack_weights`, the
    activation-cast kernel) — a training step re-packs every weight behind each Adam update.  The images must be the bits the host-side
    tensor arithmetic produces (what the CPU tests and every earlier golden were built on): random weights, a given and a derived scale,
    strided rows, zeros, values that overflow the fp16 hi plane (inf, and inf - inf = NaN in the lo plane) and NaN."""
    def same_planes(got, want):                  # fp16 planes bit for bit; a NaN is a NaN (payload / sign differ between CPU and device)
        a, b = got.cpu().view(torch.float16), want.view(torch.float16)
        na, nb = torch.isnan(a), torch.isnan(b)
        return torch.equal(na, nb) and torch.equal(torch.where(na, torch.zeros_like(a), a).view(torch.int16), torch.where(nb, torch.zeros_like(b), b).view(torch.int16))

    g = _g(5)
    for n, k in ((1, 32), (40, 192), (768, 768), (256, 1536), (67, 64)):
        w = torch.randn(n, k, generator=g) / math.sqrt(k)
        w[0, 0], w[-1, -1] = 0.0, 1e-30
        if n > 2:
            w[1, 3], w[2, 5], w[2, 6] = 1e9, float("nan"), -float("inf")
        for scale in (None, 2.0 ** 14):
            for fn in (ops.split_f16_weights, ops.split_f16_weights_h2):
                if scale is None and n > 2:                        # the derived scale needs a finite maximum
                    wc = torch.nan_to_num(w, nan=0.0, posinf=1.0, neginf=-1.0).clamp(-4.0, 4.0)
                else:
                    wc = w
                want, s_cpu = fn(wc, scale)
                got, s_dev = fn(wc.to(DEV), scale)
                assert s_cpu == s_dev and got.shape == want.shape and got.dtype == torch.float32
                assert same_planes(got, want), (fn.__name__, n, k, scale)
        base = torch.randn(n, k + 64, generator=g)
        view = base.to(DEV)[:, 32:32 + k]                              # rows strided and offset by 128 bytes
        for fn in (ops.split_f16_weights, ops.split_f16_weights_h2):
            want, _ = fn(base[:, 32:32 + k].contiguous(), 2.0 ** 10)
            got, _ = fn(view, 2.0 ** 10)
            assert same_planes(got, want), (fn.__name__, "strided", n, k)


def test_grad_prep_equals_the_separate_launches():
    """`emage_grad_prep` (one pass over a Linear's output gradient: activation backward, both EMAGE_H2 gradient operands, the bias
    gradient) against the launches it replaces — `act_backward`, `h2_cast`, `h2_cast(transpose=True)`, `col_sum`: the two images bit for
    bit (zero tails included), the column sums to float64-summation-order accuracy, with and without an activation, ragged sizes,
    strided views, accumulation into an existing gradient, and each output on its own."""
    g = _g(11)
    for m, c, slope in ((3584, 768, None), (130, 256, 0.1), (70, 337, 0.0), (64, 64, None), (1, 8, 0.2), (200, 1536, 0.0)):
        base = torch.randn(m, c + 24, generator=g).to(DEV)
        dy = base[:, 8:8 + c]                                            # strided rows
        y = torch.randn(m, c, generator=g).to(DEV)
        y[0, 0] = 0.0                                                    # y == 0 takes the slope, as in act_backward
        n_store, m_store, scale = ops.round_up(c, 64), ops.round_up(m, 64), 1024.0
        dpre = dy if slope is None else ops.act_backward(dy.contiguous(), y, slope)
        want_h = ops.h2_cast(dpre, n_store, scale=scale)
        want_t = ops.h2_cast(dpre, m_store, scale=scale, transpose=True)
        want_b = ops.col_sum(dpre.contiguous())
        got_h, got_t, got_b = ops.grad_prep(dy, None if slope is None else y, 0.0 if slope is None else slope, scale, n_store=n_store, m_store=m_store)
        torch.cuda.synchronize()
        # plane for plane as fp16 VALUES: the lo plane of an exact zero (dy * slope 0) may carry either sign (-0 - -0 through different
        # instruction selections of the same split), which no product can tell apart
        same = lambda a, b: torch.equal(a.view(torch.float16), b.view(torch.float16))
        assert same(got_h, want_h), (m, c, "row-major image")
        assert same(got_t, want_t), (m, c, "transposed image")
        tol = 1e-6 * float(dpre.abs().sum(0).max()) + 1e-30
        assert float((got_b - want_b).abs().max()) <= tol, (m, c, "bias gradient")
        acc = torch.full((c,), 3.0, device=DEV)
        only_t = ops.grad_prep(dy, None if slope is None else y, 0.0 if slope is None else slope, scale, m_store=m_store, bias_grad=acc, accumulate=True)
        assert only_t[0] is None and same(only_t[1], want_t)
        assert float((acc - 3.0 - want_b).abs().max()) <= tol + 1e-6
        only_h = ops.grad_prep(dy, None, 0.0, scale, n_store=n_store, want_bias=False)
        assert only_h[1] is None and only_h[2] is None
        assert same(only_h[0], ops.h2_cast(dy, n_store, scale=scale))


@pytest.mark.parametrize("k", [0, 4, 7])
def test_activation_shift_kernels(k):
    """Round 6: EMAGE_H2_SHIFT(k) (include/emage_hip.h) — every entry point that writes or reads an ACTIVATION image takes the model's scale
    2^(4 - k) from its dtype code.  Values up to 3000 x 2^k (beyond the k = 0 range for k > 0) through cast_pad / pack_motion / gather_rows /
    layernorm / add / attention / gemm (operand, H2 residual, H2 output, the LayerNorm fold's raw-sum residual): images bit-equal to the host
    packing at that scale where the kernel only stores, values against the CPU stand-ins elsewhere; k = 0 is the plain code."""
    dt, sc = ops.h2_shifted(k), ops.act_scale(ops.h2_shifted(k))
    assert sc == 16.0 * 2.0 ** -k and (k != 0 or dt == H2)
    g = _g(900 + k)
    amp = 3000.0 * 2.0 ** k
    m, d, t = 128, 768, 8
    x = torch.randn(m, d, generator=g)
    x[::7, ::5] *= amp / 8                                   # a heavy tail: most values O(1), some near the top of the range
    x = x.clamp(-amp, amp)
    xd = x.to(DEV)
    # pure stores: the image IS the host packing at that scale
    img = ops.cast_pad(dt, xd, d)
    assert torch.equal(img.cpu(), ops.h2_pack(x, sc)), "cast_pad"
    assert bool(torch.isfinite(ops.h2_unpack(img, sc)).all())
    table = torch.randn(50, 256, generator=g) * amp / 8
    idx = torch.randint(0, 50, (m,), generator=g)
    assert torch.equal(ops.gather_rows(table.to(DEV), idx.to(DEV), dt, 256).cpu(), ops.h2_pack(table[idx], sc)), "gather_rows"
    motion, mask = torch.randn(2, 16, 337, generator=g) * amp / 16, (torch.rand(2, 16, 337, generator=g) < 0.3).float()
    emb = torch.randn(1, 337, generator=g)
    got = ops.pack_motion(dt, motion.to(DEV), mask.to(DEV), emb.to(DEV), 384)
    assert torch.equal(got.cpu(), F.pack_motion(dt, motion, mask, emb, 384)), "pack_motion"
    # layernorm: float32 in, image (+ float32 twin) out
    gamma, beta = (1.0 + 0.2 * torch.randn(d, generator=g)) * amp / 64, 0.1 * torch.randn(d, generator=g)
    y, yf = torch.zeros(m, d, device=DEV), torch.zeros(m, d, device=DEV)
    ops.layernorm(dt, xd, gamma.to(DEV), beta.to(DEV), 1e-5, None, yf, y)
    assert torch.equal(y.cpu(), ops.h2_pack(yf.cpu(), sc)), "layernorm image vs its float32 twin"
    _cmp("shift.layernorm", yf, torch.nn.functional.layer_norm(x, (d,), gamma, beta, 1e-5), atol=amp * 2e-6, rtol=2e-6)
    # add: image + float32 -> image + float32
    o, of = torch.zeros(m, d, device=DEV), torch.zeros(m, d, device=DEV)
    ops.add(dt, img, 0.25 * xd, out_f32=of, out=o, h2_operands=(0,))
    assert torch.equal(o.cpu(), ops.h2_pack(of.cpu(), sc)), "add image vs its float32 twin"
    _cmp("shift.add", of, ops.h2_unpack(img.cpu(), sc) + 0.25 * x, atol=0.0, rtol=1e-6)
    # attention: float32 q / k / v^T, image out
    b, h, hd = m // t, 4, d // 4
    q, kk = torch.randn(m, d, generator=g), torch.randn(m, d, generator=g)
    vt = torch.zeros(b, d, 32)
    vt[:, :, :t] = torch.randn(b, d, t, generator=g) * amp / 4
    att, att_ref = torch.zeros(m, d, device=DEV), torch.zeros(m, d)
    ops.attention(dt, q.to(DEV), kk.to(DEV), vt.to(DEV), d, att, b, h, t, t, hd)
    F.attention(dt, q, kk, vt, d, att_ref, b, h, t, t, hd)
    _cmp("shift.attention", ops.h2_unpack(att.cpu(), sc), ops.h2_unpack(att_ref, sc), atol=amp * 3e-6, rtol=1e-5)
    # gemm: image operand (a_scale defaults to the image scale), H2 residual, image + float32 outputs, statistics; then the folded forms on the raw sum
    w, bias = torch.randn(d, d, generator=g) / math.sqrt(d), torch.randn(d, generator=g)
    w_p, w_s = ops.split_f16_weights_h2(w)

    def chain(mod, dev):
        mv = lambda v: v.to(dev)
        s_img, s_f, st = torch.zeros(m, d, device=dev), torch.zeros(m, d, device=dev), torch.zeros(m, d // 32, 2, device=dev)
        mod.gemm(dt, mv(ops.h2_pack(x, sc)), mv(w_p), mv(bias), None, mv(ops.h2_pack(x, sc)), s_img, s_f, None, n=d, cp=d, w_scale=w_s, res_h2=True, stats_out=st)
        s2_img = torch.zeros(m, d, device=dev)
        g1, b1 = mv(1.0 + 0.1 * torch.randn(d, generator=_g(5))), mv(0.1 * torch.randn(d, generator=_g(6)))
        mod.gemm(dt, mv(ops.h2_pack(q, sc)), mv(w_p), mv(bias), None, s_img, s2_img, None, None, n=d, cp=d, w_scale=w_s, res_h2=True, res_ln=(st, g1, b1))
        return s_img, s_f, st, s2_img

    got, ref = chain(ops, DEV), chain(F, "cpu")
    torch.cuda.synchronize()
    assert torch.equal(got[0].cpu(), ops.h2_pack(got[1].cpu(), sc)), "gemm image vs its float32 twin"
    _cmp("shift.gemm.sum", got[1], ref[1], atol=amp * 4e-6, rtol=1e-5)
    _cmp("shift.gemm.mean", got[2][..., 0], ref[2][..., 0], atol=amp * 4e-6, rtol=1e-5)
    _cmp("shift.gemm.res_ln", ops.h2_unpack(got[3].cpu(), sc), ops.h2_unpack(ref[3], sc), atol=2e-4 + amp * 1e-7, rtol=2e-5)
    with pytest.raises(Exception):
        ops.cast_pad(F32 | (3 << 8), xd, d)                  # a shift on any other storage type is an invalid code

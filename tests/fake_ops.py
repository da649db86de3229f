"""TEST-ONLY stand-in for pantomatrix_amd.ops: each C-ABI entry point of include/emage_hip.h restated with
torch on the CPU, operating on the SAME tensor views / strides / packed weights the real wrappers receive.

Purpose: exercise the HOST logic (weight packing, buffer views, launch sequence, window schedule) of
pantomatrix_amd against the oracle in the `-m "not gpu"` suite, where no MI355X exists.  It is installed by
monkeypatching inside tests only; the product has no switch that reaches this file and no CPU fallback.
"""
from __future__ import annotations

import contextlib
import math

import torch

from pantomatrix_amd import modeling_emage_audio as M
from pantomatrix_amd import ops
from pantomatrix_amd._lib import BF16, F32, F16X3, H2

TD = {F32: torch.float32, BF16: torch.bfloat16, F16X3: torch.float32, H2: torch.float32}
CALLS = []


def _leaky(v, s):
    return torch.where(v > 0, v, v * s)


def unsplit_f16_weights(packed, n, k):
    """Inverse of ops.split_f16_weights: (N, K) float32-typed image -> (hi, lo) fp32 planes of W * w_scale in natural k order."""
    planes = packed.view(torch.float16).reshape(n, k // 32, 2, 4, 2, 4)          # [kt][plane][g][g2][e]
    nat = planes.permute(2, 0, 1, 4, 3, 5).reshape(2, n, k).float()                # k = 32 kt + 16 g2 + 4 g + e
    return nat[0], nat[1]


def h2_planes(img2d, cols, scale=None):
    """(rows, >= cols) float32-typed EMAGE_H2 image (csrc/h2.h: 32-byte groups [8 hi | 8 lo]) -> (hi, lo) fp32 planes of x * scale."""
    assert cols % 8 == 0 and img2d.stride(1) == 1 and img2d.stride(0) % 8 == 0 and img2d.storage_offset() % 8 == 0
    rows = img2d.shape[0]
    g = torch.as_strided(img2d, (rows, cols), (img2d.stride(0), 1)).contiguous().view(torch.float16).reshape(rows, cols // 8, 2, 8).float()
    return g[:, :, 0, :].reshape(rows, cols), g[:, :, 1, :].reshape(rows, cols)


def h2_values(img2d, cols, scale=ops.A_SCALE_F16X3):
    hi, lo = h2_planes(img2d, cols)
    return (hi + lo) / scale


def h2_store(dst2d, values, scale=ops.A_SCALE_F16X3):
    """Write fp32 `values` (rows, cols), cols % 8 == 0, into the first cols logical columns of the H2 image dst2d."""
    rows, cols = values.shape
    torch.as_strided(dst2d, (rows, cols), (dst2d.stride(0), 1))[:] = ops.h2_pack(values.float(), scale)


def _h2s(dtype):
    """A dtype code as the kernels take it -> (plain code, activation-image scale): include/emage_hip.h EMAGE_H2_SHIFT(k) = x * 2^(4 - k)."""
    k = dtype >> 8
    assert k == 0 or ((dtype & 0xff) == H2 and 0 < k <= 12), dtype
    return dtype & 0xff, ops.A_SCALE_F16X3 * 2.0 ** -k


def _merge_row_stats(st, eps):
    """(M, P, 2) partials {mean, M2} over 32 columns each -> (mu, rstd) per row (what the consuming kernel's Chan merge computes)."""
    mean_p, m2_p = st[:, :, 0].double(), st[:, :, 1].double()
    mu = mean_p.mean(dim=1)
    m2 = m2_p.sum(dim=1) + 32.0 * ((mean_p - mu[:, None]) ** 2).sum(dim=1)
    return mu.float(), (1.0 / torch.sqrt(m2 / (32.0 * st.shape[1]) + eps)).float()


def gemm(dtype, a, w, bias=None, slope=None, res=None, out=None, out_f32=None, out_t=None, *, n, cp,
         n_store=0, t_col0=0, t_rows=0, res_first=False, taps=1, stride=1, pad=0, lin=None, lout=None, m=None,
         k_real=None, w_scale=1.0, a_scale=None, res_h2=False, workspace=None, ln=None, res_ln=None, stats_out=None, ln_eps=1e-5, splitk=None):
    CALLS.append("gemm")
    dtype, h2sc = _h2s(dtype)                        # activation images written / read as a residual: x * h2sc; the operand A: a_scale (default: the same)
    if splitk is not None:                           # scratch the kernel may use for an in-launch split-K: the counters must come and stay zero
        assert dtype == H2 and splitk[1].dtype == torch.int32 and int(splitk[1].abs().sum()) == 0
    if ln is not None or res_ln is not None or stats_out is not None:
        assert dtype == H2 and taps == 1 and workspace is None, "the LayerNorm fold: EMAGE_H2 Linears only"
    m = a.shape[0] if m is None else m
    lin = m if lin is None else lin
    lout = m if lout is None else lout
    assert a.dtype == TD[dtype] and w.dtype == TD[dtype] and w.shape == (n, taps * cp), (a.dtype, w.shape, n, taps, cp)
    assert a.stride(1) == 1 and a.stride(0) >= cp and m % lout == 0 and cp % 64 == 0
    assert a.data_ptr() % 16 == 0 and (a.stride(0) * a.element_size()) % 16 == 0
    nb = m // lout
    assert a.shape[0] >= nb * lin, (a.shape, nb, lin)
    # read exactly what the kernel reads: cp columns from the row start, even beyond a.shape[1]
    if dtype == H2:
        a_hi, a_lo = h2_planes(torch.as_strided(a, (nb * lin, cp), (a.stride(0), 1)), cp)
        af = torch.cat([a_hi, a_lo], dim=1)          # both planes ride through the row gather below
    else:
        af = torch.as_strided(a, (nb * lin, cp), (a.stride(0), 1)).float()
    assert bool(torch.isfinite(af.sum())), "padded channels of A must be finite"          # an inf / NaN anywhere reaches the sum
    if taps == 1 and stride == 1 and pad == 0 and lin == lout:
        x = af[:m]                                               # a Linear: the rows as they are
    else:
        rows = torch.arange(m)
        b_, l_ = rows // lout, rows % lout
        cols = []
        for tap in range(taps):
            pos = l_ * stride + tap - pad
            valid = (pos >= 0) & (pos < lin)
            r = (b_ * lin + pos.clamp(0, lin - 1))
            cols.append(af[r] * valid[:, None].float())
        x = torch.cat(cols, dim=1)                               # (M, taps*cp)
    if dtype == H2:
        # both operands arrive pre-split: A planes hold x * 16, W planes w * w_scale (natural k order)
        sa = h2sc if a_scale is None else a_scale
        xt = x.view(m, taps, 2, cp)
        xh, xl = xt[:, :, 0].reshape(m, taps * cp).double(), xt[:, :, 1].reshape(m, taps * cp).double()
        wh, wl = h2_planes(w, taps * cp)
        wh, wl = wh.double(), wl.double()
        v = ((xh @ wh.t() + (xh @ wl.t() + xl @ wh.t())) / (sa * w_scale)).float()
    elif dtype == F16X3:
        # the kernel's arithmetic: fp16 hi/lo planes of A * a_scale against the host-split planes of W * w_scale,
        # hi*hi + hi*lo + lo*hi (the lo*lo term is dropped), accumulated wide, scaled back by exact powers of two
        sa = ops.A_SCALE_F16X3 if a_scale is None else a_scale
        wh, wl = unsplit_f16_weights(w, n, taps * cp)
        xs = x * sa
        xh = xs.to(torch.float16).float()
        xl = (xs - xh).to(torch.float16).float()
        assert torch.isfinite(xh).all(), "activation overflows fp16 after a_scale"
        xh, xl, wh, wl = xh.double(), xl.double(), wh.double(), wl.double()
        v = ((xh @ wh.t() + (xh @ wl.t() + xl @ wh.t())) / (sa * w_scale)).float()
    else:
        v = x @ w.float().t()
    if ln is not None:                               # folded LayerNorm of the operand: v = rstd (x W'^T - mu c) (+ bias' below)
        st, cvec = ln
        assert st.shape[1] * 32 == cp and cvec.shape == (n,)
        mu, rstd = _merge_row_stats(st[:m], ln_eps)
        v = rstd[:, None] * (v - mu[:, None] * cvec[None, :])
    if bias is not None:
        v = v + bias
    rv = 0.0
    if res is not None:
        assert res.stride(1) == 1
        if res_h2:
            assert dtype == H2
            n8 = (n + 7) // 8 * 8
            rv = h2_values(torch.as_strided(res, (m, n8), (res.stride(0), 1)), n8, h2sc)[:, :n]
        else:
            rv = torch.as_strided(res, (m, n), (res.stride(0), 1)).float()
        if res_ln is not None:                       # the residual is a folded LayerNorm of the raw sum just read
            st, gamma, beta = res_ln
            assert res_h2 and st.shape[1] * 32 == n
            mu, rstd = _merge_row_stats(st[:m], ln_eps)
            rv = (rv - mu[:, None]) * rstd[:, None] * gamma[None, :] + beta[None, :]
    if res_first:
        v = v + rv
    if slope is not None:
        assert slope.shape == (n,)
        v = _leaky(v, slope)
    if not res_first:
        v = v + rv
    ncol_n = n if out_t is None else t_col0
    if stats_out is not None:                        # partial row statistics {mean, M2} over 32 columns each of the values stored
        assert out_t is None and n % 64 == 0 and stats_out.shape[1:] == (n // 32, 2)
        g = v[:, :n].reshape(m, n // 32, 32)
        mean = g.mean(dim=2)
        stats_out[:m, :, 0] = mean
        stats_out[:m, :, 1] = ((g - mean[:, :, None]) ** 2).sum(dim=2)
    if out is not None and dtype == H2:
        width = (max(ncol_n, n_store) + 7) // 8 * 8
        assert out.stride(0) >= width and out.stride(0) % 8 == 0
        full = torch.zeros(m, width)
        full[:, :ncol_n] = v[:, :ncol_n]
        h2_store(out, full, h2sc)
    elif out is not None:
        assert out.dtype == TD[dtype]
        o = torch.as_strided(out, (m, max(ncol_n, n_store)), (out.stride(0), 1))
        o[:, :ncol_n] = v[:, :ncol_n].to(TD[dtype])
        if n_store > n:
            o[:, n:n_store] = 0
    if out_f32 is not None:
        torch.as_strided(out_f32, (m, ncol_n), (out_f32.stride(0), 1))[:] = v[:, :ncol_n]
    if out_t is not None:
        nt = n - t_col0
        tp = out_t.shape[-1]
        assert m % t_rows == 0
        nb_t = m // t_rows
        o = torch.as_strided(out_t, (nb_t, nt, tp), (nt * tp, tp, 1))
        o[:, :, :t_rows] = v[:, t_col0:].reshape(nb_t, t_rows, nt).permute(0, 2, 1).to(TD[dtype])


def attention(dtype, q, k, vt, vt_rows, out, b, h, tq, tk, hd):
    dtype, h2sc = _h2s(dtype)
    if dtype == H2:          # float32 q / k / v^T, the output as an H2 image
        o = torch.empty(b * tq, h * hd)
        attention(F16X3, q, k, vt, vt_rows, o, b, h, tq, tk, hd)
        h2_store(out, o, h2sc)
        return
    CALLS.append("attention")
    tp = vt.shape[-1]
    assert tp % 32 == 0 and tp >= tk and tk <= 128 and hd == 192
    qf = torch.as_strided(q, (b * tq, h * hd), (q.stride(0), 1)).float().view(b, tq, h, hd).transpose(1, 2)
    kf = torch.as_strided(k, (b * tk, h * hd), (k.stride(0), 1)).float().view(b, tk, h, hd).transpose(1, 2)
    v_all = torch.as_strided(vt, (b, h * hd, tp), (vt_rows * tp, tp, 1)).float()
    assert torch.isfinite(v_all).all(), "V^T padding must be finite"
    vf = v_all[:, :, :tk].reshape(b, h, hd, tk).transpose(2, 3)
    p = torch.softmax((qf @ kf.transpose(-1, -2)) * (1.0 / math.sqrt(hd)), dim=-1)
    if dtype == BF16:
        p = p.to(torch.bfloat16).float()
    o = (p @ vf).transpose(1, 2).reshape(b * tq, h * hd)
    torch.as_strided(out, (b * tq, h * hd), (out.stride(0), 1))[:] = o.to(TD[dtype])


def transpose(x, out=None):
    CALLS.append("transpose")
    r = x.t().contiguous()
    if out is not None:
        out[:, :r.shape[1]].copy_(r)              # `out` may be wider (zero-padded contraction length)
        return out
    return r


def col_sum(x, y=None, out=None, accumulate=False, defer=None):
    CALLS.append("col_sum")
    s = (x.double() * (y.double() if y is not None else 1.0)).sum(0).float()
    if out is None:
        return s
    out.copy_(out + s if accumulate else s)
    return out


def act_backward(dy, y, slope, out=None):
    CALLS.append("act_backward")
    r = dy * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, slope))
    if out is not None:
        out.copy_(r)
        return out
    return r


def grad_prep(dy, y=None, slope=0.0, scale=1.0, n_store=None, m_store=None, bias_grad=None, accumulate=False, want_bias=True, defer=None):
    CALLS.append("grad_prep")
    dpre = dy if y is None else dy * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, slope))
    out_h = h2_cast(dpre, n_store, scale) if n_store is not None else None
    out_t = h2_cast(dpre, m_store, scale, transpose=True) if m_store is not None else None
    if not want_bias:
        return out_h, out_t, None
    s = dpre.double().sum(0).float()
    if bias_grad is None:
        return out_h, out_t, s
    bias_grad.copy_(bias_grad + s if accumulate else s)
    return out_h, out_t, bias_grad


def layernorm_backward(x, gamma, dy, eps=1e-5, dgamma=None, dbeta=None, defer=None):
    CALLS.append("layernorm_backward")
    mu = x.mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(((x - mu) ** 2).mean(1, keepdim=True) + eps)
    xh = (x - mu) * rstd
    g = dy * gamma
    dx = rstd * (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True))
    if dgamma is None:
        return dx, (dy * xh).sum(0), dy.sum(0)
    dgamma += (dy * xh).sum(0)
    dbeta += dy.sum(0)
    return dx, dgamma, dbeta


def attention_backward(q, k, vt, vt_rows, pmask, d_out, dq, dk, dv, b, h, tq, tk, hd):
    CALLS.append("attention_backward")
    tp = vt.shape[-1]
    qf = torch.as_strided(q, (b * tq, h * hd), (q.stride(0), 1)).view(b, tq, h, hd).transpose(1, 2)
    kf = torch.as_strided(k, (b * tk, h * hd), (k.stride(0), 1)).view(b, tk, h, hd).transpose(1, 2)
    vf = torch.as_strided(vt, (b, h * hd, tp), (vt_rows * tp, tp, 1))[:, :, :tk].reshape(b, h, hd, tk).transpose(2, 3)
    do = torch.as_strided(d_out, (b * tq, h * hd), (d_out.stride(0), 1)).view(b, tq, h, hd).transpose(1, 2)
    scale = 1.0 / math.sqrt(hd)
    p = torch.softmax((qf @ kf.transpose(-1, -2)) * scale, dim=-1)
    mk = pmask if pmask is not None else torch.ones_like(p)
    d_v = (p * mk).transpose(-1, -2) @ do
    dp = (do @ vf.transpose(-1, -2)) * mk
    ds = p * (dp - (dp * p).sum(-1, keepdim=True)) * scale
    d_q, d_k = ds @ kf, ds.transpose(-1, -2) @ qf
    for dst, src, t in ((dq, d_q, tq), (dk, d_k, tk), (dv, d_v, tk)):
        torch.as_strided(dst, (b * t, h * hd), (dst.stride(0), 1))[:] = src.transpose(1, 2).reshape(b * t, h * hd)


def mse_loss_grad(pred, target, weight):
    CALLS.append("mse_loss_grad")
    return (2.0 * weight / pred.numel()) * (pred - target)


def nll_loss_grad(logits, index, weight):
    CALLS.append("nll_loss_grad")
    g = torch.softmax(logits, dim=1)
    g[torch.arange(logits.shape[0]), index] -= 1.0
    return g * (weight / logits.shape[0])


def _im2col(x, c, taps, stride, pad, lin, lout, nseq):
    xs = x[:, :c].reshape(nseq, lin, c)
    xp = torch.nn.functional.pad(xs, (0, 0, pad, max(0, (lout - 1) * stride - pad + taps - lin)))
    idx = (torch.arange(lout) * stride).view(lout, 1) + torch.arange(taps).view(1, taps)          # padded positions
    return xp[:, idx]                                                                              # (nseq, lout, taps, c)


def im2col_t(x, c, taps, stride, pad, lin, lout, nseq, mp):
    CALLS.append("im2col_t")
    col = _im2col(x, c, taps, stride, pad, lin, lout, nseq).reshape(nseq * lout, taps * c)
    out = torch.zeros(taps * c, mp)
    out[:, :nseq * lout] = col.t()
    return out


def im2col_t_h2(x, c, taps, stride, pad, lin, lout, nseq, mp):
    mark = len(CALLS)
    r = ops.h2_pack(im2col_t(x, c, taps, stride, pad, lin, lout, nseq, mp))
    del CALLS[mark:]
    CALLS.append("im2col_t_h2")
    return r


def col2im(dcol, c, taps, stride, pad, lin, lout, nseq):
    CALLS.append("col2im")
    d = dcol.reshape(nseq, lout, taps, c)
    width = max(lin + 2 * pad, (lout - 1) * stride + taps)
    dxp = torch.zeros(nseq, width, c)
    for tap in range(taps):
        dxp[:, tap:tap + (lout - 1) * stride + 1:stride] += d[:, :, tap]
    return dxp[:, pad:pad + lin].reshape(nseq * lin, c).contiguous()


def bn_backward(x, stats, gamma, dy, eps=1e-5):
    CALLS.append("bn_backward")
    m = x.shape[0]
    rstd = 1.0 / torch.sqrt(stats[1] + eps)
    xh = (x - stats[0]) * rstd
    dg, db = (dy.double() * xh.double()).sum(0).float(), dy.double().sum(0).float()
    return gamma * rstd * (dy - db / m - xh * dg / m), dg, db


def bn_backward_sums(x, stats, dy, eps=1e-5):
    CALLS.append("bn_backward_sums")
    xh = (x - stats[0]) / torch.sqrt(stats[1] + eps)
    return (dy.double() * xh.double()).sum(0).float(), dy.double().sum(0).float()


def bn_backward_apply(x, stats, gamma, dy, sums, count, eps=1e-5):
    CALLS.append("bn_backward_apply")
    rstd = 1.0 / torch.sqrt(stats[1] + eps)
    xh = (x - stats[0]) * rstd
    return gamma * rstd * (dy - sums[1] / count - xh * sums[0] / count)


def wav_conv_in_backward(dy, wav, lout, taps, stride, pad):
    CALLS.append("wav_conv_in_backward")
    b, l = wav.shape
    col = _im2col(wav.reshape(b * l, 1), 1, taps, stride, pad, l, lout, b).reshape(b * lout, taps)
    return (dy.double().t() @ col.double()).float()


def count_nonfinite_multi(xs, counter):
    CALLS.append("count_nonfinite_multi")
    for x in xs:
        counter[:1] += int((~torch.isfinite(x)).sum())


def count_nonfinite(x, counter):
    CALLS.append("count_nonfinite")
    counter += int((~torch.isfinite(x)).sum())


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr=1.5e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    CALLS.append("adam_step")
    step = int(step)
    g = grad + weight_decay * param if weight_decay else grad
    exp_avg.copy_(beta1 * exp_avg + (1 - beta1) * g)
    exp_avg_sq.copy_(beta2 * exp_avg_sq + (1 - beta2) * g * g)
    b1, b2 = 1 - beta1 ** step, 1 - beta2 ** step
    param.copy_(param - (lr / b1) * exp_avg / (exp_avg_sq.sqrt() / math.sqrt(b2) + eps))


def adam_multi(tab, step, lr=1.5e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, grad_scale=1.0, zero_grad=False, skip=None):
    """emage_adam_multi == emage_adam_step on every tensor of the table (grad * grad_scale first, gradient cleared behind the update);
    a non-zero `skip` word leaves parameters and moments untouched."""
    mark = len(CALLS)
    step = int(step)
    skipped = skip is not None and int(skip) != 0
    for p, g, m, v in tab.keep:
        if not skipped:
            adam_step(p, g * grad_scale, m, v, step, lr, beta1, beta2, eps, weight_decay)
        if zero_grad:
            g.zero_()
    del CALLS[mark:]
    CALLS.append("adam_multi")


def dropout_mask(out, p, seed, mask_id, step):
    CALLS.append("dropout_mask")
    out.copy_(torch.from_numpy(ops.philox_dropout_reference(out.numel(), p, int(seed), int(mask_id), int(step))).view(out.shape))
    return out


def loss_workspace(device):
    return torch.zeros(1024, dtype=torch.float64)


def loss_check(workspace):
    assert int(workspace.view(torch.int32)[-2]) == 0


def mse_loss(pred, target, weight, loss, workspace):
    CALLS.append("mse_loss")
    loss += weight * ((pred - target) ** 2).double().mean()


def nll_loss(logits, index, weight, loss, workspace):
    CALLS.append("nll_loss")
    lp = torch.log_softmax(logits, dim=1)
    loss += weight * (-lp.gather(1, index.view(-1, 1)).double().mean())


def attention_dropout(dtype, q, k, vt, vt_rows, out, b, h, tq, tk, hd, pmask):
    CALLS.append("attention_dropout")
    tp = vt.shape[-1]
    assert dtype != BF16 and tuple(pmask.shape) == (b, h, tq, tk)
    qf = torch.as_strided(q, (b * tq, h * hd), (q.stride(0), 1)).float().view(b, tq, h, hd).transpose(1, 2)
    kf = torch.as_strided(k, (b * tk, h * hd), (k.stride(0), 1)).float().view(b, tk, h, hd).transpose(1, 2)
    v_all = torch.as_strided(vt, (b, h * hd, tp), (vt_rows * tp, tp, 1)).float()
    vf = v_all[:, :, :tk].reshape(b, h, hd, tk).transpose(2, 3)
    p = torch.softmax((qf @ kf.transpose(-1, -2)) * (1.0 / math.sqrt(hd)), dim=-1) * pmask
    o = (p @ vf).transpose(1, 2).reshape(b * tq, h * hd)
    torch.as_strided(out, (b * tq, h * hd), (out.stride(0), 1))[:] = o.to(TD[dtype])


def bn_stats(x, running_mean=None, running_var=None, momentum=0.1):
    CALLS.append("bn_stats")
    m = x.shape[0]
    xd = x.double()
    mean = xd.mean(0)
    var = ((xd - mean) ** 2).mean(0)
    if running_mean is not None:
        running_mean.copy_((1 - momentum) * running_mean + momentum * mean.float())
    if running_var is not None:
        running_var.copy_((1 - momentum) * running_var + momentum * (var * m / max(m - 1, 1)).float())
    return mean.float(), var.float()


def bn_apply(x, stats, gamma, beta, out, *, slope=1.0, sc=None, sc_bn=None, eps=1e-5):
    CALLS.append("bn_apply")
    v = (x - stats[0]) / torch.sqrt(stats[1] + eps) * gamma + beta
    if sc is not None:
        r = sc
        if sc_bn is not None:
            r = (sc - sc_bn[0]) / torch.sqrt(sc_bn[1] + eps) * sc_bn[2] + sc_bn[3]
        v = v + r
    out.copy_(torch.nn.functional.leaky_relu(v, slope))
    return out


def mul_add(a, mask, b=None, out=None, *, mask_t_rows=0):
    CALLS.append("mul_add")
    m, c = a.shape
    if isinstance(mask, ops.PhiloxMask):                  # drawn inside the kernel on the device: the same mask as `dropout_mask` with its key
        mark, key = len(CALLS), mask
        mask = dropout_mask(torch.empty(mask.shape), mask.p, mask.seed, mask.mask_id, mask.step)
        del CALLS[mark:]
        if not getattr(key, "_logged", False):            # one "dropout_mask" per mask SITE (forward and backward share the key object)
            key._logged = True
            CALLS.append("dropout_mask")
    mk = mask
    if mask_t_rows:
        mk = mask.view(mask_t_rows, m // mask_t_rows, c).transpose(0, 1).reshape(m, c)
    v = a * mk
    if b is not None:
        v = v + b
    out = torch.empty_like(a) if out is None else out
    out.copy_(v)
    return out


def layernorm(dtype, x, gamma, beta, eps=1e-5, add=None, y_f32=None, y=None):
    CALLS.append("layernorm")
    dtype, h2sc = _h2s(dtype)
    assert x.dtype == TD[dtype] and (add is None or add.dtype == TD[dtype])
    v = torch.nn.functional.layer_norm(x.float(), (x.shape[1],), gamma, beta, eps)
    if add is not None:
        v = v + add.float()
    if y_f32 is not None:
        y_f32[:] = v
    if y is not None:
        if dtype == H2:      # x / add / y_f32 are float32, y the H2 copy
            h2_store(y, v, h2sc)
        else:
            y[:] = v.to(TD[dtype])


def add(dtype, a, b, c=None, out_f32=None, out=None, mod_b=0, mod_c=0, h2_operands=()):
    CALLS.append("add")
    dtype, h2sc = _h2s(dtype)
    m, n = a.shape
    r = torch.arange(m)
    val = lambda t, bit: h2_values(t, n, h2sc) if (dtype == H2 and bit in h2_operands) else t.float()
    v = val(a, 0) + val(b, 1)[r % mod_b if mod_b else r]
    if c is not None:
        v = v + val(c, 2)[r % mod_c if mod_c else r]
    if out_f32 is not None:
        out_f32[:] = v
    if out is not None:
        if dtype == H2:
            h2_store(out, v, h2sc)
        else:
            out[:] = v.to(TD[dtype])


def pack_motion(dtype, motion, mask, emb, n_store, seed=None):
    CALLS.append("pack_motion")
    dtype, h2sc = _h2s(dtype)
    b, t, c = motion.shape
    assert mask.shape == motion.shape and motion.stride(2) == 1 and mask.stride(2) == 1
    assert (motion.stride(1) == c and mask.stride(1) == c) or t == 1
    assert b == 1 or motion.stride(0) == mask.stride(0), "the kernel takes one clip stride for motion and mask"
    motion, mask = motion.clone(), mask.clone()
    if seed is not None:                          # M:386-391: masked seed frames take the carried-over motion and count as unmasked
        pre = seed.shape[1]
        assert seed.shape == (b, pre, c) and seed.stride(2) == 1 and (seed.stride(1) == c or pre == 1)
        motion[:, :pre] = torch.where(mask[:, :pre] == 0, motion[:, :pre], seed)
        mask[:, :pre] = 0
    out = torch.zeros(b * t, n_store, dtype=TD[dtype])
    out[:, :c] = torch.where(mask == 1, emb.expand_as(motion), motion).reshape(b * t, c).to(TD[dtype])
    return ops.h2_pack(out, h2sc) if dtype == H2 else out


def cast_pad(dtype, src2d, n_store, out=None):
    CALLS.append("cast_pad")
    dtype, h2sc = _h2s(dtype)
    m, c = src2d.shape
    full = torch.zeros(m, n_store, dtype=TD[dtype])
    full[:, :c] = src2d.to(TD[dtype])
    if dtype == H2:
        full = ops.h2_pack(full, h2sc)
    if out is None:
        return full
    assert out.shape == (m, n_store)
    out[:] = full                                  # in place when out aliases src2d (the values were read above)
    return out


def h2_cast(src2d, n_store, scale=1.0, transpose=False):
    CALLS.append("h2_cast")
    v = (src2d.float() * scale)
    v = v.t() if transpose else v
    full = torch.zeros(v.shape[0], n_store)
    full[:, :v.shape[1]] = v
    return ops.h2_pack(full)


def gather_rows(table, idx, dtype, n_store=None):
    CALLS.append("gather_rows")
    dtype, h2sc = _h2s(dtype)
    k, d = table.shape
    n_store = d if n_store is None else n_store
    if idx.dim() == 2:
        ops.index_view(idx)                       # the strides the kernel would be handed must be supported
    out = torch.zeros(idx.numel(), n_store, dtype=TD[dtype])
    out[:, :d] = table[idx.reshape(-1)].to(TD[dtype])
    return ops.h2_pack(out, h2sc) if dtype == H2 else out


def _store_idx(res, out):
    if out is None:
        return res
    ops.index_view(out)
    assert out.numel() == res.numel()
    out.copy_(res.view(out.shape))                # in place into the (B, T) view of the code buffer
    return out


def vq_argmin(z2d, codebook, out=None):
    CALLS.append("vq_argmin")
    d = (torch.sum(z2d ** 2, dim=1, keepdim=True) + torch.sum(codebook ** 2, dim=1)) - 2 * (z2d @ codebook.t())
    return _store_idx(torch.argmin(d, dim=1), out)


def argmax_logsoftmax(logits2d, out=None):
    CALLS.append("argmax_logsoftmax")
    return _store_idx(torch.max(torch.log_softmax(logits2d, dim=1), dim=1)[1], out)


def wav_conv_in(dtype, wav, w, bias, slope, out, lout, stride, pad, nwin=1, hop=0, win_len=None):
    CALLS.append("wav_conv_in")
    assert wav.stride(1) == 1
    if win_len is not None:                       # nwin sliding windows per clip, output sequence i*B + b = window i of clip b
        assert (nwin - 1) * hop + win_len <= wav.shape[1]
        wav = torch.cat([wav[:, i * hop:i * hop + win_len] for i in range(nwin)], dim=0)
    y = torch.nn.functional.conv1d(wav.unsqueeze(1), w.unsqueeze(1), bias, stride=stride, padding=pad)   # (B,C,Lout)
    assert y.shape[2] == lout
    y = _leaky(y, slope.view(1, -1, 1)).permute(0, 2, 1).reshape(-1, w.shape[0])
    out[:] = y.to(TD[dtype])


def merge_parts(face, upper, hands, lower, m, device, want_motion=True):
    CALLS.append("merge_parts")
    from oracle import emage_oracle as orc
    z = lambda n: torch.zeros(m, n)
    to_aa = lambda r6: orc.rotation_6d_to_axis_angle(r6.reshape(m, -1, 6)).reshape(m, -1)
    jaw = to_aa(face[:, :6]) if face is not None else z(3)
    expr = face[:, 6:106].clone() if face is not None else z(100)
    up = to_aa(upper[:, :78]) if upper is not None else z(39)
    ha = to_aa(hands[:, :180]) if hands is not None else z(90)
    lo = to_aa(lower[:, :54]) if lower is not None else z(27)
    tf = lower[:, 54:61] if lower is not None else z(7)
    aa = orc.scatter_joints(up, orc.UPPER_JOINTS) + orc.scatter_joints(ha, orc.HANDS_JOINTS) + orc.scatter_joints(lo, orc.LOWER_JOINTS)
    aa[:, 66:69] = jaw
    motion = torch.cat([orc.axis_angle_to_rotation_6d(aa.reshape(m, 55, 3)).reshape(m, 330), tf], dim=1)
    return aa, motion, expr


def velocity_to_position(vel2d, col0, init, dt, b, t):
    CALLS.append("velocity_to_position")
    from oracle import emage_oracle as orc
    assert init.dim() == 2 and init.shape[0] in (1, b) and init.shape[1] == 3 and init.stride(1) == 1
    init = init.expand(b, 3)
    v = torch.as_strided(vel2d, (b * t, 3), (vel2d.stride(0), 1), vel2d.storage_offset() + col0).reshape(b, t, 3)
    x = orc.velocity2position(v[:, :, 0:1], dt, init[:, 0:1])
    zz = orc.velocity2position(v[:, :, 2:3], dt, init[:, 2:3])
    return torch.cat([x, v[:, :, 1:2], zz], dim=-1)


def rot6d_to_axis_angle(x):
    from oracle import emage_oracle as orc
    return orc.rotation_6d_to_axis_angle(x)


def axis_angle_to_rot6d(x):
    from oracle import emage_oracle as orc
    return orc.axis_angle_to_rotation_6d(x)


def conv_slab(dtype, a, w, bias, slope, res, out, *, nseq, l, taps, pad, w_scale=1.0, a_scale=None):
    """emage_conv_slab == emage_gemm with taps = k, stride 1, the shortcut added before the activation."""
    c = w.shape[0]
    assert c in (64, 128) and bias is not None and slope is not None and out.shape == (nseq * l, c)
    mark = len(CALLS)
    gemm(dtype, a, w, bias, slope, res, out, None, None, n=c, cp=c, res_first=True, taps=taps, stride=1, pad=pad, lin=l, lout=l, m=nseq * l,
         w_scale=w_scale, a_scale=a_scale)
    del CALLS[mark:]
    CALLS.append("conv_slab")
    return out


def wav_block0(dtype, wav, w1, b1, slope1, wds, bds, stride1, pad1, w2, bias2, slope2, taps2, pad2, out, l_out, *, nwin=1, hop=0, win_len=None,
               w_scale=1.0, a_scale=None):
    """emage_wav_block0 == emage_wav_conv_in (conv1 | shortcut, stored in the mode's storage type) + emage_gemm."""
    mark = len(CALLS)
    c = w1.shape[0]
    nseq = nwin * wav.shape[0]
    y0 = torch.zeros(nseq * l_out, 2 * c, dtype=TD[dtype])
    wav_conv_in(dtype, wav, torch.cat([w1, wds], 0), torch.cat([b1, bds]), torch.cat([torch.full((c,), float(slope1)), torch.ones(c)]), y0, l_out,
                stride1, pad1, nwin=nwin, hop=hop, win_len=win_len)
    gemm(dtype, y0[:, :c], w2, bias2, slope2, y0[:, c:], out, None, None, n=c, cp=c, res_first=True, taps=taps2, stride=1, pad=pad2,
         lin=l_out, lout=l_out, m=nseq * l_out, w_scale=w_scale, a_scale=a_scale)
    del CALLS[mark:]
    CALLS.append("wav_block0")
    return out


def lstm_step(dtype, h_prev, w_hh, gates_x, cstate, h_out, *, w_scale=1.0, a_scale=None):
    """emage_lstm_step: gate columns interleaved per hidden unit (4u + g, g = i, f, g, o)."""
    CALLS.append("lstm_step")
    b, hid = cstate.shape
    assert h_prev.stride(1) == 1 and gates_x.stride(1) == 1 and h_out.stride(1) == 1 and gates_x.shape == (b, 4 * hid)
    pre = torch.empty(b, 4 * hid)
    gemm(dtype, h_prev.contiguous(), w_hh, None, None, gates_x.contiguous(), None, pre, None, n=4 * hid, cp=hid, w_scale=w_scale, a_scale=a_scale)
    i, f, g, o = pre.view(b, hid, 4).unbind(dim=2)
    c_new = torch.sigmoid(f) * cstate + torch.sigmoid(i) * torch.tanh(g)
    cstate.copy_(c_new)
    h_out.copy_(torch.sigmoid(o) * torch.tanh(c_new))


def lstm_step_pair(dtype, fwd, bwd, *, a_scale=None):
    """Both directions in one launch == two independent emage_lstm_step problems."""
    mark = len(CALLS)
    for h_prev, w_hh, gates_x, cstate, h_out, w_scale in (fwd, bwd):
        lstm_step(dtype, h_prev, w_hh, gates_x, cstate, h_out, w_scale=w_scale, a_scale=a_scale)
    del CALLS[mark:]
    CALLS.append("lstm_step_pair")


def lstm_layer_supported(dtype, hidden):
    return dtype == F16X3 and hidden in (256, 512)


def lstm_layer_sync(b, hidden, device):
    return torch.zeros(((b + 255) // 256) * 544, dtype=torch.int32)


def lstm_layer_check(sync):
    assert not bool(sync.any())


def lstm_layer(dtype, gates_x, w_hh, w_scale, hseq, sync, *, a_scale=None):
    """The persistent recurrence == T paired steps from a zero state (csrc/lstmseq.hip is bit-identical to that sequence)."""
    assert lstm_layer_supported(dtype, hseq.shape[2] // 2) and sync.dtype == torch.int32
    b, t, h2 = hseq.shape
    hid = h2 // 2
    mark = len(CALLS)
    cstate = torch.zeros(2, b, hid)
    prev = [torch.zeros(b, hid), torch.zeros(b, hid)]
    for s in range(t):
        sf, sb = s, t - 1 - s
        cur = [hseq[:, sf, :hid], hseq[:, sb, hid:]]
        lstm_step_pair(dtype, (prev[0], w_hh[0], gates_x[:, sf, :4 * hid], cstate[0], cur[0], w_scale[0]),
                       (prev[1], w_hh[1], gates_x[:, sb, 4 * hid:8 * hid], cstate[1], cur[1], w_scale[1]), a_scale=a_scale)
        prev = cur
    del CALLS[mark:]
    CALLS.append("lstm_layer")
    return hseq


def softmax2_mix(sel, c1, c2, out):
    CALLS.append("softmax2_mix")
    w = torch.softmax(sel[:, :2], dim=1)
    out.copy_(w[:, 0:1] * c1 + w[:, 1:2] * c2)
    return out


def lstm_inputs(out, speaker_table, speaker_id, seed_motion, pose_dims, seed_frames, src_map, b, t):
    CALLS.append("lstm_inputs")
    f = 0 if speaker_table is None else speaker_table.shape[1]
    n_store = out.shape[1]
    assert n_store >= f + pose_dims + 1 and src_map.dtype == torch.int32 and src_map.numel() == t
    rows = torch.zeros(b, t, n_store)
    if f:
        rows[:, :, :f] = speaker_table[speaker_id.reshape(-1)].view(b, 1, f)
    t_m = t if seed_motion is None else seed_motion.shape[1]
    padded = torch.zeros(b, max(t_m, int(src_map.max()) + 1), pose_dims + 1)       # the reference's seed tensor (D:229-232)
    if seed_motion is not None:
        padded[:, :seed_frames, :pose_dims] = seed_motion[:, :seed_frames]
    padded[:, :seed_frames, pose_dims] = 1
    rows[:, :, f:f + pose_dims + 1] = padded[:, src_map.long()]
    out.copy_(rows.view(b * t, n_store))
    return out


def rot6d_scatter(rot6d2d, slot_of_joint, n_joints=55):
    CALLS.append("rot6d_scatter")
    from oracle import emage_oracle as orc
    m = rot6d2d.shape[0]
    out = torch.zeros(m, n_joints, 3)
    sel = (slot_of_joint >= 0).nonzero().reshape(-1)
    d6 = rot6d2d[:, :].reshape(m, -1, 6)[:, slot_of_joint[sel].long()]
    out[:, sel] = orc.rotation_6d_to_axis_angle(d6)
    return out.reshape(m, n_joints * 3)


_NAMES = ["im2col_t_h2", "grad_prep", "h2_cast", "adam_multi", "dropout_mask", "count_nonfinite", "count_nonfinite_multi", "loss_check", "bn_backward_sums", "bn_backward_apply", "im2col_t", "col2im", "bn_backward", "wav_conv_in_backward", "adam_step", "transpose", "col_sum", "act_backward", "layernorm_backward", "attention_backward", "mse_loss_grad", "nll_loss_grad", "loss_workspace", "mse_loss", "nll_loss", "attention_dropout", "bn_stats", "bn_apply", "mul_add", "conv_slab", "wav_block0", "lstm_step", "lstm_step_pair", "lstm_layer", "lstm_layer_supported", "lstm_layer_sync", "lstm_layer_check", "softmax2_mix", "lstm_inputs", "rot6d_scatter", "gemm", "attention", "layernorm", "add", "pack_motion", "cast_pad", "gather_rows", "vq_argmin",
          "argmax_logsoftmax", "wav_conv_in", "merge_parts", "velocity_to_position", "rot6d_to_axis_angle", "axis_angle_to_rot6d"]


@contextlib.contextmanager
def installed():
    """Patch pantomatrix_amd.ops with the CPU restatements and lift the device check, for the duration of a test."""
    saved = {n: getattr(ops, n) for n in _NAMES}
    saved_require = M._EmageModule.__dict__["_require_device"]      # the one thing lifted: `_engine()` itself (version stamp, operand-scale flags) is the product's
    try:
        for n in _NAMES:
            setattr(ops, n, globals()[n])
        M._EmageModule._require_device = staticmethod(lambda dev: None)
        CALLS.clear()
        yield
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
        M._EmageModule._require_device = saved_require

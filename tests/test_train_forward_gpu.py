"""The train-mode forward on the MI355X (pantomatrix_amd/training.py: batch-statistics BatchNorm, dropout with the
reference's masks) against the CPU training oracle and, through the three forwards of a step and the oracle's loss
functions, against the losses of the REAL reference step in tests/golden/train_step_b2.npz."""
import os

import numpy as np
import pytest
import torch

import common
import train_common as tc
from oracle import emage_train_oracle as tro
from pantomatrix_amd import ops, synthetic, training
from pantomatrix_amd.configuration_emage_audio import EmageAudioConfig

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_bn_kernels():
    g = torch.Generator().manual_seed(0)
    m, c = 5000, 192
    x = (torch.randn(m, 256, generator=g) * 3 + 1.5).to(DEV)[:, 32:32 + c]          # a strided view
    rm, rv = torch.randn(c, generator=g).to(DEV), (torch.rand(c, generator=g) + 0.5).to(DEV)
    rm0, rv0 = rm.clone(), rv.clone()
    mean, var = ops.bn_stats(x, rm, rv, 0.1)
    xd = x.double()
    assert float((mean.double() - xd.mean(0)).abs().max()) < 1e-6 and float((var.double() - xd.var(0, unbiased=False)).abs().max()) < 1e-5
    assert float((rm - (0.9 * rm0 + 0.1 * xd.mean(0).float())).abs().max()) < 1e-6
    assert float((rv - (0.9 * rv0 + 0.1 * xd.var(0, unbiased=True).float())).abs().max()) < 1e-5
    gamma, beta = torch.randn(c, generator=g).to(DEV), torch.randn(c, generator=g).to(DEV)
    sc = torch.randn(m, c, generator=g).to(DEV)
    out = torch.empty(m, c, device=DEV)
    ops.bn_apply(x, (mean, var), gamma, beta, out, slope=0.01, sc=sc, sc_bn=(mean, var, beta, gamma))
    bn = lambda t, w, b: (t - mean) / torch.sqrt(var + 1e-5) * w + b
    ref = torch.nn.functional.leaky_relu(bn(x, gamma, beta) + bn(sc, beta, gamma), 0.01)
    assert float((out - ref).abs().max()) < 1e-5
    ops.bn_apply(x, (mean, var), gamma, beta, out, slope=1.0, sc=sc)
    assert float((out - (bn(x, gamma, beta) + sc)).abs().max()) < 1e-5


def test_mul_add_and_attention_dropout():
    g = torch.Generator().manual_seed(1)
    b, t, c = 3, 7, 40
    a, res = torch.randn(b * t, c, generator=g).to(DEV), torch.randn(b * t, c, generator=g).to(DEV)
    mask_tb = (torch.rand(t, b, c, generator=g) > 0.1).float().to(DEV) / 0.9
    got = ops.mul_add(a, mask_tb.view(t * b, c), res, mask_t_rows=t)
    want = a * mask_tb.permute(1, 0, 2).reshape(b * t, c) + res
    assert torch.equal(got, want)
    assert torch.equal(ops.mul_add(a, want), a * want)
    import fake_ops as F
    from pantomatrix_amd._lib import F16X3, F32
    bsz, h, tq, tk, hd = 2, 4, 64, 65, 192
    for dtype in (F32, F16X3):
        q = torch.randn(bsz * tq, h * hd, generator=g)
        k = torch.randn(bsz * tk, h * hd, generator=g)
        vt = torch.zeros(bsz, h * hd, 96)
        vt[:, :, :tk] = torch.randn(bsz, h * hd, tk, generator=g)
        pm = (torch.rand(bsz, h, tq, tk, generator=g) > 0.1).float() / 0.9
        ref = torch.empty(bsz * tq, h * hd)
        F.attention_dropout(dtype, q, k, vt, h * hd, ref, bsz, h, tq, tk, hd, pm)
        out = torch.empty(bsz * tq, h * hd, device=DEV)
        ops.attention_dropout(dtype, q.to(DEV), k.to(DEV), vt.to(DEV), h * hd, out, bsz, h, tq, tk, hd, pm.to(DEV))
        assert float((out.cpu() - ref).abs().max()) < 5e-5


@pytest.mark.parametrize("precision", ["f16x3", "fp32", "f16x3+h2_forward"])
def test_train_forward_matches_oracle(precision):
    precision, _, h2f = precision.partition("+")            # +h2_forward: round 6's A/B switch (the forward's Linears on EMAGE_H2 operands)
    (audio, spk, motion, mask), ref, masks, ref_stats = tc.oracle_forward(seed=7)
    model, _ = common.product_models(precision=precision, device=DEV)
    fwd = training.TrainForward(model)
    fwd.h2_forward = bool(h2f)
    out, stats = fwd(audio, spk, motion, mask, masks)
    for k in ref:
        err = float((out[k].cpu() - ref[k]).abs().max())
        print(f"{precision} train forward {k}: max|err| {err:.2e}")
        assert err < 3e-4, (k, err)
    for k, v in ref_stats.items():
        got = stats[k].cpu()
        assert (int(got) == int(v)) if k.endswith("num_batches_tracked") else float((got - v).abs().max()) < 1e-5 * max(1.0, float(v.abs().max())), k


def test_loss_kernels():
    g = torch.Generator().manual_seed(2)
    m, c = 130, 256
    pred, tgt = torch.randn(m, 300, generator=g)[:, :c].to(DEV), torch.randn(m, c, generator=g).to(DEV)
    idx = torch.randint(0, c, (m,), generator=g).to(DEV)
    ws = ops.loss_workspace(DEV)
    acc = torch.zeros(1, dtype=torch.float64, device=DEV)
    ops.mse_loss(pred, tgt, 3.0, acc, ws)
    ops.nll_loss(pred, idx, 0.5, acc, ws)
    want = 3.0 * torch.nn.functional.mse_loss(pred, tgt).double() + 0.5 * torch.nn.functional.nll_loss(torch.log_softmax(pred, 1), idx).double()
    assert abs(float(acc) - float(want)) < 1e-5 * abs(float(want))
    assert float(ws[-1]) == 0.0
    ops.nll_loss(pred, torch.full((m,), c, dtype=torch.long, device=DEV), 1.0, acc, ws)       # class index out of range: flagged
    torch.cuda.synchronize()
    assert ws.view(torch.int32)[-2].item() != 0


def test_step_losses_on_device(golden_dir):
    """The whole loss side of a training step on the GPU — targets through the HIP VQ models, three train-mode forwards, the
    loss kernels — with the draws of the reference's generator: the six losses of tests/golden/train_step_b2.npz."""
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    batch, ref, masks, random_mask, ref_stats = tc.oracle_step(int(g["seed"]), int(g["iteration"]))
    model, vq = common.product_models(precision="f16x3", device=DEV)
    got, stats = training.step_losses(training.TrainForward(model), vq, {k: v.to(DEV) for k, v in batch.items()}, int(g["iteration"]),
                                      masks, random_mask.to(DEV))
    for k in ("rec_seed", "cls_seed", "rec_audio", "cls_audio", "rec_mask", "cls_mask", "all"):
        want = float(g["loss_" + k])
        print(f"loss {k}: on device {got[k]:.6f}  reference {want:.6f}")
        assert abs(got[k] - want) < 2e-4 * max(1.0, abs(want)), k
    for k, v in ref_stats.items():
        if not k.endswith("num_batches_tracked"):
            assert float((stats[k].cpu() - v).abs().max()) < 1e-5 * max(1.0, float(v.abs().max())), k


def test_backward_kernels():
    """Each backward building block against torch autograd / the CPU stand-in."""
    import fake_ops as F
    g = torch.Generator().manual_seed(6)
    m, c = 130, 768
    x = torch.randn(m, 800, generator=g)[:, :c]
    assert torch.equal(ops.transpose(x.to(DEV)).cpu(), x.t().contiguous())
    y = torch.randn(m, c, generator=g)
    got = ops.col_sum(x.to(DEV), y.to(DEV)).cpu()
    assert float((got - (x.double() * y.double()).sum(0).float()).abs().max()) < 1e-4
    acc = torch.ones(c, device=DEV)
    ops.col_sum(x.to(DEV), None, out=acc, accumulate=True)
    assert float((acc.cpu() - (1 + x.double().sum(0)).float()).abs().max()) < 1e-4
    assert torch.equal(ops.act_backward(x.contiguous().to(DEV), y.to(DEV), 0.1).cpu(), F.act_backward(x.contiguous(), y, 0.1))
    # LayerNorm
    xs = x.contiguous().clone().requires_grad_(True)
    gamma, beta = torch.randn(c, generator=g).requires_grad_(True), torch.randn(c, generator=g).requires_grad_(True)
    dy = torch.randn(m, c, generator=g)
    torch.nn.functional.layer_norm(xs, (c,), gamma, beta, 1e-5).backward(dy)
    dx, dg, db = ops.layernorm_backward(xs.detach().to(DEV), gamma.detach().to(DEV), dy.to(DEV))
    assert float((dx.cpu() - xs.grad).abs().max()) < 2e-5 and float((dg.cpu() - gamma.grad).abs().max()) < 2e-4 and float((db.cpu() - beta.grad).abs().max()) < 2e-4
    # the fused form (dx + float64 partials of both affine gradients in one kernel) against the separate launches: the same dx bits, the
    # same sums to summation order; accumulation into existing gradients; ragged row counts
    from pantomatrix_amd.ops import _layernorm_backward
    monkey = ops.FUSED_LAYERNORM_BACKWARD
    try:
        for per_block, rows in ((16, 130), (16, 16), (16, 1), (16, 77), (4, 130), (4, 3), (4, 77)):
            ops.FUSED_LAYERNORM_BACKWARD = per_block            # (the default is measured, see ops.py; both forms are kept equal all the same)
            xd, dyd, gd = xs.detach()[:rows].to(DEV), dy[:rows].to(DEV), gamma.detach().to(DEV)
            dx_old, t_old = torch.empty(rows, c, device=DEV), torch.empty(rows, c, device=DEV)
            _layernorm_backward(xd, gd, dyd, 1e-5, dx_old, t_old)
            dx_new, dg_new, db_new = ops.layernorm_backward(xd, gd, dyd)
            assert torch.equal(dx_new, dx_old), rows
            assert float((dg_new - ops.col_sum(t_old)).abs().max()) < 1e-5 * max(1.0, float(t_old.abs().sum(0).max())), rows
            assert float((db_new - ops.col_sum(dyd)).abs().max()) < 1e-5 * max(1.0, float(dyd.abs().sum(0).max())), rows
            acc_g, acc_b = torch.full((c,), 2.0, device=DEV), torch.full((c,), -1.0, device=DEV)
            ops.layernorm_backward(xd, gd, dyd, dgamma=acc_g, dbeta=acc_b)
            assert float((acc_g - 2.0 - dg_new).abs().max()) < 1e-5 and float((acc_b + 1.0 - db_new).abs().max()) < 1e-5, rows
    finally:
        ops.FUSED_LAYERNORM_BACKWARD = monkey
    # attention (with probability dropout), Tk != Tq
    b, h, tq, tk, hd = 2, 4, 64, 65, 192
    q = torch.randn(b * tq, h * hd, generator=g)
    k = torch.randn(b * tk, 2 * h * hd, generator=g)[:, h * hd:]           # a strided key view
    vt = torch.zeros(b, h * hd, 96)
    vt[:, :, :tk] = torch.randn(b, h * hd, tk, generator=g)
    pm = (torch.rand(b, h, tq, tk, generator=g) > 0.1).float() / 0.9
    d_out = torch.randn(b * tq, h * hd, generator=g)
    ref = [torch.zeros(b * tq, h * hd), torch.zeros(b * tk, h * hd), torch.zeros(b * tk, h * hd)]
    F.attention_backward(q, k, vt, h * hd, pm, d_out, *ref, b, h, tq, tk, hd)
    got = [torch.zeros(b * tq, h * hd, device=DEV), torch.zeros(b * tk, 3 * h * hd, device=DEV), torch.zeros(b * tk, h * hd, device=DEV)]
    ops.attention_backward(q.to(DEV), k.to(DEV), vt.to(DEV), h * hd, pm.to(DEV), d_out.to(DEV), got[0], got[1][:, h * hd:2 * h * hd], got[2], b, h, tq, tk, hd)
    for a, r in zip((got[0], got[1][:, h * hd:2 * h * hd], got[2]), ref):
        assert float((a.cpu() - r).abs().max()) < 1e-4 * max(1.0, float(r.abs().max()))
    assert float(got[1][:, :h * hd].abs().max()) == 0.0
    # loss gradients
    pred, tgt = torch.randn(m, 256, generator=g), torch.randn(m, 256, generator=g)
    idx = torch.randint(0, 256, (m,), generator=g)
    assert float((ops.mse_loss_grad(pred.to(DEV), tgt.to(DEV), 3.0).cpu() - F.mse_loss_grad(pred, tgt, 3.0)).abs().max()) < 1e-7
    assert float((ops.nll_loss_grad(pred.to(DEV), idx.to(DEV), 0.5).cpu() - F.nll_loss_grad(pred, idx, 0.5)).abs().max()) < 1e-7


def test_conv_and_bn_backward_kernels():
    """im2col / col2im / BatchNorm backward / first-layer weight gradient against the CPU stand-ins (themselves checked against
    torch autograd in tests/test_train_forward_host.py)."""
    import fake_ops as F
    g = torch.Generator().manual_seed(9)
    for c, taps, stride, pad, lin, nseq in ((64, 15, 1, 7, 1241, 2), (64, 15, 6, 0, 1241, 2), (337, 3, 1, 1, 64, 3)):
        lout = (lin + 2 * pad - taps) // stride + 1
        x = torch.randn(nseq * lin, c + 7, generator=g)[:, :c]
        mp = (nseq * lout + 63) // 64 * 64
        assert torch.equal(ops.im2col_t(x.to(DEV), c, taps, stride, pad, lin, lout, nseq, mp).cpu(), F.im2col_t(x, c, taps, stride, pad, lin, lout, nseq, mp))
        dcol = torch.randn(nseq * lout, taps * c, generator=g)
        got, want = ops.col2im(dcol.to(DEV), c, taps, stride, pad, lin, lout, nseq).cpu(), F.col2im(dcol, c, taps, stride, pad, lin, lout, nseq)
        assert float((got - want).abs().max()) < 1e-5
    m, c = 5000, 128
    x, dy = torch.randn(m, c, generator=g) * 2 + 1, torch.randn(m, c, generator=g)
    gamma = torch.randn(c, generator=g)
    stats = F.bn_stats(x)
    want = F.bn_backward(x, stats, gamma, dy)
    got = ops.bn_backward(x.to(DEV), (stats[0].to(DEV), stats[1].to(DEV)), gamma.to(DEV), dy.to(DEV))
    for a, b in zip(got, want):
        assert float((a.cpu() - b).abs().max()) < 1e-4 * max(1.0, float(b.abs().max()))
    wav, dyw = torch.randn(3, 5000, generator=g), torch.randn(3 * 1638, 128, generator=g)          # (5000 + 3200 - 15) // 5 + 1 = 1638
    got, want = ops.wav_conv_in_backward(dyw.to(DEV), wav.to(DEV), 1638, 15, 5, 1600).cpu(), F.wav_conv_in_backward(dyw, wav, 1638, 15, 5, 1600)
    assert float((got - want).abs().max()) < 1e-4 * float(want.abs().max())
    p, gr = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    mm, vv = torch.rand(1000, generator=g), torch.rand(1000, generator=g)
    ref = [t.clone() for t in (p, mm, vv)]
    F.adam_step(ref[0], gr, ref[1], ref[2], 3)
    dev = [t.to(DEV) for t in (p, mm, vv)]
    ops.adam_step(dev[0], gr.to(DEV), dev[1], dev[2], 3)
    for a, b in zip(dev, ref):
        assert float((a.cpu() - b).abs().max()) < 5e-7


def test_backward_matches_autograd_on_device():
    """TrainForward.backward on the GPU against torch autograd through the training oracle (one forward, random targets):
    every trainable parameter of the forward, front ends included."""
    from test_train_forward_host import _oracle_grads, compare_grads
    (audio, spk, motion, mask), masks, index, latent, ref = _oracle_grads(seed=4)
    model, _ = common.product_models(precision="f16x3", device=DEV)
    fwd = training.TrainForward(model)
    fwd(audio, spk, motion, mask, masks, tape=True)
    grads = fwd.backward(index, latent)
    worst = compare_grads(grads, ref, rel=1e-3, to_cpu=True)
    print(f"backward on device: {len(grads)} parameter gradients, worst relative error outside the WavEncoders {worst:.2e}")


def test_wav_encoder_gradients_block_by_block_on_device():
    """VERDICT round 2, Weak 1(ii): the WavEncoder parameters pinned on the device WITHOUT the loose whole-encoder tolerances — the
    exact-fp32 kernels against float64 autograd, block by block from the output: wherever no LeakyReLU activation changed sign against
    the float64 run, the gradient arriving at the block AND the block's conv / BatchNorm parameter gradients agree to fp32 accuracy
    (tests/train_common.py::wav_encoder_backward_check)."""
    model, _ = common.product_models(precision="fp32", device=DEV)
    res = tc.wav_encoder_backward_check(model, DEV)
    print("WavEncoder backward on the device, block by block:", res)
    assert res["clean_blocks"] >= 1 and res["params_checked"] >= 8, res


def test_training_step_matches_the_reference_in_losses_gradient_norms_and_parameter_sums(golden_dir):
    """One whole optimisation step on the GPU — targets, three forwards, three backwards, Adam, BatchNorm buffers — with the
    reference's draws: the seven losses, every gradient's norm and first entry, and every parameter's sum after the update
    against the REAL reference step (tests/golden/train_step_b2.npz)."""
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    batch, _, masks, random_mask, _ = tc.oracle_step(int(g["seed"]), int(g["iteration"]))
    model, vq = common.product_models(precision="f16x3", device=DEV)
    before = {k: v.clone() for k, v in model._flat_params().items()}
    trainer = training.Trainer(model, vq)
    seen = {}
    losses = trainer.step({k: v.to(DEV) for k, v in batch.items()}, int(g["iteration"]), masks, random_mask.to(DEV),
                          grad_hook=lambda gr: seen.update({k: v.clone() for k, v in gr.items()}))
    for k in ("rec_seed", "cls_seed", "rec_audio", "cls_audio", "rec_mask", "cls_mask", "all"):
        want = float(g["loss_" + k])
        assert abs(losses[k] - want) < 2e-4 * max(1.0, abs(want)), k
    names = [str(n) for n in g["grad_names"]]
    gmax = float(np.max(g["grad_norms"]))
    params = model._flat_params()
    checked, lr = 0, 1.5e-4
    for n, norm, first, shadowed, s in zip(names, g["grad_norms"], g["grad_first"], g["shadowed"], g["param_sum_after"]):
        assert n in seen, n
        if shadowed:
            continue
        wav = n.startswith(("audio_encoder_face.", "audio_encoder_body."))
        gn = float(seen[n].norm())
        assert abs(gn - float(norm)) <= (3e-2 if wav else 5e-3) * float(norm) + 1e-6 * gmax, (n, gn, float(norm))
        if not wav:
            assert abs(float(seen[n].reshape(-1)[0]) - float(first)) <= 5e-3 * float(seen[n].abs().max()) + 1e-6 * gmax, n
        p = params[n]
        assert abs(float(p.double().sum()) - float(s)) <= 3e-5 * p.numel() ** 0.5 + 2e-3 + (0.3 * lr * p.numel() if wav else 0), n
        assert not torch.equal(p, before[n]), n
        checked += 1
    print(f"training step on the GPU: {checked} parameters: gradient norms and post-Adam sums equal the reference's")
    assert checked > 440
    assert torch.equal(params["transformer_en_layer.linear1.weight"], before["transformer_en_layer.linear1.weight"])


def test_captured_training_step_equals_the_eager_step(golden_dir):
    """Trainer.capture / replay (the whole step as one hipGraph: operand re-packing, targets, three forwards + backwards, Adam with
    the step count on the device, BatchNorm buffers) against two eager Trainer.step calls on a twin model with the same masks:
    the same parameters after step 1 and after step 2."""
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    batch, _, masks, random_mask, _ = tc.oracle_step(int(g["seed"]), int(g["iteration"]))
    batch = {k: v.to(DEV) for k, v in batch.items()}
    masks = [[m.to(DEV).contiguous() for m in fm] for fm in masks]
    random_mask = random_mask.to(DEV)
    model_e, vq = common.product_models(precision="fp32", device=DEV)
    model_g, _ = common.product_models(precision="fp32", device=DEV)
    eager, graphed = training.Trainer(model_e, vq), training.Trainer(model_g, vq)
    graphed.capture(batch, random_mask, masks)
    keys = ("face_out_proj.weight", "audio_encoder_body.feat_extractor.0.conv1.weight", "audio_motion_cross_attn.layers.7.linear2.bias",
            "mask_embedding", "audio_encoder_face.feat_extractor.3.bn1.running_var", "motion_encoder.main.0.weight")
    for step in (1, 2):
        le = eager.step(batch, 0, masks, random_mask)
        lg = graphed.replay()
        for k in le:
            assert abs(le[k] - lg[k]) <= 1e-6 * max(1.0, abs(le[k])), (step, k, le[k], lg[k])
        pe, pg = model_e._flat_params(), model_g._flat_params()
        for k in keys:
            assert float((pe[k] - pg[k]).abs().max()) <= 1e-7 * max(1.0, float(pe[k].abs().max())), (step, k)
    want = float(g["loss_all"])
    assert abs(eager.state["face_out_proj.weight"]["step"] - 2) == 0 and graphed.state["face_out_proj.weight"]["step"] == 2


def test_captured_f16x3_step_matches_the_reference(golden_dir):
    """VERDICT round 2, "do this" 4: the step as ONE hipGraph in the split-fp16 mode (forward and backward contractions as f16x3 MFMA,
    operand re-packing with cached scales, multi-tensor Adam with the step count on the device) against the REAL reference step's
    golden: the seven losses and every live parameter's sum after the update."""
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    batch, _, masks, random_mask, _ = tc.oracle_step(int(g["seed"]), int(g["iteration"]))
    batch = {k: v.to(DEV) for k, v in batch.items()}
    masks = [[m.to(DEV).contiguous() for m in fm] for fm in masks]
    model, vq = common.product_models(precision="f16x3", device=DEV)
    before = {k: v.clone() for k, v in model._flat_params().items()}
    trainer = training.Trainer(model, vq)
    random_mask = random_mask.to(DEV)                            # an input buffer of the graph
    trainer.capture(batch, random_mask, masks)
    for k, v in before.items():                                  # capture() ran and undid a warm-up step
        assert torch.equal(model._flat_params()[k], v), k
    losses = trainer.replay()
    for k in ("rec_seed", "cls_seed", "rec_audio", "cls_audio", "rec_mask", "cls_mask", "all"):
        want = float(g["loss_" + k])
        assert abs(losses[k] - want) < 2e-4 * max(1.0, abs(want)), (k, losses[k], want)
    params, lr, checked = model._flat_params(), 1.5e-4, 0
    for n, shadowed, s in zip([str(x) for x in g["grad_names"]], g["shadowed"], g["param_sum_after"]):
        if shadowed:
            continue
        wav = n.startswith(("audio_encoder_face.", "audio_encoder_body."))
        p = params[n]
        assert abs(float(p.double().sum()) - float(s)) <= 3e-5 * p.numel() ** 0.5 + 2e-3 + (0.3 * lr * p.numel() if wav else 0), n
        assert not torch.equal(p, before[n]), n
        checked += 1
    assert checked > 400 and trainer.steps_done == 1


def test_device_dropout_mask_and_multi_tensor_adam_kernels():
    """`emage_dropout_mask` is the documented Philox4x32-10 stream bit for bit (numpy restatement, itself pinned to Random123's known
    answers in tests/test_train_forward_host.py), also with the step read from device memory; `emage_adam_multi` equals
    `emage_adam_step` on every tensor (gradient scaling and clearing included)."""
    from pantomatrix_amd import ops
    for n, p, seed, mid, step in ((1, 0.1, 0, 0, 0), (1023, 0.1, 1234567890123456789, 7, 3), (64 * 64 * 768 + 5, 0.25, 99, 130, 12345)):
        out = torch.empty(n, device=DEV)
        ops.dropout_mask(out, p, seed, mid, step)
        ref = torch.from_numpy(ops.philox_dropout_reference(n, p, seed, mid, step))
        assert torch.equal(out.cpu(), ref), (n, seed, mid, step)
        out2 = torch.empty(n, device=DEV)
        ops.dropout_mask(out2, p, seed, mid, torch.tensor([step], dtype=torch.int32, device=DEV))
        assert torch.equal(out2, out)
    g = torch.Generator().manual_seed(3)
    shapes = [(5,), (4097,), (300, 257), (1, 1), (8192,)]
    mk = lambda: [torch.randn(*sh, generator=g).to(DEV) for sh in shapes]
    p1, gr, m1, v1 = mk(), mk(), [t.abs() * 0.01 for t in mk()], [t.abs() * 0.001 for t in mk()]
    p2, g2, m2, v2 = [t.clone() for t in p1], [t.clone() for t in gr], [t.clone() for t in m1], [t.clone() for t in v1]
    tab = ops.AdamTable(list(zip(p2, g2, m2, v2)), DEV)
    step = torch.tensor([4], dtype=torch.int32, device=DEV)
    ops.adam_multi(tab, step, 1e-3, 0.9, 0.999, 1e-8, 0.01, grad_scale=0.5, zero_grad=True)
    for a, b_, c, d in zip(p1, gr, m1, v1):
        ops.adam_step(a, (b_ * 0.5).contiguous(), c, d, 4, 1e-3, 0.9, 0.999, 1e-8, 0.01)
    for a, b_ in zip(p1 + m1 + v1, p2 + m2 + v2):
        assert torch.equal(a, b_)
    assert all(float(t.abs().max()) == 0.0 for t in g2)


def test_reference_style_training_loop_on_the_device(golden_dir):
    """The reference's loop shape on the MI355X through the PRODUCT classes (Missing #2): model.train(), three
    model(...) forwards, torch losses, ONE loss.backward(), torch.optim.Adam.step() — against the REAL reference step's golden
    (losses, gradient norms, post-Adam parameter sums), the reference's recorded dropout draws injected."""
    import torch.nn.functional as F
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    batch, _, masks, random_mask, _ = tc.oracle_step(int(g["seed"]), int(g["iteration"]))
    batch = {k: v.to(DEV) for k, v in batch.items()}
    model, vq = common.product_models(precision="f16x3", device=DEV)
    cfg = model.config
    with torch.no_grad():
        index, latent, masked_motion = training.targets(vq, batch["motion"], batch["expressions"], batch["trans"], batch["foot_contact"])
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1.5e-4)
    opt.zero_grad()
    model.dropout_masks_override = [[m.to(DEV) for m in fm] for fm in masks]
    bs = masked_motion.shape[0]
    spk = torch.zeros(bs, 1, dtype=torch.long, device=DEV)
    seed_mask = torch.ones_like(masked_motion)
    seed_mask[:, :cfg.seed_frames] = 0
    total, got = 0.0, {}
    for tag, mask, use_audio in (("seed", seed_mask, True), ("audio", random_mask.to(DEV), True), ("mask", random_mask.to(DEV), False)):
        out = model(batch["audio"], spk, masked_motion, mask, use_audio=use_audio)
        rec = sum(getattr(cfg, "l" + q[0]) * F.mse_loss(out[f"rec_{q}"], latent[q]) for q in ("upper", "lower", "hands", "face"))
        cls = sum(getattr(cfg, "c" + q[0]) * F.nll_loss(F.log_softmax(out[f"cls_{q}"], dim=2).reshape(-1, 256), index[q].reshape(-1))
                  for q in ("upper", "lower", "hands", "face"))
        got["rec_" + tag], got["cls_" + tag] = float(rec), float(cls)
        total = total + rec + cls
    total.backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    opt.step()
    model.eval()
    for k in ("rec_seed", "cls_seed", "rec_audio", "cls_audio", "rec_mask", "cls_mask"):
        want = float(g["loss_" + k])
        assert abs(got[k] - want) < 2e-4 * max(1.0, abs(want)), (k, got[k], want)
    gmax = float(np.max(g["grad_norms"]))
    params = model._flat_params()
    checked, lr = 0, 1.5e-4
    for n, norm, shadowed, s in zip([str(x) for x in g["grad_names"]], g["grad_norms"], g["shadowed"], g["param_sum_after"]):
        if shadowed:
            continue
        wav = n.startswith(("audio_encoder_face.", "audio_encoder_body."))
        gn = float(grads[n].norm())
        assert abs(gn - float(norm)) <= (3e-2 if wav else 5e-3) * float(norm) + 1e-6 * gmax, (n, gn, float(norm))
        p = params[n]
        assert abs(float(p.double().sum()) - float(s)) <= 3e-5 * p.numel() ** 0.5 + 2e-3 + (0.3 * lr * p.numel() if wav else 0), n
        checked += 1
    assert checked > 440
    print(f"class-API training loop on the GPU: {checked} parameters match the reference step")


def test_device_drawn_masks_step_and_capture():
    """Without given masks the trainer draws them on the device: reproducible per seed, and the captured step (masks drawn inside the
    graph from the in-graph step counter) equals the eager steps of a twin trainer with the same seed."""
    from test_train_oracle import train_batch
    batch = {k: v.to(DEV) for k, v in train_batch(bs=2).items()}
    random_mask = (torch.rand(2, 64, 337, generator=torch.Generator().manual_seed(5)) < 0.5).float().to(DEV)
    model_e, vq = common.product_models(precision="fp32", device=DEV)
    model_g, _ = common.product_models(precision="fp32", device=DEV)
    eager, graphed = training.Trainer(model_e, vq, seed=11), training.Trainer(model_g, vq, seed=11)
    graphed.capture(batch, random_mask)
    l_first = None
    for step in (1, 2, 3):
        le = eager.step(batch, random_mask=random_mask)
        l_first = le if l_first is None else l_first
        lg = graphed.replay()
        for k in le:
            assert abs(le[k] - lg[k]) <= 1e-6 * max(1.0, abs(le[k])), (step, k, le[k], lg[k])
    pe, pg = model_e._flat_params(), model_g._flat_params()
    for k in ("face_out_proj.weight", "audio_motion_cross_attn.layers.7.linear2.bias", "motion_encoder.main.0.weight"):
        assert float((pe[k] - pg[k]).abs().max()) <= 1e-7 * max(1.0, float(pe[k].abs().max())), k
    other, _ = common.product_models(precision="fp32", device=DEV)
    l_other = training.Trainer(other, vq, seed=12).step(batch, random_mask=random_mask)
    fresh, _ = common.product_models(precision="fp32", device=DEV)
    l_same = training.Trainer(fresh, vq, seed=11).step(batch, random_mask=random_mask)
    assert l_same == l_first and l_same != l_other          # a fresh trainer with seed 11 repeats the first eager step exactly
    # the (T, B, d) masks as Philox KEYS drawn inside `mul_add` (the default) against masks written to memory first: the same bits
    stored, _ = common.product_models(precision="fp32", device=DEV)
    tr = training.Trainer(stored, vq, seed=11)
    tr.fwd.lazy_masks = False
    assert tr.step(batch, random_mask=random_mask) == l_first
    ps = stored._flat_params()
    one, _ = common.product_models(precision="fp32", device=DEV)
    training.Trainer(one, vq, seed=11).step(batch, random_mask=random_mask)
    assert all(torch.equal(v, ps[k]) for k, v in one._flat_params().items())


def test_mul_add_with_the_mask_drawn_inside_the_kernel():
    """`emage_mul_add_philox` (dropout with the keep mask drawn from its Philox key inside the kernel) equals `emage_dropout_mask` +
    `emage_mul_add` bit for bit: both row orders ((B, T) rows against a (T, B, d) mask), with and without the residual, strided
    operands, the step as a host integer and read from device memory."""
    from pantomatrix_amd import ops
    g = torch.Generator().manual_seed(17)
    for m, c, t_rows, seed, mid, step in ((128, 768, 64, 5, 3, 1), (3584, 1536, 64, 1234567890123456789, 200, 77), (6, 8, 0, 2 ** 64 - 3, 0, 0), (130, 256, 0, 9, 41, 3)):
        a = torch.randn(m, c + 8, generator=g).to(DEV)[:, 4:4 + c]
        b = torch.randn(m, c, generator=g).to(DEV)
        shape = (t_rows, m // t_rows, c) if t_rows else (m, c)
        mask = ops.dropout_mask(torch.empty(shape, device=DEV), 0.1, seed, mid, step).view(m, c)
        for res in (None, b):
            want = ops.mul_add(a, mask, res, mask_t_rows=t_rows)
            key = ops.PhiloxMask(shape, 0.1, seed, mid, step, DEV).view(m, c)
            assert torch.equal(ops.mul_add(a, key, res, mask_t_rows=t_rows), want), (m, c, t_rows, res is None)
            key_dev = ops.PhiloxMask(shape, 0.1, seed, mid, torch.tensor([step], dtype=torch.int32, device=DEV), DEV).view(m, c)
            assert torch.equal(ops.mul_add(a, key_dev, res, mask_t_rows=t_rows), want)
        assert torch.equal(ops.PhiloxMask(shape, 0.1, seed, mid, step, DEV).materialize().view(m, c), mask)


def test_batched_finalize_launches_change_no_bit(golden_dir):
    """Round 5: the ~600 finalize launches of a step's bias / affine-gradient reductions are queued and issued 64 at a time
    (`emage_col_sum_finalize_multi`, `ops.FinalizeQueue`).  (i) the kernel: a mixed batch — 70 reductions of both producers
    (`col_sum`, `grad_prep`), writing and accumulating, ragged widths — equals the one-by-one finalize launches bit for bit, and a
    destination queued twice is flushed in between; (ii) the step: the same parameters after Adam, bit for bit, as with
    `defer_finalize = False`."""
    from pantomatrix_amd import ops
    g = torch.Generator().manual_seed(11)
    q = ops.FinalizeQueue()
    ref, got = [], []
    for i in range(70):
        m, c = (3584, 768) if i % 3 == 0 else ((448, 250) if i % 3 == 1 else (3000, 1536))
        x = torch.randn(m, c, generator=g).to(DEV)
        base = torch.randn(c, generator=g).to(DEV)
        acc = bool(i % 2)
        a, b = base.clone(), base.clone()
        if i % 4 < 2:
            ops.col_sum(x, None, out=a, accumulate=acc)
            ops.col_sum(x, None, out=b, accumulate=acc, defer=q)
        else:
            ops.grad_prep(x, None, 0.0, 1.0, None, None, bias_grad=a, accumulate=acc)
            ops.grad_prep(x, None, 0.0, 1.0, None, None, bias_grad=b, accumulate=acc, defer=q)
        ref.append(a)
        got.append(b)
    assert len(q.entries) == 70 - 64                                # one full batch went out on its own
    q.flush()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), i
    d = torch.zeros(768, device=DEV)
    x = torch.randn(3584, 768, generator=g).to(DEV)
    ops.col_sum(x, None, out=d, accumulate=True, defer=q)
    ops.col_sum(x, None, out=d, accumulate=True, defer=q)           # the same destination again: the first entry is flushed first
    assert len(q.entries) == 1
    q.flush()
    one = ops.col_sum(x)
    assert torch.equal(d, one + one)
    gg = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    batch, _, masks, random_mask, _ = tc.oracle_step(int(gg["seed"]), int(gg["iteration"]))
    batch = {k: v.to(DEV) for k, v in batch.items()}
    ends = []
    for defer in (True, False):
        model, vq = common.product_models(precision="f16x3", device=DEV)
        trainer = training.Trainer(model, vq)
        trainer.fwd.defer_finalize = defer
        trainer.step(batch, int(gg["iteration"]), masks, random_mask.to(DEV))
        ends.append({k: v.clone() for k, v in model._flat_params().items() if v.is_floating_point()})
    diff = [k for k in ends[0] if not torch.equal(ends[0][k], ends[1][k])]
    assert not diff, diff[:8]


def test_sync_batchnorm_on_one_device_equals_plain_batchnorm(golden_dir):
    """sync_bn=True with a world of one (no process group): the SyncBatchNorm code path of the forward and of the backward on the
    MI355X gives the step of the plain BatchNorm path (Weak #1 iii)."""
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    g_batch, _, masks, random_mask, _ = tc.oracle_step(int(g["seed"]), int(g["iteration"]))        # the replay the other tests of this file share
    batch = {k: v.to(DEV) for k, v in g_batch.items()}
    seen = []
    for sync in (False, True):
        model, vq = common.product_models(precision="fp32", device=DEV)
        got = {}
        training.Trainer(model, vq, sync_bn=sync).step(batch, 0, masks, random_mask.to(DEV), grad_hook=lambda gr: got.update({k: v.clone() for k, v in gr.items()}))
        seen.append((got, {k: v.clone() for k, v in model._flat_params().items() if "running_" in k}))
    (g0, b0), (g1, b1) = seen
    for k in g0:
        tol = 5e-2 if k.startswith("audio_encoder") else 2e-3
        assert float((g0[k] - g1[k]).norm()) <= tol * float(g0[k].norm()) + 1e-7, k
    for k in b0:
        assert float((b0[k] - b1[k]).abs().max()) <= 1e-5 * max(1.0, float(b0[k].abs().max())), k


def _check_step_against(g, losses, params, before, grads=None, what=""):
    """Losses, (optionally) gradient norms, and post-Adam parameter sums of a device step against a reference-step golden; returns
    the worst ratios to the allowances (printed by the callers: how close to the bound the step sits)."""
    worst = dict(loss=0.0, norm=0.0, psum=0.0)
    for k in ("rec_seed", "cls_seed", "rec_audio", "cls_audio", "rec_mask", "cls_mask", "all"):
        want = float(g["loss_" + k])
        r = abs(losses[k] - want) / (2e-4 * max(1.0, abs(want)))
        worst["loss"] = max(worst["loss"], r)
        assert r < 1, (what, k, losses[k], want)
    gmax, lr, checked = float(np.max(g["grad_norms"])), 1.5e-4, 0
    for n, norm, shadowed, s in zip([str(x) for x in g["grad_names"]], g["grad_norms"], g["shadowed"], g["param_sum_after"]):
        if shadowed:
            continue
        wav = n.startswith(("audio_encoder_face.", "audio_encoder_body."))
        if grads is not None:
            r = abs(grads[n] - float(norm)) / ((3e-2 if wav else 5e-3) * float(norm) + 1e-6 * gmax)
            worst["norm"] = max(worst["norm"], r)
            assert r <= 1, (what, n, grads[n], float(norm))
        p = params[n]
        r = abs(float(p.double().sum()) - float(s)) / (3e-5 * p.numel() ** 0.5 + 2e-3 + (0.3 * lr * p.numel() if wav else 0))
        worst["psum"] = max(worst["psum"], r)
        assert r <= 1, (what, n, float(p.double().sum()), float(s))
        assert not torch.equal(p, before[n]), (what, n)
        checked += 1
    assert checked > 440
    return worst


def test_baseline_batch_step_matches_the_reference_in_losses_gradient_norms_and_parameter_sums(golden_dir):
    """VERDICT round 3, next #1a: BASELINE configs[2] at its PER-GPU BATCH — 56 clips x 64 frames (BatchNorm couples the clips) — against
    the REAL reference's step (tests/golden/train_step_b56.npz, generated by tests/golden/make_golden_train.py 56): the eager f16x3 step
    (seven losses, every gradient norm, post-Adam parameter sums) and the step as ONE hipGraph replay, captured twice on fresh models.
    Round 5: the split-K weight gradients go through emage_gemm_ws (K-slices as workspace planes added in slice order, no fp32 atomics),
    so the two captures must end in the SAME BITS (VERDICT round 4 next #2b / ADVICE: f16x3 gradients are run-to-run reproducible)."""
    g = np.load(os.path.join(golden_dir, "train_step_b56.npz"))
    bs = int(g["bs"])
    batch, oracle_losses, masks, random_mask, _ = tc.oracle_step(int(g["seed"]), int(g["iteration"]), bs=bs)      # CPU oracle: replays the reference's draws
    for k, v in oracle_losses.items():                           # the restatement itself at this batch size
        assert abs(v - float(g["loss_" + k])) <= 2e-4 * max(1.0, abs(float(g["loss_" + k]))), k
    batch = {k: v.to(DEV) for k, v in batch.items()}
    masks = [[m.to(DEV).contiguous() for m in fm] for fm in masks]
    random_mask = random_mask.to(DEV)
    model, vq = common.product_models(precision="f16x3", device=DEV)
    before = {k: v.clone() for k, v in model._flat_params().items()}
    norms = {}
    losses = training.Trainer(model, vq).step(batch, int(g["iteration"]), masks, random_mask,
                                              grad_hook=lambda gr: norms.update({k: float(v.norm()) for k, v in gr.items()}))
    worst = _check_step_against(g, losses, model._flat_params(), before, norms, "eager")
    print(f"eager f16x3 step at {bs} clips vs the reference: worst fraction of the allowance: {worst}")
    del model
    runs = []
    for rep in range(2):
        model, _ = common.product_models(precision="f16x3", device=DEV)
        trainer = training.Trainer(model, vq).capture(batch, random_mask, masks)
        lg = trainer.replay()
        worst = _check_step_against(g, lg, model._flat_params(), before, None, f"captured #{rep}")
        assert trainer.steps_done == 1 and trainer.skipped_steps == 0 and trainer.rescaled == 0 and int(trainer.health) == 0
        runs.append((lg, {k: float(v.double().sum()) for k, v in model._flat_params().items() if v.is_floating_point()},
                     {k: v.clone() for k, v in model._flat_params().items() if v.is_floating_point()}))
        print(f"captured f16x3 step #{rep} at {bs} clips vs the reference: {worst}")
        del trainer, model
        torch.cuda.empty_cache()
    (l0, s0, p0), (l1, s1, p1) = runs
    jitter = max(abs(l0[k] - l1[k]) / max(1.0, abs(l0[k])) for k in l0)
    assert jitter < 1e-6, jitter                                   # forward: no atomics; the losses of step 1 do not depend on dW order at all
    drift = max(abs(s0[k] - s1[k]) for k in s0)
    differing = [k for k in p0 if not torch.equal(p0[k], p1[k])]
    print(f"two captures of the same step: loss jitter {jitter:.2e}, largest parameter-sum difference {drift:.3e}, tensors that differ in any bit: {len(differing)}")
    assert not differing, differing[:8]                            # no atomics anywhere in the step: bit-reproducible


def test_captured_step_health_and_operand_rescaling(golden_dir):
    """VERDICT round 3, next #1b: (i) a weight pushed out of the range its cached power-of-two operand scale was chosen for is reported by
    the device-side flag of the graph's own re-packing; `replay()` re-derives the scales and re-captures, keeping the optimiser state;
    (ii) a weight large enough to overflow the fp16 planes of the backward poisons the gradients: the in-graph count of non-finite
    gradient words makes Adam skip, nothing is written, and the step raises."""
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    batch, _, masks, random_mask, _ = tc.oracle_step(int(g["seed"]), int(g["iteration"]))
    batch = {k: v.to(DEV) for k, v in batch.items()}
    masks = [[m.to(DEV).contiguous() for m in fm] for fm in masks]
    random_mask = random_mask.to(DEV)
    model, vq = common.product_models(precision="f16x3", device=DEV)
    trainer = training.Trainer(model, vq).capture(batch, random_mask, masks)
    l1 = trainer.replay()
    assert trainer.steps_done == 1 and trainer.rescaled == 0
    m1 = trainer.state["face_out_proj.weight"]["exp_avg"].clone()
    name = "audio_motion_cross_attn.layers.3.linear1.weight"
    with torch.no_grad():
        model._flat_params()[name].mul_(4.0)                        # 4x: max |w| x cached scale in [2^14, 2^15) — finite in fp16, flagged
    l2 = trainer.replay()                                            # this step is still computed correctly; behind it the scales are re-derived
    assert trainer.rescaled == 1 and trainer.steps_done == 2 and all(np.isfinite(v) for v in l2.values())
    assert not torch.equal(trainer.state["face_out_proj.weight"]["exp_avg"], m1)
    m2 = trainer.state["face_out_proj.weight"]["exp_avg"].clone()
    l3 = trainer.replay()                                            # the re-captured graph: same state, fresh scales
    assert trainer.rescaled == 1 and trainer.steps_done == 3 and all(np.isfinite(v) for v in l3.values())
    assert not torch.equal(trainer.state["face_out_proj.weight"]["exp_avg"], m2)
    # (ii) overflow: 2^12 x on top pushes the hi plane past 65504 -> inf - inf = NaN in the products
    snap = {k: v.clone() for k, v in model._flat_params().items()}
    with torch.no_grad():
        model._flat_params()[name].mul_(4096.0)
    snap[name] = model._flat_params()[name].clone()
    moments = {k: st["exp_avg"].clone() for k, st in trainer.state.items()}
    with pytest.raises(FloatingPointError, match="non-finite gradient words"):
        trainer.replay()
    assert trainer.steps_done == 3
    after = model._flat_params()
    assert all(torch.equal(after[k], snap[k]) for k in snap)        # parameters and BatchNorm buffers untouched
    assert all(torch.equal(trainer.state[k]["exp_avg"], moments[k]) for k in moments)
    assert all(float(b.abs().max()) == 0.0 for b in trainer.buckets.flat)


def test_train_then_eval_uses_the_updated_weights(golden_dir):
    """ADVICE round 3 (medium #1): the reference's train / val loop — train-mode forward, backward, `optimizer.step()`, then
    `model.eval(); model(...)` — on the device in the configuration where nothing else forced a re-pack (fp32 precision): the eval
    forward after the update equals a fresh model loaded with the updated state dict."""
    g = np.load(os.path.join(golden_dir, "train_step_b2.npz"))
    batch, _, masks, random_mask, _ = tc.oracle_step(int(g["seed"]), int(g["iteration"]))
    model, vq = common.product_models(precision="fp32", device=DEV)
    audio, spk, motion, mask = (x.to(DEV) for x in common.window_inputs(2))
    with torch.no_grad():
        ev0 = model(audio, spk, motion, mask)                      # packs the eval operands (BatchNorm folded)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    model.dropout_masks_override = [[m.to(DEV) for m in masks[0]]]
    out = model(batch["audio"].to(DEV), spk, motion, mask)
    sum(v.square().mean() for v in out.values()).backward()
    opt.step()
    model.eval()
    with torch.no_grad():
        ev1 = model(audio, spk, motion, mask)
        fresh, _ = common.product_models(precision="fp32", device=DEV)
        fresh.load_state_dict(model.state_dict())
        ev2 = fresh(audio, spk, motion, mask)
    for k in ev1:
        assert torch.equal(ev1[k], ev2[k]), k
        assert float((ev1[k] - ev0[k]).abs().max()) > 1e-4, k      # lr 1e-2 moved every head; the BatchNorm buffers moved too

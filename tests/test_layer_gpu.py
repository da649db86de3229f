"""emage_transformer_layer (one launch per transformer layer) against the per-op launch sequence it replaces.
Both run the same tile / attention / LayerNorm routines in the same order, so the bar is bit equality."""
import pytest
import torch

import common
from pantomatrix_amd import modeling_emage_audio as M
from pantomatrix_amd import ops, spec

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def model():
    m, _ = common.product_models(precision="bf16", device=DEV)
    return m


def _inputs(b, t=64, d=768, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b * t, d, generator=g).to(torch.bfloat16).to(DEV)
    mem = torch.randn(b * t, d, generator=g).to(torch.bfloat16).to(DEV)
    add = torch.randn(b * t, d, generator=g).to(torch.bfloat16).to(DEV)
    return x, mem, add


def _both(model, fn):
    outs = []
    for fused in (True, False):
        model.fused_layers = fused
        outs.append(fn())
        torch.cuda.synchronize()
    model.fused_layers = False
    return outs


@pytest.mark.parametrize("b", [1, 3, 8, 64])
@pytest.mark.parametrize("post_add", [False, True])
def test_decoder_layer_matches_per_op_sequence(model, b, post_add):
    cx = M._Ctx(model._engine())
    d, t = model.config.hidden_size, 64
    x, mem, add = _inputs(b, seed=b)
    name = "audio_motion_cross_attn.layers.3"
    nc = spec.N_CROSS_LAYERS

    def run():
        bk, bvt = model._memory_kv(cx, "cross.kv_all", mem, b, t, nc)
        return model._decoder_layer(cx, name, x, b, t, bk[:, 3 * d:4 * d], bvt[:, 3 * d:], nc * d, t,
                                    post_add=add if post_add else None)

    fused, per_op = _both(model, run)
    assert torch.isfinite(fused.float()).all()
    assert torch.equal(fused, per_op), float((fused.float() - per_op.float()).abs().max())


@pytest.mark.parametrize("b", [2, 64])
def test_encoder_layer_matches_per_op_sequence(model, b):
    cx = M._Ctx(model._engine())
    x, _, add = _inputs(b, seed=10 + b)
    fused, per_op = _both(model, lambda: model._encoder_layer(cx, "motion_self_encoder.layers.0", x, b, 64, post_add=add))
    assert torch.equal(fused, per_op), float((fused.float() - per_op.float()).abs().max())


def test_refinement_layer_and_status_word(model):
    """Single-layer memory (the three body_motion_decoder_* layers) + the barrier status word stays clear."""
    cx = M._Ctx(model._engine())
    b, t, d = 5, 64, model.config.hidden_size
    x, mem, _ = _inputs(b, seed=3)
    name = "body_motion_decoder_hands.layers.0"

    def run():
        k1, vt1 = model._memory_kv(cx, name + ".ca.kv", mem, b, t, 1)
        return model._decoder_layer(cx, name, x, b, t, k1, vt1, d, t)

    fused, per_op = _both(model, run)
    assert torch.equal(fused, per_op)
    w = cx.pk.w
    k1, vt1 = model._memory_kv(cx, name + ".ca.kv", mem, b, t, 1)
    out, ws = ops.transformer_layer(cx.dt, x, [w[name + s] for s in (".sa.qkv", ".sa.out", ".ca.q", ".ca.out", ".ff1", ".ff2")],
                                    [w[name + s] for s in (".norm1", ".norm2", ".norm3")], cx.pk.slope(0.0, 1536), b, t,
                                    heads=4, ffn=1536, mem_k=k1, mem_vt=vt1, vt_rows=d, tk=t)
    assert ops.transformer_layer_status(ws, b) == 0
    assert torch.equal(out, per_op)


def test_unsupported_geometry_keeps_the_per_op_sequence(model):
    """Tail windows (T != 64) and the fp32 parity mode never reach the fused kernel."""
    assert not ops.transformer_layer_supported(cx_dt := M._Ctx(model._engine()).dt, 37, 768, 4, 1536)
    assert ops.transformer_layer_supported(cx_dt, 64, 768, 4, 1536, 64)
    assert not ops.transformer_layer_supported(cx_dt, 64, 768, 4, 1536, 20)
    from pantomatrix_amd._lib import F32
    assert not ops.transformer_layer_supported(F32, 64, 768, 4, 1536, 64)


def test_layer_is_deterministic_under_concurrency(model):
    """Two layers in flight on two streams (the face and body lanes do this in forward()) give the single-stream bits."""
    cx = M._Ctx(model._engine())
    b = 64
    x, _, add = _inputs(b, seed=21)
    name = "motion_self_encoder.layers.0"
    model.fused_layers = True
    ref = model._encoder_layer(cx, name, x, b, 64, post_add=add)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(3):
        for s in (s1, s2):
            with torch.cuda.stream(s):
                outs.append(model._encoder_layer(cx, name, x, b, 64, post_add=add))
    torch.cuda.synchronize()
    model.fused_layers = False
    for o in outs:
        assert torch.equal(o, ref)
    assert torch.equal(model._encoder_layer(cx, name, x, b, 64, post_add=add), ref)      # and the per-op bits


def test_whole_clip_with_fused_layers_matches_per_op(model):
    """inference() + decode of a 2-window clip: the fused-layer pipeline gives the per-op pipeline's exact outputs."""
    from pantomatrix_amd import synthetic
    _, vq = common.product_models(precision="bf16", device=DEV)
    audio = synthetic.synthetic_audio(4, synthetic.samples_for_frames(128)).to(DEV)
    spk = torch.zeros(4, 1, dtype=torch.long, device=DEV)
    outs = []
    for fused in (True, False):
        model.fused_layers = fused
        lat = model.inference(audio, spk, vq)
        outs.append(vq.decode(**model._select_codes(lat), get_global_motion=True, ref_trans=torch.zeros(1, 3, device=DEV)))
        torch.cuda.synchronize()
    model.fused_layers = False
    for k in ("motion_axis_angle", "expression", "trans"):
        assert torch.equal(outs[0][k], outs[1][k]), k

"""The autoregressive window schedule of `inference()` (M:343-470) — window boundaries, the audio slice per window, the
4-frame seed carried over through the decode, the masks, the kept frame counts, the optional tail window — swept over
many clip lengths with a cheap stand-in for the network and the decode (the same stand-ins on the product and on the
oracle), so that lengths the golden fixtures do not cover (exact multiples, 1..4 frames past a boundary, clips shorter
than a window, user-supplied motion + mask) are compared frame for frame.  CPU only."""
import pytest
import torch

import common
import fake_ops
from oracle import emage_oracle as orc

KEYS = orc.OUT_KEYS


def _stub_forward(audio, speaker_id, masked_motion, mask, use_audio=True, **_):
    """Frame-local function of exactly what a window is given: the audio slice, the (seeded) motion, the mask."""
    b, t, c = masked_motion.shape
    spf = 16000 // 30
    a = audio[:, :t * spf].reshape(b, t, spf).mean(dim=2, keepdim=True)                  # (B,T,1)
    base = (masked_motion * (1 - mask)).sum(dim=2, keepdim=True) + 10.0 * a + mask.sum(dim=2, keepdim=True) / c
    ramp = torch.linspace(-1, 1, 256).view(1, 1, 256)
    out = {}
    for i, k in enumerate(KEYS):
        out[k] = torch.sin(base * (1.0 + 0.1 * i) + ramp * (3 + i)) + 0.01 * speaker_id.view(b, 1, 1).float()
    return out


def _stub_forward_product(audio, speaker_id, masked_motion, mask, use_audio=True, _seed=None, **_):
    """The product hands forward() the window views plus the seed; the splice of M:386-391 happens in its packing
    kernel.  The stand-in restates that splice, then is the same frame-local function."""
    if _seed is not None:
        pre = _seed.shape[1]
        masked_motion, mask = masked_motion.clone(), mask.clone()
        masked_motion[:, :pre] = torch.where(mask[:, :pre] == 0, masked_motion[:, :pre], _seed)
        mask[:, :pre] = 0
    return _stub_forward(audio, speaker_id, masked_motion, mask, use_audio)


def _stub_decode(codebooks, **kw):
    """Frame-local decode: 337 channels from the code indices; a latent goes through its nearest code first, like
    `decode_from_latent` (M:60-70) — so the index a lean code path hands over decodes like the latent it came from."""
    parts = []
    for p in ("face", "upper", "hands", "lower"):
        v = kw.get(f"{p}_index")
        if v is None:
            v = orc.vq_nearest(kw[f"{p}_latent"], codebooks[p])
        parts.append(v.float().unsqueeze(-1) / 256.0)
    x = torch.cat(parts, dim=2)                                                            # (B,T,4)
    return {"all_motion4inference": torch.cos(x.sum(dim=2, keepdim=True) * torch.arange(1, 338).view(1, 1, 337) * 0.01)}


class _StubVQ:
    def __init__(self, real):
        self._real = real
        if hasattr(real, "vq_model_face"):       # product classes / oracle classes
            self._cb = {p: getattr(real, f"vq_model_{p}").state_dict()["quantizer.embedding.weight"] for p in common.PARTS}
        else:
            self._cb = {p: getattr(real, p).codebook for p in common.PARTS}

    def __getattr__(self, name):                 # config lookups (vae_layer) of the seed-only decode
        return getattr(self._real, name)

    def decode(self, **kw):
        kw.pop("get_global_motion", None), kw.pop("ref_trans", None)
        return _stub_decode(self._cb, **kw)


LENGTHS = [5, 30, 63, 64, 65, 68, 69, 70, 100, 123, 124, 125, 128, 129, 184, 185, 188, 189, 250]


@pytest.mark.parametrize("frames", LENGTHS)
def test_window_schedule_matches_oracle(frames, monkeypatch):
    model, vq = common.product_models(precision="fp32")
    omodel, ovq = common.oracle_models()
    model.hoist_audio = False                    # the stand-in network takes the raw audio slice of each window
    bs = 2 if frames < 130 else 1
    g = torch.Generator().manual_seed(frames)
    audio = torch.randn(bs, frames * 16000 // 30 + 7, generator=g)
    spk = torch.zeros(bs, 1, dtype=torch.long)
    given = frames % 3 == 0                      # every third length: user-supplied motion + mask for a prefix of the clip
    mm = mk = None
    if given:
        n = min(frames, 40)
        mm = torch.randn(bs, n, 337, generator=g)
        mk = (torch.rand(bs, n, 337, generator=g) > 0.5).float()
    monkeypatch.setattr(type(omodel), "forward", lambda self, *a, **k: _stub_forward(*a, **k))
    monkeypatch.setattr(type(model), "forward", lambda self, *a, **k: _stub_forward_product(*a, **k))
    length = audio.shape[1] * 30 // 16000
    window, pre = 64, 4
    rounds, remain = (length - pre) // (window - pre), (length - pre) % (window - pre)
    expect = rounds * (window - pre) + (pre + remain if remain > pre else 0)
    if expect == 0:                              # clips of <= 8 frames yield no window: the reference fails in torch.cat, so do we
        with fake_ops.installed(), torch.no_grad():
            with pytest.raises((RuntimeError, ValueError)):
                omodel.inference(audio, spk, _StubVQ(ovq), masked_motion=mm, mask=mk)
            with pytest.raises((RuntimeError, ValueError)):
                model.inference(audio, spk, _StubVQ(vq), masked_motion=mm, mask=mk)
        return
    with fake_ops.installed(), torch.no_grad():
        ref = omodel.inference(audio, spk, _StubVQ(ovq), masked_motion=mm, mask=mk)
        got = model.inference(audio, spk, _StubVQ(vq), masked_motion=mm, mask=mk)
        codes = model.infer_codes(audio, spk, _StubVQ(vq), masked_motion=mm, mask=mk)
    for k in KEYS:
        assert got[k].shape == ref[k].shape == (bs, expect, 256), (k, got[k].shape, ref[k].shape, expect)
        assert torch.equal(got[k], ref[k]), k
    sel = omodel.select_codes(ref)               # the lean per-window code path selects the same frames
    for p in common.PARTS:
        assert codes[f"{p}_latent"] is None
        if sel[f"{p}_index"] is not None:
            assert torch.equal(codes[f"{p}_index"], sel[f"{p}_index"]), p
        else:                                    # latent-routed part: the lean path returns the index of its nearest code
            want = orc.vq_nearest(sel[f"{p}_latent"], getattr(ovq, p).codebook)
            assert torch.equal(codes[f"{p}_index"], want), p

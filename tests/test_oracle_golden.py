"""Pin the CPU oracle against golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  Runs anywhere (no GPU, no /root/reference)."""
import os

import numpy as np
import pytest
import torch

import common
from oracle import emage_oracle as orc
from pantomatrix_amd import synthetic

ATOL = 2e-4  # fp32 CPU vs fp32 CPU, different BLAS blocking / hosts


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_forward_window(golden_dir):
    g = _load(golden_dir, "forward_b1.npz")
    model, _ = common.oracle_models()
    audio, spk, motion, mask = common.window_inputs(1)
    with torch.no_grad():
        out = model.forward(audio, spk, motion, mask)
        out_na = model.forward(audio, spk, motion, mask, use_audio=False)
    for k in orc.OUT_KEYS:
        np.testing.assert_allclose(out[k].numpy(), g[k], atol=ATOL, rtol=0)
    for k in ("rec_face", "rec_upper", "rec_hands", "rec_lower"):
        np.testing.assert_allclose(out_na[k].numpy(), g["noaudio_" + k], atol=ATOL, rtol=0)
    assert np.abs(g["rec_upper"] - g["noaudio_rec_upper"]).max() > 1e-2  # use_audio matters


@pytest.mark.parametrize("frames,batch,expect", [(128, 2, 120), (70, 1, 70), (129, 1, 129), (310, 1, 310), (40, 1, 40), (64, 1, 60)])
def test_end_to_end_clip(golden_dir, frames, batch, expect):
    g = _load(golden_dir, f"infer_{frames}f_b{batch}.npz")
    model, vq = common.oracle_models()
    audio = synthetic.synthetic_audio(batch, synthetic.samples_for_frames(frames))
    poses, expr, trans = orc.infer_clip(model, vq, audio)
    assert poses.shape == (batch, expect, 165) and expr.shape == (batch, expect, 100) and trans.shape == (batch, expect, 3)
    np.testing.assert_allclose(poses, g["poses"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(expr, g["expressions"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(trans, g["trans"], atol=1e-3, rtol=0)


@pytest.mark.parametrize("layer", [2, 3])
def test_vq_stacks(golden_dir, layer):
    g = _load(golden_dir, f"vq_layer{layer}.npz")
    from pantomatrix_amd.configuration_emage_audio import EmageVQVAEConvConfig, EmageVAEConvConfig
    _, vqc, gc = common.cfg_dicts(vae_layer=layer, global_layer=4 if layer == 2 else 3)
    gen = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for p in common.PARTS:
            cfg = EmageVQVAEConvConfig(**vqc[p])
            m = orc.VQVAE(synthetic.vqvae_state(cfg, p, 0), cfg)
            x = torch.randn(2, 40, cfg.vae_test_dim, generator=gen)
            idx = torch.randint(0, 256, (2, 40), generator=gen)
            z = torch.randn(2, 40, 256, generator=gen)
            assert np.array_equal(m.map2index(x).numpy(), g[f"{p}_map2index"])
            np.testing.assert_allclose(m.encode(x).numpy(), g[f"{p}_pre_latent"], atol=ATOL, rtol=0)
            np.testing.assert_allclose(m.decode(idx).numpy(), g[f"{p}_decode"], atol=ATOL, rtol=0)
            assert np.array_equal(orc.vq_nearest(z, m.codebook).numpy(), g[f"{p}_nearest"])
            np.testing.assert_allclose(m.decode_from_latent(z).numpy(), g[f"{p}_decode_from_latent"], atol=ATOL, rtol=0)
        gcfg = EmageVAEConvConfig(**gc)
        ae = orc.VAE(synthetic.vae_state(gcfg, 0), gcfg)
        x = torch.randn(2, 40, 61, generator=gen)
        np.testing.assert_allclose(ae.forward(x)["rec_pose"].numpy(), g["global_rec_pose"], atol=ATOL, rtol=0)
        vqm = orc.VQModel(None, None, None, None, ae)
        np.testing.assert_allclose(vqm.get_global_motion(x, torch.zeros(1, 3)).numpy(), g["global_trans"], atol=ATOL, rtol=0)


def test_rotations(golden_dir):
    g = _load(golden_dir, "rotations.npz")
    gen = torch.Generator().manual_seed(5)
    d6 = torch.randn(4, 50, 6, generator=gen)
    aa = torch.randn(4, 50, 3, generator=gen) * torch.tensor([1.0, 0.3, 1e-4, 0.0]).view(4, 1, 1)
    np.testing.assert_allclose(orc.rotation_6d_to_axis_angle(d6).numpy(), g["rot6d_to_aa"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(orc.axis_angle_to_rotation_6d(aa).numpy(), g["aa_to_rot6d"], atol=1e-6, rtol=0)
    # behavioural pins the reference implies: identity pose <-> [1,0,0,0,1,0]
    ident = orc.axis_angle_to_rotation_6d(torch.zeros(3))
    assert torch.equal(ident, torch.tensor([1.0, 0, 0, 0, 1, 0]))
    back = orc.axis_angle_to_rotation_6d(orc.rotation_6d_to_axis_angle(d6))
    np.testing.assert_allclose(orc.rotation_6d_to_axis_angle(back).numpy(), g["rot6d_to_aa"], atol=1e-4, rtol=0)


def test_codebook_round_trip():
    """map2index(get_codebook_entry(i)) == i for distinct codebook rows (SURVEY §8c invariant)."""
    cb = torch.randn(256, 256, generator=torch.Generator().manual_seed(3))
    idx = torch.randint(0, 256, (3, 17), generator=torch.Generator().manual_seed(4))
    assert torch.equal(orc.vq_nearest(orc.vq_lookup(idx, cb), cb), idx)


def test_window_schedule():
    """Output length rule of inference(): 60*rounds + (4+remain if remain>4 else 0)."""
    for frames, expect in ((70, 70), (84, 84), (128, 120), (129, 129), (133, 133)):
        rounds, remain = (frames - 4) // 60, (frames - 4) % 60
        assert 60 * rounds + (4 + remain if remain > 4 else 0) == expect


def test_vq_model_api(golden_dir):
    """EmageVQModel.spilt_inputs / map2index / map2latent (M:97-124) and EmageVQVAEConv.forward incl. the Quantizer's
    embedding_loss and perplexity (M:42-46, P:144-156) against the reference's outputs."""
    g = _load(golden_dir, "vq_api.npz")
    _, vq = common.oracle_models()
    rot6d, expr, contact, trans = common.vq_api_inputs()
    with torch.no_grad():
        sp = vq.split_inputs(rot6d, expr, contact, trans)
        idx = vq.map2index(rot6d, expr, contact, trans)
        idx0 = vq.map2index(rot6d, expr)
        lat = vq.map2latent(rot6d, expr, contact, trans)
        for p in common.PARTS:
            assert np.array_equal(sp[p].numpy(), g[f"split_{p}"])
            assert np.array_equal(idx[p].numpy(), g[f"index_{p}"]) and np.array_equal(idx0[p].numpy(), g[f"index0_{p}"])
            np.testing.assert_allclose(lat[p].numpy(), g[f"latent_{p}"], atol=1e-6, rtol=0)
            fw = getattr(vq, p).forward(sp[p])
            np.testing.assert_allclose(fw["poses_feat"].numpy(), g[f"fwd_{p}_poses_feat"], atol=ATOL, rtol=0)
            np.testing.assert_allclose(fw["rec_pose"].numpy(), g[f"fwd_{p}_rec_pose"], atol=ATOL, rtol=0)
            np.testing.assert_allclose(float(fw["embedding_loss"]), float(g[f"fwd_{p}_embedding_loss"]), rtol=1e-4)
            np.testing.assert_allclose(float(fw["perplexity"]), float(g[f"fwd_{p}_perplexity"]), rtol=1e-5)


def test_batch64_indices(golden_dir):
    """BASELINE config 2 (64 x 128-frame clips): the oracle reproduces every VQ code index of the reference's run and the
    decoded motion of the stored clips; the decode of golden indices is what the GPU test uses for the other clips."""
    g = _load(golden_dir, "infer_128f_b64.npz")
    model, vq = common.oracle_models()
    torch.set_num_threads(max(1, min(8, torch.get_num_threads())))
    audio = synthetic.synthetic_audio(64, synthetic.samples_for_frames(128))
    spk = torch.zeros(64, 1, dtype=torch.long)
    with torch.no_grad():
        lat = model.inference(audio, spk, vq)
        sel = model.select_codes(lat)
        face_idx = orc.vq_nearest(lat["rec_face"], vq.face.sd["quantizer.embedding.weight"])
        pred = vq.decode(**sel, get_global_motion=True, ref_trans=torch.zeros(1, 3))
    for p in ("upper", "hands", "lower"):
        assert np.array_equal(sel[f"{p}_index"].numpy(), g[f"index_{p}"].astype(np.int64)), p
    assert np.array_equal(face_idx.numpy(), g["index_face"].astype(np.int64))
    sub = slice(0, 64, 8)
    np.testing.assert_allclose(pred["motion_axis_angle"][sub].numpy(), g["poses_sub"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(pred["expression"][sub].numpy(), g["expressions_sub"], atol=1e-3, rtol=0)
    np.testing.assert_allclose(pred["trans"][sub].numpy(), g["trans_sub"], atol=1e-3, rtol=0)

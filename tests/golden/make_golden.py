"""Generate tests/golden/*.npz by running the REAL reference (/root/reference, read-only) on the
deterministic synthetic weights of `pantomatrix_amd.synthetic` (seed 0) and seeded inputs.

Run in the build container only:  python tests/golden/make_golden.py
The reference cannot travel to the GPU box, these small fixtures do.  Inputs are NOT stored:
tests regenerate them from the same seeds (tests/common.py).
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import reference_harness as rh  # noqa: E402
import common  # noqa: E402
from pantomatrix_amd import synthetic  # noqa: E402


def ref_infer_clip(model, vq, audio):
    """test_emage_audio.py:16-53 with the un-importable bits (librosa, smplx, render) left out."""
    spk = torch.zeros(audio.shape[0], 1, dtype=torch.long)
    with torch.no_grad():
        lat = model.inference(audio, spk, vq, masked_motion=None, mask=None)
        c = model.cfg
        kw = {}
        for p, l, cc in (("face", c.lf, c.cf), ("upper", c.lu, c.cu), ("hands", c.lh, c.ch), ("lower", c.ll, c.cl)):
            kw[f"{p}_latent"] = lat[f"rec_{p}"] if l > 0 and cc == 0 else None
            kw[f"{p}_index"] = torch.max(F.log_softmax(lat[f"cls_{p}"], dim=2), dim=2)[1] if cc > 0 else None
        pred = vq.decode(get_global_motion=True, ref_trans=torch.zeros(1, 1, 3)[:, 0], **kw)
    idx = {p: (kw[f"{p}_index"] if kw[f"{p}_index"] is not None else
               getattr(vq, f"vq_model_{p}").quantizer.map2index(kw[f"{p}_latent"])) for p in common.PARTS}
    return lat, pred, idx


def golden_b64(model, vq):
    """BASELINE config 2 itself: the 64 x 128-frame batch.  All four code-index arrays of all 64 clips (int16) plus the
    decoded poses / expressions / trans of every 8th clip (the decode of the other clips is checked through the
    oracle from the golden indices) — keeps the fixture near 1 MB."""
    a = synthetic.synthetic_audio(64, synthetic.samples_for_frames(128))
    lat, pred, idx = ref_infer_clip(model, vq, a)
    sub = slice(0, 64, 8)
    np.savez_compressed(os.path.join(HERE, "infer_128f_b64.npz"),
                        poses_sub=pred["motion_axis_angle"][sub].numpy(), expressions_sub=pred["expression"][sub].numpy(),
                        trans_sub=pred["trans"][sub].numpy(),
                        **{f"index_{p}": idx[p].numpy().astype(np.int16) for p in common.PARTS})


def golden_long(model, vq):
    """A clip of many windows: 310 frames = 5 full 64-frame windows (stride 60) + a 10-frame tail, so the seed chain
    (last 4 frames of window w -> first 4 of window w+1) is exercised four times and ends in a T+1 audio memory."""
    a = synthetic.synthetic_audio(1, synthetic.samples_for_frames(310))
    lat, pred, idx = ref_infer_clip(model, vq, a)
    np.savez_compressed(os.path.join(HERE, "infer_310f_b1.npz"),
                        poses=pred["motion_axis_angle"].numpy(), expressions=pred["expression"].numpy(),
                        trans=pred["trans"].numpy(), rec_face=lat["rec_face"].numpy(),
                        **{f"index_{p}": idx[p].numpy() for p in common.PARTS})


def golden_vq_api(acfg, vqc, gc):
    """EmageVQModel.spilt_inputs / map2index / map2latent (M:97-124) and EmageVQVAEConv.forward (M:42-46, P:144-156:
    straight-through latents, embedding_loss, perplexity) on seeded rot-6D motion — the calls train_emage_audio.py:149-150
    makes.  Inputs are regenerated from the seed by the tests."""
    _, vq = rh.build_reference(acfg, vqc, gc, seed=0)
    rot6d, expr, contact, trans = common.vq_api_inputs()
    rec = {}
    with torch.no_grad():
        sp = vq.spilt_inputs(rot6d, expr, contact, trans)
        idx = vq.map2index(rot6d, expr, contact, trans)
        lat = vq.map2latent(rot6d, expr, contact, trans)
        idx0 = vq.map2index(rot6d, expr)                      # tar_contact / tar_trans default to zeros
        for p in common.PARTS:
            rec[f"split_{p}"] = sp[p].numpy()
            rec[f"index_{p}"] = idx[p].numpy()
            rec[f"index0_{p}"] = idx0[p].numpy()
            rec[f"latent_{p}"] = lat[p].numpy()
            fw = getattr(vq, f"vq_model_{p}")(sp[p])
            rec[f"fwd_{p}_poses_feat"] = fw["poses_feat"].numpy()
            rec[f"fwd_{p}_rec_pose"] = fw["rec_pose"].numpy()
            rec[f"fwd_{p}_embedding_loss"] = fw["embedding_loss"].numpy()
            rec[f"fwd_{p}_perplexity"] = fw["perplexity"].numpy()
    np.savez_compressed(os.path.join(HERE, "vq_api.npz"), **rec)


def golden_clips(model, vq, which):
    """End-to-end clips: 128 frames (2 windows, no tail), 70 frames (tail of 10 frames -> T+1 audio memory), 129 frames (tail of 9);
    round 3: 40 frames (shorter than a window: only the remainder pass runs, 40 frames out) and 64 frames (exactly one window and no
    remainder pass: 60 frames out)."""
    for frames, batch in which:
        a = synthetic.synthetic_audio(batch, synthetic.samples_for_frames(frames))
        lat, pred, idx = ref_infer_clip(model, vq, a)
        np.savez(os.path.join(HERE, f"infer_{frames}f_b{batch}.npz"),
                 poses=pred["motion_axis_angle"].numpy(), expressions=pred["expression"].numpy(),
                 trans=pred["trans"].numpy(), rec_face=lat["rec_face"].numpy(),
                 **{f"index_{p}": idx[p].numpy() for p in common.PARTS})


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    acfg, vqc, gc = common.cfg_dicts(vae_layer=2)
    model, vq = rh.build_reference(acfg, vqc, gc, seed=0)
    if "--only-long" in sys.argv:
        golden_long(model, vq)
        print("wrote infer_310f_b1.npz")
        return
    if "--only-short" in sys.argv:           # round-3 additions only
        golden_clips(model, vq, ((40, 1), (64, 1)))
        print("wrote infer_40f_b1.npz, infer_64f_b1.npz")
        return
    if "--only-new" in sys.argv:             # round-2 additions only (the round-1 fixtures are unchanged)
        golden_long(model, vq)
        golden_b64(model, vq)
        golden_vq_api(acfg, vqc, gc)
        print("wrote infer_128f_b64.npz, vq_api.npz")
        return
    golden_long(model, vq)
    golden_b64(model, vq)
    golden_vq_api(acfg, vqc, gc)

    # 1. one forward() window, B=1, partly masked motion
    audio, spk, motion, mask = common.window_inputs(1)
    with torch.no_grad():
        out = model(audio, spk, motion, mask, use_audio=True)
        out_na = model(audio, spk, motion, mask, use_audio=False)
    np.savez(os.path.join(HERE, "forward_b1.npz"), **{k: v.numpy() for k, v in out.items()},
             **{"noaudio_" + k: v.numpy() for k, v in out_na.items() if k.startswith("rec")})

    # 2. end-to-end clips
    golden_clips(model, vq, ((128, 2), (70, 1), (129, 1), (40, 1), (64, 1)))

    # 3. VQ-VAE / AE stacks at two depths (vae_layer is a checkpoint parameter, SURVEY §8a note)
    for layer in (2, 3):
        _, vqc_l, gc_l = common.cfg_dicts(vae_layer=layer, global_layer=4 if layer == 2 else 3)
        _, vq_l = rh.build_reference(acfg, vqc_l, gc_l, seed=0)
        g = torch.Generator().manual_seed(11)
        rec = {}
        with torch.no_grad():
            for p in common.PARTS:
                m = getattr(vq_l, f"vq_model_{p}")
                dim = vqc_l[p]["vae_test_dim"]
                x = torch.randn(2, 40, dim, generator=g)
                idx = torch.randint(0, 256, (2, 40), generator=g)
                z = torch.randn(2, 40, 256, generator=g)
                rec[f"{p}_map2index"] = m.map2index(x).numpy()
                rec[f"{p}_pre_latent"] = m.encoder(x).numpy()
                rec[f"{p}_decode"] = m.decode(idx).numpy()
                rec[f"{p}_nearest"] = m.quantizer.map2index(z).numpy()
                rec[f"{p}_decode_from_latent"] = m.decode_from_latent(z).numpy()
            x = torch.randn(2, 40, 61, generator=g)
            rec["global_rec_pose"] = vq_l.global_motion(x)["rec_pose"].numpy()
            rec["global_trans"] = vq_l.get_global_motion(x, torch.zeros(1, 3)).numpy()
        np.savez(os.path.join(HERE, f"vq_layer{layer}.npz"), **rec)

    # 4. rotation helpers and the decode merge
    ref = rh.import_reference()
    from models.emage_audio import processing_emage_audio as P
    g = torch.Generator().manual_seed(5)
    d6 = torch.randn(4, 50, 6, generator=g)
    aa = torch.randn(4, 50, 3, generator=g) * torch.tensor([1.0, 0.3, 1e-4, 0.0]).view(4, 1, 1)
    np.savez(os.path.join(HERE, "rotations.npz"), rot6d_to_aa=P.rotation_6d_to_axis_angle(d6).numpy(),
             aa_to_rot6d=P.axis_angle_to_rotation_6d(aa).numpy())
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()

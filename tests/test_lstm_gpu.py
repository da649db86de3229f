"""DisCo / CaMN on the MI355X (BASELINE configs[3] / [4]): the LSTM-specific kernels against their CPU
restatements, both models against the REFERENCE's golden outputs (tests/golden/lstm_models.npz, generated from the real
modules by tests/golden/make_golden_lstm.py) and the oracle, and size-independent properties at the BASELINE batch sizes."""
import os

import numpy as np
import pytest
import torch

import fake_ops as F
from oracle import lstm_models_oracle as lo
from pantomatrix_amd import ops
from pantomatrix_amd._lib import F32, F16X3
from test_lstm_host_logic import product
from test_lstm_models_oracle import CFG, inputs, weights, run_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype", [F32, F16X3], ids=["fp32", "f16x3"])
@pytest.mark.parametrize("b", [2, 130, 256])
def test_lstm_step_kernel(dtype, b):
    """Two consecutive steps of one direction as strided views of (B, T, .) tensors, against the CPU restatement."""
    g = torch.Generator().manual_seed(b)
    hid, t = 512, 3
    w = torch.randn(4 * hid, hid, generator=g) / hid ** 0.5
    wp, ws = (ops.split_f16_weights(w) if dtype == F16X3 else (w, 1.0))
    gx = torch.randn(b, t, 8 * hid, generator=g)
    c0 = 0.5 * torch.randn(b, hid, generator=g)
    h0 = torch.tanh(torch.randn(b, hid, generator=g))

    def run(mod, dev):
        gxd, c, hseq, h_init = gx.to(dev), c0.clone().to(dev), torch.zeros(b, t, 2 * hid, device=dev), h0.to(dev)
        prev = h_init
        for s in (1, 2):                                                # direction 1 of the layout: columns [H, 2H), gates [4H, 8H)
            cur = hseq[:, s, hid:]
            mod.lstm_step(dtype, prev, wp.to(dev), gxd[:, s, 4 * hid:], c, cur, w_scale=ws)
            prev = cur
        return hseq.cpu(), c.cpu()

    got_h, got_c = run(ops, DEV)
    ref_h, ref_c = run(F, "cpu")
    assert float((got_h - ref_h).abs().max()) < 2e-5 and float((got_c - ref_c).abs().max()) < 2e-5
    assert float(got_h[:, 0].abs().max()) == 0.0 and float(got_h[:, :, :hid].abs().max()) == 0.0      # nothing written elsewhere


@pytest.mark.parametrize("dtype", [F32, F16X3], ids=["fp32", "f16x3"])
def test_lstm_step_pair_kernel(dtype):
    """Both directions in one launch: forward direction at steps 0, 1 beside the backward one at steps T-1, T-2 — the same
    results as two independent emage_lstm_step launches (and as the CPU restatement)."""
    g = torch.Generator().manual_seed(5)
    b, hid, t = 130, 512, 4
    ws, wp = [], []
    for _ in range(2):
        w = torch.randn(4 * hid, hid, generator=g) / hid ** 0.5
        p, s = (ops.split_f16_weights(w) if dtype == F16X3 else (w, 1.0))
        wp.append(p)
        ws.append(s)
    gx = torch.randn(b, t, 8 * hid, generator=g)

    def run(mod, dev, paired):
        gxd, c, hseq = gx.to(dev), torch.zeros(2, b, hid, device=dev), torch.zeros(b, t, 2 * hid, device=dev)
        prev = [torch.zeros(b, hid, device=dev), torch.zeros(b, hid, device=dev)]
        for s in range(2):
            sf, sb = s, t - 1 - s
            cur = [hseq[:, sf, :hid], hseq[:, sb, hid:]]
            fwd = (prev[0], wp[0].to(dev), gxd[:, sf, :4 * hid], c[0], cur[0], ws[0])
            bwd = (prev[1], wp[1].to(dev), gxd[:, sb, 4 * hid:], c[1], cur[1], ws[1])
            if paired:
                mod.lstm_step_pair(dtype, fwd, bwd)
            else:
                for h_prev, w, gt, cs, ho, sc in (fwd, bwd):
                    mod.lstm_step(dtype, h_prev, w, gt, cs, ho, w_scale=sc)
            prev = cur
        return hseq.cpu(), c.cpu()

    pair_h, pair_c = run(ops, DEV, True)
    single_h, single_c = run(ops, DEV, False)
    ref_h, ref_c = run(F, "cpu", True)
    assert torch.equal(pair_h, single_h) and torch.equal(pair_c, single_c)
    assert float((pair_h - ref_h).abs().max()) < 2e-5 and float((pair_c - ref_c).abs().max()) < 2e-5


@pytest.mark.parametrize("b,t,hid", [(2, 5, 512), (130, 4, 512), (256, 6, 512), (300, 3, 512), (70, 5, 256)])
def test_lstm_layer_kernel(b, t, hid):
    """The persistent recurrence (one launch per layer, W_hh and the cell state in registers, group barriers through
    agent-scope atomics) against T paired step launches from a zero state: the same bits, incl. ragged 64-clip slices and a
    batch that needs two launches; and against the CPU restatement."""
    g = torch.Generator().manual_seed(b + t)
    wp, ws, wraw = [], [], []
    for _ in range(2):
        w = torch.randn(4 * hid, hid, generator=g) / hid ** 0.5
        p, s = ops.split_f16_weights(w)
        wp.append(p.to(DEV))
        ws.append(s)
    gx = torch.randn(b, t, 8 * hid, generator=g)
    gxd = gx.to(DEV)

    hseq = torch.full((b, t, 2 * hid), float("nan"), device=DEV)
    sync = ops.lstm_layer_sync(b, hid, DEV)
    ops.lstm_layer(F16X3, gxd, wp, ws, hseq, sync)
    torch.cuda.synchronize()
    ops.lstm_layer_check(sync)

    def steps(mod, dev):
        gxx = gx.to(dev)
        c, ref = torch.zeros(2, b, hid, device=dev), torch.zeros(b, t, 2 * hid, device=dev)
        prev = [torch.zeros(b, hid, device=dev), torch.zeros(b, hid, device=dev)]
        for s in range(t):
            sf, sb = s, t - 1 - s
            cur = [ref[:, sf, :hid], ref[:, sb, hid:]]
            mod.lstm_step_pair(F16X3, (prev[0], wp[0].to(dev), gxx[:, sf, :4 * hid], c[0], cur[0], ws[0]),
                               (prev[1], wp[1].to(dev), gxx[:, sb, 4 * hid:], c[1], cur[1], ws[1]))
            prev = cur
        return ref.cpu()

    got = hseq.cpu()
    assert torch.equal(got, steps(ops, DEV))
    if b <= 130:
        assert float((got - steps(F, "cpu")).abs().max()) < 2e-5


def test_lstm_layer_long_sequence_and_replay():
    """415 steps (the CaMN length) at the full 256-clip batch, twice on the same buffers: every group barrier of a long
    launch holds, the scratch counters are re-armed by the call, results repeat bit for bit."""
    g = torch.Generator().manual_seed(3)
    b, t, hid = 256, 415, 512
    wp, ws = [], []
    for _ in range(2):
        p, s = ops.split_f16_weights(torch.randn(4 * hid, hid, generator=g) / hid ** 0.5)
        wp.append(p.to(DEV))
        ws.append(s)
    gx = torch.randn(b, t, 8 * hid, generator=g).to(DEV)
    sync = ops.lstm_layer_sync(b, hid, DEV)
    outs = []
    for _ in range(2):
        hseq = torch.empty(b, t, 2 * hid, device=DEV)
        ops.lstm_layer(F16X3, gx, wp, ws, hseq, sync)
        torch.cuda.synchronize()
        ops.lstm_layer_check(sync)
        outs.append(hseq)
    assert torch.equal(outs[0], outs[1]) and bool(torch.isfinite(outs[0]).all())
    # spot-check the last forward step and the last backward step of a few clips against per-step launches of those clips
    sel = torch.tensor([0, 63, 64, 200, 255], device=DEV)
    ref = torch.empty(len(sel), t, 2 * hid, device=DEV)
    c = torch.zeros(2, len(sel), hid, device=DEV)
    prev = [torch.zeros(len(sel), hid, device=DEV), torch.zeros(len(sel), hid, device=DEV)]
    gsel = gx[sel].contiguous()
    for s in range(t):
        sf, sb = s, t - 1 - s
        cur = [ref[:, sf, :hid], ref[:, sb, hid:]]
        ops.lstm_step_pair(F16X3, (prev[0], wp[0], gsel[:, sf, :4 * hid], c[0], cur[0], ws[0]), (prev[1], wp[1], gsel[:, sb, 4 * hid:], c[1], cur[1], ws[1]))
        prev = cur
    assert torch.equal(outs[0][sel], ref)


def test_small_lstm_kernels():
    g = torch.Generator().manual_seed(4)
    m = 300
    sel, c1, c2 = torch.randn(m, 2, generator=g) * 3, torch.randn(m, 128, generator=g), torch.randn(m, 128, generator=g)
    buf = torch.zeros(m, 576)
    F.softmax2_mix(sel, c1, c2, buf[:, :128])
    dbuf = torch.zeros(m, 576, device=DEV)
    ops.softmax2_mix(sel.to(DEV), c1.to(DEV), c2.to(DEV), dbuf[:, :128])
    assert float((dbuf.cpu() - buf).abs().max()) < 2e-6
    # input tail: speaker | seed | flag | zeros, written into a column block of the LSTM input rows
    b, t, pd = 3, 20, 258
    table, sid = torch.randn(4, 16, generator=g), torch.tensor([2, 0, 3])
    seed = torch.randn(b, 11, pd, generator=g)
    src = torch.tensor(list(range(11)) + list(range(2, 11)), dtype=torch.int32)       # the shorter-seed map of D:238-242
    for sm in (seed, None):
        ref = torch.full((b * t, 576), 7.0)
        F.lstm_inputs(ref[:, 256:], table, sid, sm, pd, 4, src, b, t)
        got = torch.full((b * t, 576), 7.0, device=DEV)
        ops.lstm_inputs(got[:, 256:], table.to(DEV), sid.to(DEV), None if sm is None else sm.to(DEV), pd, 4, src.to(DEV), b, t)
        assert torch.equal(got.cpu(), ref)
    # rot-6D -> axis-angle scattered to the 55 joints
    from pantomatrix_amd import spec
    slot = torch.full((55,), -1, dtype=torch.int32)
    for i, j in enumerate(spec.LOCAL_UPPER_JOINTS):
        slot[j] = i
    r6 = torch.randn(m, 258, generator=g)
    got = ops.rot6d_scatter(r6.to(DEV), slot.to(DEV)).cpu()
    ref = F.rot6d_scatter(r6, slot)
    assert float((got - ref).abs().max()) < 1e-3 and float(got.view(m, 55, 3)[:, 0].abs().max()) == 0.0


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
@pytest.mark.parametrize("kind", ["disco", "camn"])
def test_models_match_reference_golden(golden_dir, kind, precision):
    g = np.load(os.path.join(golden_dir, "lstm_models.npz"))
    model = product(kind, precision, DEV)
    for tag, wsm in (("plain", False), ("seeded", True)):
        audio, spk, motion = inputs(with_seed_motion=wsm)
        out = model(audio.to(DEV), spk.to(DEV), seed_frames=CFG["seed_frames"], seed_motion=None if motion is None else motion.to(DEV))
        err_m = float(np.abs(out["motion"].reshape(2, -1, 258).cpu().numpy() - g[f"{kind}_{tag}_motion"]).max())
        err_a = float(np.abs(out["motion_axis_angle"].cpu().numpy() - g[f"{kind}_{tag}_axis_angle"]).max())
        print(f"{kind} {precision} {tag}: rot-6D max|err| {err_m:.2e}, axis-angle max|err| {err_a:.2e} vs the reference")
        assert err_m < 2e-4 and err_a < 1e-3          # north_star: 1e-3 on rotation parameters
        if kind == "disco":
            ref = run_oracle(kind, weights(kind), audio, spk, motion)
            for k in ("audio_fea_c", "audio_fea_r"):
                assert float((out[k].cpu() - ref[k]).abs().max()) < 1e-4, k


@pytest.mark.parametrize("kind,batch,seconds", [("disco", 128, 8.5), ("camn", 256, 28.0)])
def test_baseline_batch_properties(kind, batch, seconds):
    """BASELINE configs[3] (DisCo, batch 128) and configs[4] (CaMN, batch 256 long clips): finite, deterministic, and each
    clip independent of its batch-mates (clips 0-1 against the same clips run as a batch of 2 and against the oracle)."""
    from pantomatrix_amd import synthetic
    model = product(kind, "f16x3", DEV)
    n = int(seconds * 16000)
    audio = synthetic.synthetic_audio(batch, n, seed=77)
    spk = torch.zeros(batch, 1, dtype=torch.long)
    out = model(audio.to(DEV), spk.to(DEV))["motion"].reshape(batch, -1, 258).cpu()
    again = model(audio.to(DEV), spk.to(DEV))["motion"].reshape(batch, -1, 258).cpu()
    small = model(audio[:2].to(DEV), spk[:2].to(DEV))["motion"].reshape(2, -1, 258).cpu()
    t = out.shape[1]
    assert abs(t - seconds * 15) <= 0.02 * seconds * 15 + 1          # the WavEncoder loses a few frames at the borders (28 s -> 415)
    assert torch.isfinite(out).all() and torch.equal(out, again)
    assert float((out[:2] - small).abs().max()) < 1e-4
    ref = run_oracle(kind, weights(kind), audio[:2], spk[:2], None)["motion"].reshape(2, -1, 258)
    err = float((out[:2] - ref).abs().max())
    print(f"{kind} B={batch} T={t}: clips 0-1 vs the CPU oracle max|err| {err:.2e}")
    assert err < 5e-4


def test_graph_runner_matches_eager():
    from pantomatrix_amd.runtime import LstmClipRunner
    model = product("camn", "f16x3", DEV)
    audio, spk, _ = inputs(bs=4, frames=40)
    eager = model(audio.to(DEV), spk.to(DEV))
    runner = LstmClipRunner(model, 4, audio.shape[1])
    for _ in range(2):
        motion, aa = runner(audio.to(DEV))
        assert np.array_equal(motion, eager["motion"].reshape(4, -1, 258).cpu().numpy())
        assert np.array_equal(aa, eager["motion_axis_angle"].cpu().numpy())


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
def test_pair_rows_route_matches_padded_route(precision):
    """The narrow WavEncoder blocks on (L/2, 64) pair rows (two first-layer launches, 64-channel slab / implicit-GEMM
    kernels with re-packed weights) against the zero-padded route and the CPU oracle, on a clip whose frame counts are even."""
    audio, spk, _ = inputs(bs=3, frames=34)
    audio = audio[:, :30000]
    outs = []
    for pair in (True, False):
        model = product("disco", precision, DEV)
        model.pair_convs = pair
        assert model._wav_pairs_ok(model._wav_lengths(audio.shape[1]))
        o = model(audio.to(DEV), spk.to(DEV))
        outs.append({k: o[k].cpu() for k in ("motion", "audio_fea_c", "audio_fea_r")})
    for k in outs[0]:
        assert float((outs[0][k] - outs[1][k]).abs().max()) < 2e-5, k
    ref = run_oracle("disco", weights("disco"), audio, spk, None)
    for k in ("audio_fea_c", "audio_fea_r"):
        assert float((outs[0][k] - ref[k]).abs().max()) < 2e-5, k
    assert float((outs[0]["motion"].reshape(3, -1, 258) - ref["motion"].reshape(3, -1, 258)).abs().max()) < 5e-5


@pytest.mark.parametrize("precision", ["f16x3", "fp32"])
@pytest.mark.parametrize("kind", ["disco", "camn"])
def test_pair_rows_route_matches_reference_golden_on_device(golden_dir, kind, precision):
    """The pair-rows WavEncoder route (what DisCo / CaMN take at the BASELINE clip lengths) on the MI355X against the REAL
    reference's golden (tests/golden/lstm_models_even.npz: even frame counts behind blocks 0-2, seed motion longer than the audio)."""
    g = np.load(os.path.join(golden_dir, "lstm_models_even.npz"))
    audio, spk, motion = inputs(with_seed_motion=True)
    audio = audio[:, :30000]
    model = product(kind, precision, DEV)
    assert model.pair_convs and model._wav_pairs_ok(model._wav_lengths(audio.shape[1]))
    out = model(audio.to(DEV), spk.to(DEV), seed_frames=CFG["seed_frames"], seed_motion=motion.to(DEV))
    err_m = float(np.abs(out["motion"].reshape(2, -1, 258).cpu().numpy() - g[f"{kind}_motion"]).max())
    err_a = float(np.abs(out["motion_axis_angle"].cpu().numpy() - g[f"{kind}_axis_angle"]).max())
    print(f"{kind} {precision} pair-rows route: rot-6D max|err| {err_m:.2e}, axis-angle max|err| {err_a:.2e} vs the reference")
    assert err_m < 2e-4 and err_a < 1e-3


def test_lost_block_is_reported_on_every_replay():
    """The persistent recurrence's error words travel in the runner's in-graph health counter: a launch record whose error word
    is set makes EVERY later replay raise (not only the first two)."""
    from pantomatrix_amd._lib import EmageKernelError
    from pantomatrix_amd.runtime import LstmClipRunner
    assert ops.lstm_layer_supported(F16X3, 512) and ops.lstm_layer_supported(F16X3, 256) and not ops.lstm_layer_supported(F32, 512)
    sync = ops.lstm_layer_sync(300, 512, DEV)                  # 300 clips: two launch records
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.lstm_layer_health(sync, counter)
    assert int(counter) == 0
    sync.view(-1, ops.LSTM_SYNC_WORDS_PER_LAUNCH)[1, ops.LSTM_SYNC_ERROR_WORD] = 1
    ops.lstm_layer_health(sync, counter)
    assert int(counter) == 1
    model = product("disco", "f16x3", DEV)
    audio, spk, _ = inputs(bs=2, frames=34)
    runner = LstmClipRunner(model, 2, audio.shape[1])
    for _ in range(4):
        runner(audio.to(DEV))                                   # healthy replays pass
    # the fold is part of `_step` (what the graph captures): poison an error word that the layer's own memset does not clear (a spare
    # record behind its launches) and every later call raises — not only the first two
    eager = LstmClipRunner(model, 2, audio.shape[1], use_graph=False)
    eager(audio.to(DEV))
    key = next(iter(model._sync))
    victim = model._sync[key]
    extra = torch.zeros(victim.numel() + ops.LSTM_SYNC_WORDS_PER_LAUNCH, dtype=torch.int32, device=DEV)
    extra[-ops.LSTM_SYNC_WORDS_PER_LAUNCH + ops.LSTM_SYNC_ERROR_WORD] = 1
    model._sync[key] = extra
    try:
        for _ in range(3):
            with pytest.raises(EmageKernelError):
                eager(audio.to(DEV))
    finally:
        model._sync[key] = victim
    eager(audio.to(DEV))

"""BEAT2 `.npz` motion files and 16 kHz audio input — the on-disk formats either side of the EMAGE hot path
(SURVEY.md §8f row 2).  Host-side numpy code: file formats are not GPU work.

* ``beat_format_save / beat_format_load / select_with_mask / recover_from_mask / time_upsample_numpy`` follow
  /root/reference/emage_utils/motion_io.py:16-44,69-179 (same keys, dtypes and array layouts; written with joint
  index arrays instead of boolean-mask reshapes).
* ``load_audio`` replaces the ``librosa.load(path, sr=16000)`` call of /root/reference/test_emage_audio.py:17 for
  RIFF/WAVE input without third-party decoders.
"""
from __future__ import annotations

import struct

import numpy as np

# joint masks of emage_utils/motion_io.py:5-15
MASK_DICT = {
    "local_upper": [j in (3, 6, 9, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21) or j >= 25 for j in range(55)],
    "local_full": [j != 0 for j in range(55)],
}


def _joint_ids(mask):
    return np.flatnonzero(np.asarray(mask, dtype=bool))


def select_with_mask(motion: np.ndarray, mask) -> np.ndarray:
    """Keep the channels of the joints flagged in `mask`: (..., J*c) -> (..., sum(mask)*c)  (motion_io.py:17-26)."""
    j = len(mask)
    c = motion.shape[-1] // j
    ids = _joint_ids(mask)
    return motion.reshape(motion.shape[:-1] + (j, c))[..., ids, :].reshape(motion.shape[:-1] + (len(ids) * c,))


def recover_from_mask(selected: np.ndarray, mask) -> np.ndarray:
    """Inverse of select_with_mask with zeros for the dropped joints  (motion_io.py:28-41)."""
    j = len(mask)
    ids = _joint_ids(mask)
    c = selected.shape[-1] // len(ids)
    out = np.zeros(selected.shape[:-1] + (j, c), dtype=selected.dtype)
    out[..., ids, :] = selected.reshape(selected.shape[:-1] + (len(ids), c))
    return out.reshape(selected.shape[:-1] + (j * c,))


def time_upsample_numpy(data: np.ndarray, k: int) -> np.ndarray:
    """Linear interpolation along time, (..., t, c) -> (..., k*t, c), sample positions linspace(0, t-1, k*t)
    (motion_io.py:69-101)."""
    if k == 1:
        return data.copy()
    t = data.shape[-2]
    pos = np.linspace(0, t - 1, k * t)
    lo = np.clip(np.searchsorted(np.arange(t), pos, side="right") - 1, 0, t - 2)
    w = (pos - lo)[:, None]
    a, b = np.take(data, lo, axis=-2), np.take(data, lo + 1, axis=-2)
    return a + (b - a) * w


def beat_format_save(save_path, motion_data, mask=None, betas=None, expressions=None, trans=None, upsample=None):
    """Write a BEAT2 / SMPL-X `.npz` (motion_io.py:103-163): betas (300,), poses (T,165), expressions (T,100),
    trans (T,3), model='smplx2020', gender='neutral', mocap_frame_rate=30."""
    n = motion_data.shape[0]
    if betas is None:
        betas = np.zeros((n, 300), dtype=motion_data.dtype)
    if expressions is None:
        expressions = np.zeros((n, 100), dtype=motion_data.dtype)
    if trans is None:
        # the reference derives a default root translation from the licensed SMPL-X body model (motion_io.py:116-140)
        raise NotImplementedError("trans=None needs the SMPL-X body model assets; pass the predicted translation")
    if mask is not None:
        motion_data = recover_from_mask(motion_data, mask)
    if upsample is not None and upsample > 1:
        motion_data, betas = time_upsample_numpy(motion_data, upsample), time_upsample_numpy(betas, upsample)
        expressions, trans = time_upsample_numpy(expressions, upsample), time_upsample_numpy(trans, upsample)
    np.savez(save_path, betas=betas[0], poses=motion_data, expressions=expressions, trans=trans,
             model="smplx2020", gender="neutral", mocap_frame_rate=30)


def beat_format_load(load_path, mask=None):
    """motion_io.py:165-179."""
    data = np.load(load_path, allow_pickle=True)
    poses = data["poses"]
    if mask is not None:
        poses = select_with_mask(poses, mask)
    return {"poses": poses, "betas": data["betas"], "expressions": data["expressions"], "trans": data["trans"]}


# --------------------------------------------------------------------------------------------------------------
def _read_wav(path):
    """Minimal RIFF/WAVE reader: PCM 8/16/24/32-bit and IEEE float 32/64, any channel count -> (float32 (n, ch), sr)."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:4] != b"RIFF" or raw[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file (the reference's bundled example is an MP3 named .wav; decode it "
                         "to PCM first)")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(raw):
        cid, size = raw[pos:pos + 4], struct.unpack("<I", raw[pos + 4:pos + 8])[0]
        body = raw[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE and len(body) >= 26:            # WAVE_FORMAT_EXTENSIBLE: real tag in the GUID
                fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, ch, sr, _, _, bits = fmt
    if tag == 1:
        if bits == 8:
            x = (np.frombuffer(data, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(data, dtype="<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(data[:len(data) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            x = ((v ^ 0x800000) - 0x800000).astype(np.float32) / 8388608.0
        elif bits == 32:
            x = np.frombuffer(data, dtype="<i4").astype(np.float32) / 2147483648.0
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif tag == 3:
        x = np.frombuffer(data, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag}")
    n = len(x) // ch
    return x[:n * ch].reshape(n, ch), sr


def load_audio(path, sr=16000):
    """Mono float32 waveform at `sr` Hz, like `librosa.load(path, sr=sr)` (test_emage_audio.py:17): channels are
    averaged, then a polyphase low-pass resampler converts the rate (librosa's default is soxr; values agree to
    resampler tolerance, not bit-for-bit).  Returns (audio (n,), sr)."""
    x, in_sr = _read_wav(path)
    x = x.mean(axis=1) if x.shape[1] > 1 else x[:, 0]
    if in_sr != sr:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(in_sr), int(sr))
        x = resample_poly(x.astype(np.float64), sr // g, in_sr // g).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32), sr

"""Deterministic synthetic weights and inputs for the EMAGE hot path.

The HuggingFace checkpoints are unreachable offline, so parity tests,
golden fixtures and ``bench.py`` all use weights drawn here.  Every tensor is drawn from
its own generator seeded by (seed, crc32(name)), so the result does not depend on module
construction order, on torch's default RNG state, or on which other tensors exist — the
same name and shape always get the same values, here and on the GPU box.

Scales are chosen so the network is numerically "alive" (SURVEY.md §7 step 1):
variance-preserving weights, BatchNorm running statistics away from (0, 1), LayerNorm /
BatchNorm affine terms away from (1, 0), and N(0,1) codebooks so nearest-code margins are
not degenerate (the reference's default codebook init is U(+-1/256),
/root/reference/models/emage_audio/processing_emage_audio.py:142).
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict

import torch

from . import spec as _spec


def _gen(seed: int, name: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1_000_003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
    return g


def periodic_positional_encoding(period: int, d_model: int, max_seq_len: int) -> torch.Tensor:
    """The `pe` buffer of PeriodicPositionalEncoding
    (/root/reference/models/emage_audio/processing_emage_audio.py:332-340)."""
    position = torch.arange(0, period, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe = torch.zeros(period, d_model)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0).repeat(1, (max_seq_len // period) + 1, 1)


def draw(name: str, shape, role: str, seed: int, cfg=None) -> torch.Tensor:
    g = _gen(seed, name)
    n = lambda *s: torch.randn(*s, generator=g, dtype=torch.float32)
    if role == "conv_w":
        fan_in = shape[1] * shape[2]
        return n(*shape) * (1.0 / math.sqrt(fan_in))
    if role == "linear_w":
        return n(*shape) * (1.0 / math.sqrt(shape[1]))
    if role == "bias":
        return n(*shape) * 0.05
    if role == "norm_w":
        return 1.0 + 0.1 * n(*shape)
    if role == "norm_b":
        return 0.1 * n(*shape)
    if role == "bn_mean":
        return 0.1 * n(*shape)
    if role == "bn_var":
        return 0.5 + torch.rand(*shape, generator=g, dtype=torch.float32)
    if role == "bn_count":
        return torch.tensor(1000, dtype=torch.int64)
    if role == "embedding":
        return n(*shape) * 0.5
    if role == "mask_emb":
        return n(*shape) * 0.5
    if role == "codebook":
        return n(*shape)
    if role == "ppe":
        return periodic_positional_encoding(cfg.pose_length, shape[2], cfg.pose_length)
    raise ValueError(f"unknown parameter role {role!r} for {name}")


def state_dict_from_spec(spec, seed: int, cfg=None, prefix: str = "") -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k, draw(prefix + k, shape, role, seed, cfg)) for k, (shape, role) in spec.items())


def audio_model_state(cfg, seed: int = 0):
    return state_dict_from_spec(_spec.audio_model_spec(cfg), seed, cfg, prefix="emage_audio/")


def vqvae_state(cfg, part: str, seed: int = 0):
    return state_dict_from_spec(_spec.vqvae_spec(cfg), seed, cfg, prefix=f"emage_vq/{part}/")


def vae_state(cfg, seed: int = 0):
    return state_dict_from_spec(_spec.vae_spec(cfg), seed, cfg, prefix="emage_vq/global/")


def synthetic_audio(batch: int, n_samples: int, seed: int = 1234) -> torch.Tensor:
    """0.1*N(0,1) 16 kHz audio, the BASELINE.md §3 input (seed 1234).  Each clip has its own generator
    (seed, clip index), so clip i is the same waveform whatever the batch size."""
    rows = []
    for i in range(batch):
        g = torch.Generator(device="cpu")
        g.manual_seed(seed * 1_000_003 + i)
        rows.append(0.1 * torch.randn(n_samples, generator=g, dtype=torch.float32))
    return torch.stack(rows)


def samples_for_frames(frames: int, sr: int = 16000, fps: int = 30) -> int:
    """Smallest L with L*fps//sr == frames (68 267 for 128 frames)."""
    return -(-frames * sr // fps)

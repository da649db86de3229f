"""Shared pieces of the parity tests: the model builders of tools/workloads.py (synthetic configs, oracle models, product models) and
seeded inputs."""
from __future__ import annotations

import torch

from pantomatrix_amd import spec, synthetic
from pantomatrix_amd.configuration_emage_audio import EmageAudioConfig, EmageVQVAEConvConfig, EmageVAEConvConfig

from tools.workloads import PARTS, cfg_dicts, oracle_models, product_models, product_infer_clip, train_batch  # noqa: F401  (shared with bench.py / smoke(): tools/workloads.py)


def window_inputs(batch, frames=64, seed=7):
    """A forward() window: audio, speaker ids, a partly-masked motion window."""
    from oracle import emage_oracle as orc
    g = torch.Generator().manual_seed(seed)
    audio = 0.1 * torch.randn(batch, frames * 533, generator=g)
    aa = 0.3 * torch.randn(batch, frames, 55, 3, generator=g)
    motion = torch.cat([orc.axis_angle_to_rotation_6d(aa).reshape(batch, frames, 330),
                        0.1 * torch.randn(batch, frames, 7, generator=g)], dim=-1)
    mask = torch.ones(batch, frames, 337)
    mask[:, :4] = 0
    mask[:, 20:24, :100] = 0
    return audio, torch.zeros(batch, 1, dtype=torch.long), motion, mask


def vq_api_inputs(batch=3, frames=24, seed=41):
    """Seeded inputs of EmageVQModel.map2index / map2latent: (rot6d (B,T,330), expression (B,T,100), contact (B,T,4) in
    {0,1}, trans (B,T,3))."""
    from oracle import emage_oracle as orc
    g = torch.Generator().manual_seed(seed)
    aa = 0.4 * torch.randn(batch, frames, 55, 3, generator=g)
    rot6d = orc.axis_angle_to_rotation_6d(aa).reshape(batch, frames, 330)
    expr = 0.5 * torch.randn(batch, frames, 100, generator=g)
    contact = (torch.rand(batch, frames, 4, generator=g) > 0.5).float()
    trans = 0.1 * torch.randn(batch, frames, 3, generator=g)
    return rot6d, expr, contact, trans

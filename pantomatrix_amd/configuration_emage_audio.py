"""Configuration objects of the EMAGE hot path.

Mirrors the three config classes of
/root/reference/models/emage_audio/configuration_emage_audio.py:4-32: each one is an
attribute bag that splats a dict (or an OmegaConf node, when omegaconf is installed)
into attributes.  Field names are kept verbatim because ``inference()`` semantics
depend on them (``cfg.pose_length``, ``cfg.seed_frames``, ``cfg.lf / cf ...``,
modeling_emage_audio.py:365-366,403-410).  They read / write the HuggingFace
``config.json`` format without depending on ``transformers``.
"""
from __future__ import annotations

import json
import os


class _AttrConfig:
    model_type = ""

    def __init__(self, config_obj=None, **kwargs):
        if config_obj is not None:
            try:  # OmegaConf node -> plain container, as the reference does
                from omegaconf import OmegaConf  # type: ignore
                kwargs.update(OmegaConf.to_container(config_obj, resolve=True))
            except ImportError:
                kwargs.update(dict(config_obj))
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to_dict(self):
        d = {k: v for k, v in self.__dict__.items() if not k.startswith("_")}
        d["model_type"] = self.model_type
        return d

    def save_pretrained(self, save_directory):
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)

    @classmethod
    def from_pretrained(cls, directory):
        with open(os.path.join(directory, "config.json")) as f:
            d = json.load(f)
        d.pop("model_type", None)
        return cls(**d)

    def __repr__(self):
        return f"{type(self).__name__}({self.to_dict()})"


class EmageAudioConfig(_AttrConfig):
    model_type = "emage_audio"


class EmageVQVAEConvConfig(_AttrConfig):
    model_type = "emage_vqvaeconv"


class EmageVAEConvConfig(_AttrConfig):
    model_type = "emage_vaeconv"

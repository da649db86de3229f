"""pantomatrix_amd — the EMAGE speech-to-gesture hot path of PantoMatrix, built MI355X-first.

Exports the same names as /root/reference/models/emage_audio/__init__.py:4-12.  Importing the
package does not need a GPU; running any model does (no CPU fallback).
"""
from .configuration_emage_audio import EmageAudioConfig, EmageVQVAEConvConfig, EmageVAEConvConfig
from .modeling_emage_audio import EmageAudioModel, EmageVQVAEConv, EmageVQModel, EmageVAEConv

__all__ = [
    "EmageAudioConfig",
    "EmageAudioModel",
    "EmageVQVAEConvConfig",
    "EmageVQVAEConv",
    "EmageVQModel",
    "EmageVAEConvConfig",
    "EmageVAEConv",
]

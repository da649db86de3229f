"""pantomatrix_amd — the EMAGE speech-to-gesture hot path of PantoMatrix (and its two LSTM siblings), built MI355X-first.

Exports the same names as /root/reference/models/emage_audio/__init__.py:4-12, plus the model / config classes of
models/disco_audio/__init__.py and models/camn_audio/__init__.py.  Importing the
package does not need a GPU; running any model does (no CPU fallback).
"""
from .configuration_emage_audio import EmageAudioConfig, EmageVQVAEConvConfig, EmageVAEConvConfig
from .modeling_emage_audio import EmageAudioModel, EmageVQVAEConv, EmageVQModel, EmageVAEConv
from .modeling_lstm_audio import DiscoAudioConfig, DiscoAudioModel, CamnAudioConfig, CamnAudioModel   # models/disco_audio, models/camn_audio

__all__ = [
    "EmageAudioConfig",
    "EmageAudioModel",
    "EmageVQVAEConvConfig",
    "EmageVQVAEConv",
    "EmageVQModel",
    "EmageVAEConvConfig",
    "EmageVAEConv",
    "DiscoAudioConfig",
    "DiscoAudioModel",
    "CamnAudioConfig",
    "CamnAudioModel",
]

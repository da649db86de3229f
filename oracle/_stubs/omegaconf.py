"""Minimal stand-in for the `omegaconf` package (not installed in this image).

The reference's configuration module imports `OmegaConf` at module scope
(/root/reference/models/emage_audio/configuration_emage_audio.py:2) but only calls
`OmegaConf.to_container` when a `config_obj` is passed.  Test infrastructure only.
"""


class OmegaConf:
    @staticmethod
    def to_container(obj, resolve=True):
        return dict(obj)

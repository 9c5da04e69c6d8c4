"""`audiocodec.model.AudioCodec` — the import the reference's callers use (inference.py:7).
Resolves to the MI355X implementation in simwhisper_codec_amd.codec."""
from simwhisper_codec_amd.codec import AudioCodec  # noqa: F401

__all__ = ["AudioCodec"]

"""Drop-in import path of the reference (`from audiocodec.model import AudioCodec`, inference.py:7)."""

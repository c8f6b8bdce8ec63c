"""TEST-ONLY stand-in for `librosa` (only `librosa.core.load`, `inference.py:45`), backed by efficientat_amd.audio_io."""
from . import core  # noqa: F401

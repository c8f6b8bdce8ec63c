"""Stand-in for torchaudio 0.13 (test infrastructure only, see ../README.md)."""
from . import compliance, transforms  # noqa: F401

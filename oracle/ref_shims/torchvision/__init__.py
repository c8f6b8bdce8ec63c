"""Stand-in for torchvision 0.14 (test infrastructure only, see ../README.md)."""
from . import ops  # noqa: F401

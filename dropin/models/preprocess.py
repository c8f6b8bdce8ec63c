"""`models.preprocess` of the reference, served by the fused HIP mel kernel."""
from efficientat_amd.preprocess import AugmentMelSTFT  # noqa: F401

"""efficientat_amd: MI355X-native hot path of EfficientAT (mel front-end + MN/DyMN).

Public mirrors of the reference API:
    efficientat_amd.preprocess.AugmentMelSTFT   <- models.preprocess.AugmentMelSTFT
    efficientat_amd.mn.get_model                <- models.mn.model.get_model
The arithmetic lives in libeat_hip.so (efficientat_amd/csrc, C ABI in include/eat_hip.h).
"""
__version__ = "0.1.0"

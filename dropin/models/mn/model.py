"""`models.mn.model` of the reference, served by the HIP launch plan."""
from efficientat_amd.mn import (MN, InvertedResidualConfig, get_model, mobilenet_v3, model_dir, model_url,  # noqa: F401
                                pretrained_models)

# `windowed_inference.py:8` imports get_ensemble_model from THIS module (the reference itself only defines it in
# models/ensemble.py, so that script fails on its own import there); exported here so the script runs
from models.ensemble import EnsemblerModel, get_ensemble_model  # noqa: E402,F401

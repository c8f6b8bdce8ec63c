"""`models.mn.model` of the reference, served by the HIP launch plan."""
from efficientat_amd.mn import (MN, InvertedResidualConfig, get_model, mobilenet_v3, model_dir, model_url,  # noqa: F401
                                pretrained_models)

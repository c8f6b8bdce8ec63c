"""`models.dymn.model` of the reference, served by the HIP launch plan."""
from efficientat_amd.dymn import (DyMN, DynamicInvertedResidualConfig, dymn, get_model, model_dir, model_url,  # noqa: F401
                                  pretrained_models)

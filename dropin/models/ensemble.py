"""`models.ensemble` of the reference (models/ensemble.py): logit averaging over several networks."""
import torch.nn as nn

from efficientat_amd.dymn import get_model as get_dymn
from efficientat_amd.mn import get_model as get_mobilenet
from efficientat_amd.utils import NAME_TO_WIDTH


class EnsemblerModel(nn.Module):
    def __init__(self, models):
        super().__init__()
        self.models = nn.ModuleList(models)

    def forward(self, x):
        logits = sum(m(x)[0] for m in self.models) / len(self.models)
        return logits, logits


def get_ensemble_model(model_names):
    build = lambda n: (get_dymn if n.startswith("dymn") else get_mobilenet)(width_mult=NAME_TO_WIDTH(n), pretrained_name=n)
    return EnsemblerModel([build(n) for n in model_names])

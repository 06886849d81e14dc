"""Registered disparity predictors, K4b: same names, constructor arguments and forward contract as
architecture/modeling/prediction/soft_argmin.py:9-73 and argmin.py:9-60, on the HIP kernels."""
import torch.nn as nn

from . import functional as TF
from .registry import PREDICTION_REGISTRY, configurable


@PREDICTION_REGISTRY.register()
class SOFTARGMIN(nn.Module):
    @configurable
    def __init__(self, temperature: float = 1.0, normalize: bool = True):
        super().__init__()
        self.temperature, self.normalize = temperature, normalize

    @classmethod
    def from_config(cls, cfg):
        return {"temperature": cfg.MODEL.PREDICTION.get("TEMPERATURE", 1.0),
                "normalize": cfg.MODEL.PREDICTION.get("NORMALIZE", True)}

    def forward(self, cost_volume, disp_sample):
        return TF.soft_argmin(cost_volume, disp_sample, self.temperature, self.normalize)

    @property
    def name(self):
        return 'SoftArgmin'


@PREDICTION_REGISTRY.register()
class ARGMIN(nn.Module):
    @configurable
    def __init__(self, dim: int = 1):
        super().__init__()
        if dim != 1:
            raise ValueError("only dim=1 (the candidate axis) is supported")
        self.dim = dim

    @classmethod
    def from_config(cls, cfg):
        return {"dim": cfg.MODEL.PREDICTION.get("DIM", 1)}

    def forward(self, cost_volume, disp_sample):
        return TF.argmin_select(cost_volume, disp_sample)

    @property
    def name(self):
        return 'Argmin'

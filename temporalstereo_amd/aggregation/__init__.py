from .blocks import (ConvexUpsample, DepthwiseConv3D, DepthwiseConvTranspose3D, PredictionHeads,  # noqa: F401
                     PyramidFusion, ResidualBlock3D, UNet)
from .levels import CoarseAggregation, FineAggregation, PreciseAggregation, TEMPORALSTEREO  # noqa: F401
from ..registry import AGGREGATION_REGISTRY, build_aggregation  # noqa: F401

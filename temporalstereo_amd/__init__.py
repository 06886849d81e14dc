"""temporalstereo_amd -- MI355X-native (gfx950) cost-volume stereo hot path.

Drop-in for the hot path of youmi-zym/TemporalStereo (cost-volume build, temporal cost warp,
3-D aggregation pyramid, disparity regression): hand-written HIP kernels in libts_hip.so behind
the C ABI of include/ts_hip.h, wrapped here as torch.autograd.Function ops and nn.Modules that
mirror the reference's interfaces.  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"

from .functional import (cat_fms, dif_fms, inverse_warp_3d, correlation, correlation1d, block_cost, topk_softargmax, soft_argmin, argmin_select,  # noqa: F401
                         FunctionSoftsplat, project_to_3d)
from .registry import (AGGREGATION_REGISTRY, PREDICTION_REGISTRY, build_aggregation, build_prediction,  # noqa: F401
                       CfgView, register_into)
from .aggregation import (TEMPORALSTEREO, CoarseAggregation, FineAggregation, PreciseAggregation)  # noqa: F401
from .prediction import SOFTARGMIN, ARGMIN  # noqa: F401
from . import temporal  # noqa: F401

"""CPU oracle for the cost-volume stereo hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (plain torch fp32/fp64 ops + numpy, no HIP) of the
reference algorithm youmi-zym/TemporalStereo runs on the path named by BASELINE.json.  Every
function cites the reference file:line it follows.

Rules (enforced by tests/test_layout.py):
  * only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it;
  * the product package `temporalstereo_amd` never imports it and has no CPU fallback;
  * it is the checker, never the thing measured or shipped.

Pinning: the restatement is checked against outputs of the real reference, imported in the
build container from /root/reference by tools/gen_golden.py, whose inputs/outputs are
committed as tests/golden/*.npz (tests/test_oracle_golden.py).  The one op the reference
cannot run on CPU -- the cupy soft-splat (architecture/modeling/layers/softsplat.py:252,269) --
is "parity unpinned by the reference": it is pinned by analytic known-answer tests instead
(tests/test_oracle_splat.py), see DESIGN.md.
"""
from .cost_volume import (cost_volume_int, cost_volume_sampled, block_cost,  # noqa: F401
                          warp_candidates, groupwise_neg_sqdiff)
from .regress import topk_softargmax, soft_argmin, argmin_select  # noqa: F401
from .splat import splat_sum, softsplat  # noqa: F401
from .geometry import project_to_3d  # noqa: F401

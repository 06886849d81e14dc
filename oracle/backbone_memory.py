"""Oracle (test infrastructure): the per-frame feature-memory plumbing of the reference's backbone, SURVEY.md section 8(f)-4.

Restates the tensor plumbing of `_inverted_residual_forward` (architecture/modeling/backbone/TemporalStereo.py:183-197, :218) --
everything of that function that is not the residual block itself:
    mc = int(ic * memory_percent)                      :185-191
    input1, input2 = input[:, :mc], input[:, mc:]       :193
    memory = input1 when there is none                  :194-195
    x = cat([memory, input2], 1) -> the block           :197
    return ..., input1  (the next frame's memory)      :218
Pinned: tests/golden/feature_memory_*.npz are outputs of the reference's own function (its definition compiled from the reference
file by tools/gen_golden.py, the timm block replaced by identities so that `out - input` is exactly x).
"""
import torch


def exchange(inp, memory=None, memory_percent=-1.0):
    ic = inp.shape[1]
    if memory is not None:
        mc = memory.shape[1]
        assert mc == int(ic * memory_percent), "input shape: {}; memory shape: {}!".format(inp.shape, memory.shape)
    else:
        mc = int(ic * memory_percent)
    input1, input2 = inp[:, :mc], inp[:, mc:]
    if memory is None:
        memory = input1
    return torch.cat([memory, input2], dim=1), input1

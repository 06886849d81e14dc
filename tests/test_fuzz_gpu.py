"""GPU: a seeded random-shape sweep of every autograd wrapper against float64 torch / the oracle (tests/fuzz_ops.py) -- odd sizes,
one-pixel maps, channel counts that are multiples of nothing.  Round 5's three wrong-result / refusal bugs on shapes no BASELINE
geometry has (K1's ragged-width LDS sizing, the 4x4 deconvolution with <= 8 output channels, its backward on odd widths) were found
by this sweep and the unit test that led to it; the regressions are pinned below, the sweep keeps looking."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("op", ["conv3d", "deconv2d", "block_cost", "dense", "pool_resize", "regress", "upsample", "topk", "correlation",
                                "sort_gather", "conv_bn_act", "splat", "losses", "glue"])
def test_random_shapes(op):
    import fuzz_ops
    found = fuzz_ops.sweep(op, 30, seed=0)
    assert not found, "\n".join(" ".join(map(str, f)) for f in found)


@pytest.mark.parametrize("family", ["hw", "d", "deconv"])
def test_random_shapes_inference_forms(family):
    """tests/fuzz_native.py: the inference forms over the C ABI with every kernel family forced where its `supported` query allows it
    (f32-input MFMA, bf16-split x6 incl. split-K, x6s stride 2 / transposed / 4x4 deconvolution, the (k,1,1) family), folded scale /
    shift, activation, per-plane addend."""
    import fuzz_native
    found = fuzz_native.sweep(family, 40, seed=0)
    assert not found, "\n".join(" ".join(map(str, f)) for f in found)


def test_random_geometries_end_to_end():
    """tests/fuzz_e2e.py: the launch-plan engine against the oracle aggregation, frame by frame, on planted scenes of random size
    (multiples of 16 only), candidate count, batch, sequence length and local-map size: |dEPE| < 1e-3 px per frame."""
    import fuzz_e2e
    log = []
    found = fuzz_e2e.sweep(8, seed=0, log=log)
    _dump("parity_random_geometries_inference.txt", log)
    assert not found, "\n".join(" ".join(map(str, f)) for f in found)
    assert len(log) >= 8


def test_random_geometries_train_mode_loss_and_all_gradients():
    """tests/fuzz_e2e.py (train): batch-statistics forward, the reference's objective and the backward of the WHOLE module path
    against the float64 oracle at random geometry: loss to 1e-4, the full parameter-gradient vector and every feature gradient to
    1e-3 in the relative L2 sense (measured 1e-6 ... 5e-5)."""
    import fuzz_e2e
    log = []
    found = fuzz_e2e.sweep_train(5, seed=0, log=log)
    _dump("parity_random_geometries_train.txt", log)
    assert not found, "\n".join(" ".join(map(str, f)) for f in found)


def _dump(name, lines):
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(root, exist_ok=True)
        with open(os.path.join(root, name), "w") as f:
            f.write("\n".join(lines) + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("shape", [(1, 13, 8, 23, 54), (2, 27, 2, 34, 49), (1, 2, 4, 35, 7), (1, 24, 7, 4, 39), (2, 33, 16, 5, 13)])
def test_deconv2d_k4s2_few_output_channels_and_odd_widths(shape):
    """ConvTranspose2d(4, 2, 1) (module.py:453-457) with Cout <= 8 (the entry assumed a 16-wide weight pitch where its callers lay
    out 8) and with an odd input width (the backward's space-to-depth pass refused rows that are not 16-byte multiples)."""
    import temporalstereo_amd.functional as TF
    dev = torch.device("cuda:0")
    B, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Cin, Cout, 4, 4, generator=g, dtype=torch.float64) / (4 * Cin) ** 0.5
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    go = torch.randn(B, Cout, 2 * H, 2 * W, generator=g, dtype=torch.float64)
    ref_in = [t.clone().requires_grad_() for t in (x, w, b)]
    F.conv_transpose2d(*ref_in, stride=2, padding=1).backward(go)
    gpu_in = [t.float().to(dev).requires_grad_() for t in (x, w, b)]
    y = TF.conv_transpose2d_k4s2(*gpu_in)
    y.backward(go.float().to(dev))
    want = F.conv_transpose2d(x, w, b, stride=2, padding=1)
    np.testing.assert_allclose(y.detach().cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-5)
    for a, r in zip(gpu_in, ref_in):
        scale = float(r.grad.abs().max())
        np.testing.assert_allclose(a.grad.cpu().numpy(), r.grad.numpy(), rtol=1e-4, atol=2e-5 * scale)


def test_sampled_paths_refuse_single_row_or_column_maps():
    """With one row (or column, or candidate) the reference's coordinate normalisation divides by zero (inverse_warp_3d.py:45-47) and
    its output is whatever grid_sample makes of NaN: the kernels say so instead of answering."""
    import temporalstereo_amd.functional as TF
    dev = torch.device("cuda:0")
    for H, W in ((1, 12), (12, 1)):
        L = torch.randn(1, 8, H, W, device=dev)
        d = torch.rand(1, 3, H, W, device=dev)
        with pytest.raises(RuntimeError, match="H, W >= 2"):
            TF.cat_fms(L, L, d)

"""Oracle (test infrastructure): the correlation volumes of architecture/modeling/aggregation/utils/correlation.py:10-57.

PARITY UNPINNED by reference output: the arithmetic lives in the third-party package `spatial_correlation_sampler`
(ClementPinard/Pytorch-Correlation-extension), which the reference imports inside try/except (correlation.py:4-7), does not
vendor and does not pin; it is absent here.  This restates the sampler's published definition for the arguments the reference
passes (kernel_size 1, stride 1, padding 0, dilation 1, dilation_patch 1):
    out[b, ph, pw, y, x] = sum_c in1[b, c, y, x] * in2[b, c, y + ph - pH//2, x + pw - pW//2]      (zeros outside the image)
followed by the reference's own reshape / slice / leaky_relu(0.1) (correlation.py:22-27, :48-55), and is pinned by analytic
known-answer tests (tests/test_correlation_gpu.py: identical maps, integer shifts, the reversed plane order of correlation1d)."""
import torch
import torch.nn.functional as F


def _sampler(in1, in2, pH, pW):
    B, C, H, W = in1.shape
    ry, rx = pH // 2, pW // 2
    pad = F.pad(in2, (rx, rx, ry, ry))
    out = in1.new_zeros(B, pH, pW, H, W)
    for ph in range(pH):
        for pw in range(pW):
            out[:, ph, pw] = (in1 * pad[:, :, ph:ph + H, pw:pw + W]).sum(1)
    return out


def correlation(reference_fm, target_fm, patch_size=1):
    out = _sampler(reference_fm, target_fm, patch_size, patch_size)
    B, pH, pW, H, W = out.shape
    return F.leaky_relu(out.reshape(B, pH * pW, H, W), negative_slope=0.1)


def correlation1d(reference_fm, target_fm, max_disp=1):
    out = _sampler(reference_fm, target_fm, 1, 2 * max_disp - 1)
    B, pH, pW, H, W = out.shape
    return F.leaky_relu(out.reshape(B, pH * pW, H, W)[:, :max_disp], negative_slope=0.1)
